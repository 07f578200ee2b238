// fp32-equivalent 3x3 convolution on the f16 matrix cores: every fp32 product a*b is formed as
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,   x_hi = f16(x), x_lo = f16(x - x_hi)
// (Ootomo-Yokota error-corrected splitting) with fp32 accumulation inside v_mfma_f32_32x32x16_f16.
// The dropped a_lo*b_lo term and the rounding of the lo parts are ~2^-22 relative, i.e. fp32 class:
// on the full-depth SR x4 / x8 / rescaling nets the end-to-end deviation from an fp64 evaluation is
// 3.8e-6 .. 5.5e-6 versus 3.1e-6 .. 4.4e-6 for plain fp32 (tools/split_precision_check.py), far inside
// the 1e-4 parity gate of BASELINE.json. f16 MFMA runs at 16x the fp32 MFMA rate, so three of them per
// fp32-equivalent K=16 step lift the compute ceiling 5.3x (157 -> 833 TFLOP/s-equivalent at 2.4 GHz;
// the chip sustains ~1.45-1.6 GHz under this load, see profiles/).
//
// Structure (one block = 256 threads = 4 waves, output tile 8 rows x 32 cols x 32*NTB channels)
//   * HBM tensors stay plain fp32 NHWC (same Views / epilogue as hcf_conv.hip). The activation split
//     happens once per staged element, in registers, between the global load and the LDS write;
//     every staged element is then reused by 9 taps x 32..64 output channels.
//   * K is walked in chunks of 16 channels = one MFMA K. Per chunk LDS holds
//       A: the 10 x 34 halo tile as 80-byte records [16 hi halves | 16 lo halves | 16 B pad]
//          (the 80-byte pixel stride makes the ds_read_b128 fragment reads conflict-free), 27.2 KB
//       B: the chunk's weights for all 9 taps, two f16 planes P1 = b_hi*2^11, P2 = b_lo*2^11 laid out
//          [tap][plane][k-half][n][8 halves] (conflict-free fragment reads), 18.4 KB * NTB.
//     v1 of this kernel fetched B fragments per wave straight from global memory: with 4 waves
//     fetching identical (NTB=1) or pairwise identical (NTB=2) fragments it was L1/TA-bandwidth
//     bound (~130 KB through a 64 B/clk L1 per chunk). Staging B once per block through LDS cuts L1
//     traffic 2.4x / 1.6x; LDS read bandwidth (256 B/clk) has the headroom.
//   * LDS is single-buffered (45.6 / 64 KB -> 3 / 2 blocks per CU); the NEXT chunk travels in
//     registers: its global loads are issued before the chunk's 27*MT MFMAs, converted after them,
//     and written between two barriers (only ds_writes sit between the barriers; the other resident
//     blocks keep the matrix cores busy meanwhile).
//   * The activation lo part is left unscaled (f16 subnormals are kept by the matrix core: absolute
//     error <= 2^-25, fp32-class for |a| >~ 0.25 and a 3e-8 absolute floor below), so the third term
//     a_lo*P1 re-uses plane 1 and ONE fp32 accumulator per tile holds 2^11 (a_hi b_hi + a_hi b_lo +
//     a_lo b_hi); the 2^-11 is applied in the epilogue.
//   * Waves: NTB = 2 (33..64 out channels): wave = (row half, n tile), 4 row tiles each;
//     NTB = 1: 4 waves x 2 row tiles.
//   * Epilogue: the fp32 tile is transposed through LDS (free after the K loop) so that every lane reads its
//     residuals and stores its outputs as 16 contiguous bytes, with a branch-free activation; views that are not
//     16-byte addressable (odd channel counts) keep a scalar epilogue whose residual reads go out 8 at a time.
//   * |a| >= 65504 cannot be represented by the hi part and turns the accumulators it touches into
//     inf / NaN; the epilogue tests the raw accumulators and raises a device flag, upon which the engine
//     re-runs the pass on the exact fp32 kernel (hcf_conv.hip).
#include "hcf_common.h"
#include <cstdlib>
#include "hcf_step_math.h"

namespace hcf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// f16 hi / lo split of four fp32 values -> {hi.xy, hi.zw, lo.xy, lo.zw} as packed halves. hi = RNE f16 (v_cvt_pk_f16_f32);
// lo = f16(v - hi) in ONE instruction per value: the mixed-precision FMA reads hi as f16 and v as fp32 and writes its f16
// result into one half of the destination (6 VALU per slot instead of 10, and none of them packed-fp32, which is slow
// beside MFMAs -- MI355X_MICROARCH.md).
__device__ __forceinline__ f32x4 split4(const f32x4 v) {
  const f16x2 h01 = {(_Float16)v.x, (_Float16)v.y}, h23 = {(_Float16)v.z, (_Float16)v.w};
  const uint32_t H0 = __builtin_bit_cast(uint32_t, h01), H1 = __builtin_bit_cast(uint32_t, h23);
  uint32_t L0, L1;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(L0) : "v"(H0), "v"(v.x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L0) : "v"(H0), "v"(v.y));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(L1) : "v"(H1), "v"(v.z));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L1) : "v"(H1), "v"(v.w));
  f32x4 r;
  r.x = __builtin_bit_cast(float, H0); r.y = __builtin_bit_cast(float, H1);
  r.z = __builtin_bit_cast(float, L0); r.w = __builtin_bit_cast(float, L1);
  return r;
}
// explicit global address space: a pointer rebuilt from SGPR halves would otherwise be "generic" and
// its loads become flat_load, which also tick lgkmcnt and so serialise with every LDS wait
typedef const float __attribute__((address_space(1)))* gfptr;
typedef const f32x4 __attribute__((address_space(1)))* gf4ptr;

int g_f16x3_ablation = 0;   // tools/conv_bench.py --ablate N (bit0/bit1 toggle the tall-tile variants, see launch_t)

// Variants measured and NOT kept (profiles/r01_f16x3_notes.md has the numbers; the code is in the history):
// a dx-major sliding-window tap loop (fewer LDS reads, same time), a pre-split activation format with register staging,
// persistent blocks with the next tile's first chunk prefetched (VGPR cap: spills), a first-round block stagger.

int conv_f16x3_scaled_blocks(int B, int H, int W, int* strip_w, int* tile_h) {
  int tiles_y = (H + 7) / 8;
  const int per_row = B * ((W + 31) / 32);
  int row_tiles = per_row;
  if (strip_w) *strip_w = 0;
  if (tile_h) *tile_h = 8;
  const bool off = getenv("HCF_NO_DG_STRIP") != nullptr;             // A/B knob
  if (!off && B >= 2 && (long long)B * (W + 1) < 65536) {
    const int strips = (B * (W + 1) + 31) / 32;
    if (strips * 10 <= per_row * 9) {
      row_tiles = strips;
      if (strip_w) *strip_w = W + 1;
    }
  }
  // 4-row tiles while even they leave the grid within one block per CU (HCF_NO_DG_TH4: A/B knob)
  if (tile_h && getenv("HCF_NO_DG_TH4") == nullptr && (long long)row_tiles * ((H + 3) / 4) <= 256) {
    *tile_h = 4;
    tiles_y = (H + 3) / 4;
  }
  return row_tiles * tiles_y;
}

namespace f16x3 {

constexpr int KC = 16;
constexpr int TW = 32;             // tile width in pixels = MFMA M (one image row segment per fragment)
constexpr int REC = 80;          // bytes per halo pixel in LDS
constexpr float SPLIT = 2048.f;  // 2^11

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

__device__ __forceinline__ gfptr uniform_ptr(const float* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gfptr)(((uint64_t)hi << 32) | lo);
}

// UP: some source window is read through a nearest upsample (conv_first only); compile-time so that the
// staging macro is straight-line code (control flow inside it makes the waitcnt insertion serialise the loads)
// FUSE2 (NTB = 2 only): the block also applies the FCN's second layer (Basic.py:443-444), a 1x1 conv 64 -> 64 with
// ActNorm + ReLU, to its own output tile before storing: relu(AN1(conv3x3)) goes to LDS as split f16, 48 more MFMAs
// per wave form the 64 x 64 GEMM, and only H2 is written to HBM (the stand-alone 1x1 kernel was HBM-bound).
// TAILC > 0 (Conv2dZeros at the end of a coupling net, inverse pass): the block finishes the flow step itself
// (FlowStep.py:53-64): its h tile goes to LDS (pixel-major), then one thread per pixel applies coupling^-1, the
// invertible 1x1 conv and ActNorm^-1 to z in place. TAILC = register-array bucket for z (8 / 12 / 24 / 48).
// TH = tile height (8: 4 waves / 256 threads; 16: 8 waves / 512 threads). The taller tile halves the weight
// staging per pixel, trims the halo overhead (1.33x -> 1.2x) and halves the staging registers per thread.
// SCALED (training, data-gradient convs): the input tensor is multiplied by a power of two derived from its max |x|
// (a.in_max, device) before the split and the accumulators are scaled back in the epilogue: gradients of 1e-8 would
// otherwise fall below the f16 split's absolute floor (a_lo is unscaled).
// K1 (training: the FCNs' stand-alone 1x1 conv2 and its data gradient; the inference passes run it fused, FUSE2): one tap, no halo.
// N16 (round 5; fused-tail variants with <= 16 output channels: the Conv2dZeros of the 6- and 12-channel flow steps, whose 32-wide
// channel tile was 62 / 81 % zero padding): the products run on v_mfma_f32_16x16x32_f16 -- M = 16 pixels, N = 16 channels, K = 32 =
// [16 hi | 16 lo] of one chunk: ONE instruction forms a_hi b_hi + a_lo b_hi (B = [b_hi ; b_hi]), a second one a_hi b_lo
// (B = [b_lo ; 0]) -- 2/3 of the matrix work of the 32-wide tile (4 half-size MFMAs per 32 pixels and tap instead of 3 full-size
// ones), same pack, same LDS images.
template <int NTB, bool VEC, bool UP, bool FUSE2 = false, int TAILC = 0, int TH = 8, bool SCALED = false, bool K1 = false, bool N16 = false>
__global__ __launch_bounds__((TH == 4) ? 256 : 32 * TH, (TH == 16) ? ((NTB == 1) ? 4 : 2) : ((NTB == 1) ? (N16 ? 4 : 3) : 2)) void conv_f16x3_kernel(const ConvArgs a) {
  static_assert(!(FUSE2 && TAILC), "one fused epilogue at a time");
  static_assert(!N16 || (NTB == 1 && TAILC > 0 && (TH == 8 || TH == 4)), "the 16-wide channel tile is built for the fused-tail variants");
  static_assert(!SCALED || (!FUSE2 && TAILC == 0), "input scaling is for the plain variants");
  static_assert(TH == 8 || TH == 4 || (!FUSE2 && TAILC == 0), "the fused epilogues are sized for the 8- and 4-row tiles");
  static_assert(TH == 4 || TH == 8 || TH == 16, "tile heights");
  // TH = 4 (small grids, training sizes): the same four waves on HALF the rows each -- half the MFMAs per block on twice as many
  // blocks. A grid of at most one block per CU is bound by its blocks' MFMA streams (profiles/r04_notes.md), so the launch is as
  // long as ONE block: 4-row tiles when the 8-row grid would leave half of the CUs idle anyway.
  constexpr int NTHR = (TH == 4) ? 256 : 32 * TH;
  // Interleaving the next chunk's split into the last taps is worth ~10 % on the plain kernels. The fused-tail
  // variants keep it off: they are a handful of small launches, and the extra live registers make the 24-channel
  // variant spill. (The wrong pixels once blamed on this combination came from the tail's matrix being read with
  // uniform-address VECTOR loads, see hcf_step_math.h const_table(); tests/test_gpu_f16x3.py::test_large_grid_*.)
  constexpr bool INTERLEAVE = (TAILC == 0) && !K1;
  static_assert(!K1 || (!FUSE2 && TAILC == 0 && !UP && TH <= 8), "the one-tap form is a plain / scaled variant");
  static_assert(!FUSE2 || NTB == 2, "the fused 1x1 layer needs all 64 channels of the tile in one block");
  constexpr int TAPS = K1 ? 1 : 9, PAD = K1 ? 0 : 1;
  constexpr int HH = TH + 2 * PAD, HW = TW + 2 * PAD, HP = HH * HW;
  constexpr int NLOAD = HP * (KC / 4);
  constexpr int NSLOT = (NLOAD + NTHR - 1) / NTHR;   // float4 staging slots per thread (A)
  constexpr int NPAD = NTB * 32;
  constexpr int MT = (TH == 4) ? NTB : 2 * NTB;     // 32-pixel row tiles per wave
  constexpr int A_BYTES = HP * REC;
  // N16: only the first 16 of the pack's 32 channel columns are staged (the launcher guarantees <= 16 output channels): half the
  // weight tile in LDS (36.4 instead of 45.6 KB per block: FOUR blocks per CU instead of three -- the launch is bound by the
  // bytes its blocks keep in flight, profiles/r05_notes.md section 5) and half the weight traffic from L2
  constexpr int NB = N16 ? 16 : NPAD;               // channel columns of the weight tile kept in LDS
  constexpr int BHALF = NB * 16;                    // bytes of one (tap, plane, k-half): n x 8 halves
  constexpr int B_BYTES = TAPS * 2 * 2 * BHALF;     // per chunk
  constexpr int BV = B_BYTES / 16;                  // float4 units
  constexpr int BVG = TAPS * 2 * 2 * NPAD;          // float4 units of one chunk in the PACK (all 32 * NTB columns)
  constexpr int BSLOT = (BV + NTHR - 1) / NTHR;
  constexpr int F2_BYTES = FUSE2 ? (TH * TW) * 4 * 64 : 0;      // [256 px][4 k-chunks][16 hi | 16 lo]
  constexpr int HCS = (NTB == 1) ? 33 : 49;                      // odd row stride of the h tile (floats): conflict-free
  constexpr int T_BYTES = TAILC ? (TH * TW) * HCS * 4 : 0;
  constexpr int LDS_MAIN = A_BYTES + B_BYTES;
  constexpr int E_BYTES = (TH <= 8 && TAILC == 0) ? TH * TW * NPAD * 4 : 0;     // fp32 output tile, transposed for 16-byte stores
  constexpr int LDS_M1 = (LDS_MAIN > F2_BYTES ? (LDS_MAIN > T_BYTES ? LDS_MAIN : T_BYTES) : (F2_BYTES > T_BYTES ? F2_BYTES : T_BYTES));
  constexpr int LDS_BYTES = LDS_M1 > E_BYTES ? LDS_M1 : E_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char* const ldsB = lds + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
#ifdef HCF_CONV_TIMERS      // phase timers of tools/conv_bench.py: measurement builds only (make TIMERS=1), never in the product .so
  const unsigned long long dbg_c0 = a.dbg ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long dbg_r0 = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
  const int wm = (NTB == 2) ? (wave >> 1) : wave;   // which group of MT tile rows
  const int wn = (NTB == 2) ? (wave & 1) : 0;       // which 32-channel n tile
  const int H = a.H, W = a.W;
  const int sw = SCALED ? a.strip_w : 0, svw = a.B * sw;      // strips (ConvArgs::strip_w): virtual row width B (W + 1)
  const unsigned smagic = a.strip_magic;
  const int tiles_x = sw ? (svw + TW - 1) / TW : (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int txb = bid % tiles_x;
  const int tyb = (bid / tiles_x) % tiles_y;
  const int b = sw ? 0 : bid / (tiles_x * tiles_y);
  const int x0 = txb * TW, y0 = tyb * TH;

  int pos[NSLOT], pix0[NSLOT];
  unsigned okmask = 0;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int q = tid + NTHR * s;
    const int hp = min(q >> 2, HP - 1);
    const int hy = hp / HW, hx = hp - hy * HW;
    const int y = y0 + hy - PAD;
    int x = x0 + hx - PAD, bs = b;
    bool okx = x >= 0 && x < W;
    if (SCALED && sw) {                              // virtual column -> (image, column); column W is the zero separator
      const int vc = min(max(x, 0), svw - 1);
      bs = (int)__umulhi((unsigned)vc, smagic);
      okx = x >= 0 && x < svw && (vc - bs * sw) < W;
      x = vc - bs * sw;
    }
    const bool ok = y >= 0 && y < H && okx;
    okmask |= ok ? (1u << s) : 0u;
    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
    pos[s] = (yc << 16) | xc;
    pix0[s] = (bs * H + yc) * W + xc;              // pixel index for sources read at full resolution
  }
  const int uq = tid & 3;
  const int u0 = (a.src[0].n + 3) >> 2;
  const int u1 = u0 + ((a.nsrc > 1) ? ((a.src[1].n + 3) >> 2) : 0);
  const int u2 = u1 + ((a.nsrc > 2) ? ((a.src[2].n + 3) >> 2) : 0);
  const gfptr sp0 = uniform_ptr(a.src[0].p + a.src[0].c0);
  const gfptr sp1 = uniform_ptr((a.nsrc > 1) ? a.src[1].p + a.src[1].c0 : a.src[0].p);
  const gfptr sp2 = uniform_ptr((a.nsrc > 2) ? a.src[2].p + a.src[2].c0 : a.src[0].p);
  const int cs0 = __builtin_amdgcn_readfirstlane(a.src[0].cs), cs1 = __builtin_amdgcn_readfirstlane(a.src[1].cs),
            cs2 = __builtin_amdgcn_readfirstlane(a.src[2].cs);
  const int up0 = __builtin_amdgcn_readfirstlane(a.src[0].up), up1 = __builtin_amdgcn_readfirstlane(a.src[1].up),
            up2 = __builtin_amdgcn_readfirstlane(a.src[2].up);
  const int n0 = __builtin_amdgcn_readfirstlane(a.src[0].n), n1 = __builtin_amdgcn_readfirstlane(a.src[1].n),
            n2 = __builtin_amdgcn_readfirstlane(a.src[2].n);
  const gf4ptr wq = (gf4ptr)uniform_ptr(a.wpack) + tid;   // this thread's float4 lane of the weight stream
  const gf4ptr wq0 = (gf4ptr)uniform_ptr(a.wpack);        // (N16: LDS unit q = pack unit (q >> 4) * 32 + (q & 15): columns 0..15 of each block)
  const gfptr zpage = uniform_ptr(a.zeros);
  float in_s = 1.f, out_s = 1.f;                 // SCALED: x * in_s lands in [2^9, 2^10] at the tensor's max |x|
  int in_max_bits = 0;
  if (SCALED) {
    in_max_bits = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.in_max));
    const float mx = __builtin_bit_cast(float, in_max_bits);
    if (mx > 0.f && mx < 3.0e38f) {
      int ex = 0;
      (void)frexpf(mx, &ex);                     // mx = m * 2^ex, m in [0.5, 1)
      in_s = ldexpf(1.f, 10 - ex);
      out_s = ldexpf(1.f, ex - 10);
    }
  }

  int stg_valid = 0;       // valid channels (0..4+) of this thread's 4-channel unit in the staged chunk
  f32x4 stg[NSLOT];        // next chunk's activations: fp32 after the load, (hi, lo) f16 pairs after the split
  f32x4 stb[BSLOT];        // next chunk's weights (already split on the host)
#define HCF_STAGE_LOAD(CHUNK)                                                                     \
  {                                                                                               \
    const int u = (CHUNK) * 4 + uq;                                                               \
    const bool in0 = u < u0, in1 = u < u1, uok = u < u2;                                          \
    const int ul = in0 ? u : in1 ? (u - u0) : (u - u1);                                           \
    gfptr sp = in0 ? sp0 : in1 ? sp1 : sp2;                                                       \
    const int css = in0 ? cs0 : in1 ? cs1 : uok ? cs2 : cs0;                                      \
    const int ups = in0 ? up0 : in1 ? up1 : uok ? up2 : up0;                                      \
    const int nn = in0 ? n0 : in1 ? n1 : n2;                                                      \
    const int valid = uok ? (nn - 4 * ul) : 0;                                                    \
    sp = uok ? sp + 4 * ul : sp0;                                                                 \
    const int Hs = H >> ups, Ws = W >> ups;                                                       \
    _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                           \
      int pidx = pix0[s];                                                                         \
      if (UP) {                                                                                    \
        const int y = (pos[s] >> 16) >> ups, x = (pos[s] & 0xffff) >> ups;                        \
        pidx = (b * Hs + y) * Ws + x;                                                             \
      }                                                                                           \
      gfptr p = sp + (unsigned)(pidx * css); /* launcher guarantees < 2^31 elements per tensor */ \
      p = ((okmask >> s) & 1u) ? p : zpage; /* conv zero padding: read a page of zeros */             \
      f32x4 v;                                                                                    \
      if (VEC) {                                                                                  \
        v = *(gf4ptr)(p);                                                                         \
      } else {                                                                                    \
        v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];                                           \
      }                                                                                           \
      stg[s] = v; /* RAW load result: nothing here may consume it, or the wave waits for HBM now */ \
    }                                                                                             \
    stg_valid = valid;                                                                            \
    _Pragma("unroll") for (int s = 0; s < BSLOT; ++s) {                                           \
      const int q = tid + NTHR * s;                                                               \
      if (N16) stb[s] = wq0[(size_t)(CHUNK) * BVG + ((q < BV) ? (((q >> 4) << 5) | (q & 15)) : 0)];  \
      else stb[s] = wq[(size_t)(CHUNK) * BV + ((q < BV) ? NTHR * s : 0)];                         \
    }                                                                                             \
  }
  // split one staged slot in registers (VALU only): stg[s] <- {hi.xy, hi.zw, lo.xy, lo.zw} as packed halves.
  // Only the channel tail of a window (valid < 4: C = 3, 6, 10, 21 ...) needs masking; image borders read the zero page.
#define HCF_SPLIT_SLOT(S)                                                                         \
  {                                                                                               \
    f32x4 v = stg[S];                                                                             \
    if (SCALED) { v.x *= in_s; v.y *= in_s; v.z *= in_s; v.w *= in_s; }                            \
    if (stg_valid < 4) {                                                                          \
      v.x = (stg_valid > 0) ? v.x : 0.f;                                                          \
      v.y = (stg_valid > 1) ? v.y : 0.f;                                                          \
      v.z = (stg_valid > 2) ? v.z : 0.f;                                                          \
      v.w = 0.f;                                                                                  \
    }                                                                                             \
    stg[S] = split4(v);                                                                           \
  }
#define HCF_STAGE_SPLIT()                                                                         \
  { _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) HCF_SPLIT_SLOT(s) }
  // only LDS writes: two 8-byte pieces per activation slot (hi half-plane, lo half-plane), 16 B per weight slot
#define HCF_STAGE_WRITE()                                                                         \
  {                                                                                               \
    _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                           \
      const int q = tid + NTHR * s;                                                               \
      if (q < NLOAD) {                                                                            \
        char* rec = lds + (q >> 2) * REC + (q & 3) * 8;                                           \
        union { f16x4 h[2]; f32x4 f; } u_;                                                        \
        u_.f = stg[s];                                                                            \
        *reinterpret_cast<f16x4*>(rec) = u_.h[0];                                                 \
        *reinterpret_cast<f16x4*>(rec + 32) = u_.h[1];                                            \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int s = 0; s < BSLOT; ++s) {                                           \
      const int q = tid + NTHR * s;                                                               \
      if (q < BV) *reinterpret_cast<f32x4*>(ldsB + q * 16) = stb[s];                              \
    }                                                                                             \
  }

  // fragment bases (bytes): A for tile row MT*wm + m, B for this wave's n tile
  const int abase = ((MT * wm) * HW + li) * REC + half * 16;
  const int bbase = half * BHALF + (wn * 32 + li) * 16;

#ifdef HCF_CONV_TIMERS
  const bool dbg_on = a.dbg && (blockIdx.x & 1023) == (gridDim.x > 512 ? 512u : gridDim.x / 2) && tid == 0;     // a few mid-grid blocks
  const unsigned long long dbg_ra = dbg_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
  HCF_STAGE_LOAD(0);
#ifdef HCF_CONV_TIMERS
  const unsigned long long dbg_rb = dbg_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
  HCF_STAGE_SPLIT();
#ifdef HCF_CONV_TIMERS
  const unsigned long long dbg_rc = dbg_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
  HCF_STAGE_WRITE();
  __syncthreads();
#ifdef HCF_CONV_TIMERS
  const unsigned long long dbg_r1 = dbg_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (dbg_on) {
    atomicAdd(a.dbg + 6, dbg_ra - dbg_r0);
    atomicAdd(a.dbg + 7, dbg_rb - dbg_ra);
    atomicAdd(a.dbg + 8, dbg_rc - dbg_rb);
    atomicAdd(a.dbg + 9, dbg_r1 - dbg_rc);
  }
#endif

  f32x16 acc[N16 ? 1 : MT];
  f32x4 acc16[N16 ? MT : 1][2];                     // N16: [tile row][16-pixel half]; lane = (channel l & 15, pixels 4 (l >> 4) .. + 3)
#pragma unroll
  for (int m = 0; m < (N16 ? 1 : MT); ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
  for (int m = 0; m < (N16 ? MT : 1); ++m)
#pragma unroll
    for (int hr = 0; hr < 2; ++hr) acc16[m][hr] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l16 = lane & 15, kg = lane >> 4;        // N16 fragment coordinates: row / column l16, K group kg (8 halves each)
  const int abase16 = ((MT * wm) * HW + l16) * REC + kg * 16;          // record = [16 hi | 16 lo]: K groups 0, 1 = hi, 2, 3 = lo
  const int bbase16 = (kg & 1) * BHALF + l16 * 16;

  const int nchunk = a.nchunk;
  for (int c = 0; c < nchunk; ++c) {
    const bool more = (c + 1 < nchunk);
    if (more) HCF_STAGE_LOAD(c + 1);               // global loads fly under this chunk's MFMAs
    __builtin_amdgcn_sched_barrier(0);             // keep them here (the scheduler would sink them to the split)
    __builtin_amdgcn_s_setprio(1);                 // MFMA phase outranks the other blocks' staging phases on this SIMD (+2 %)
    if constexpr (N16) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dy = t / 3, dx = t % 3;
        const char* bt = ldsB + bbase16 + t * (4 * BHALF);
        const f16x8 b1 = *reinterpret_cast<const f16x8*>(bt);             // [b_hi k-half (kg & 1)]: multiplies a_hi (kg 0, 1) and a_lo (kg 2, 3)
        f16x8 b2 = *reinterpret_cast<const f16x8*>(bt + 2 * BHALF);       // [b_lo k-half] for a_hi; the a_lo x b_lo term is dropped
        if (kg >= 2) b2 = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        f16x8 af[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int hr = 0; hr < 2; ++hr)
            af[m][hr] = *reinterpret_cast<const f16x8*>(lds + abase16 + ((m + dy) * HW + dx + 16 * hr) * REC);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) acc16[m][hr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][hr], b1, acc16[m][hr], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) acc16[m][hr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][hr], b2, acc16[m][hr], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int dy = t / 3, dx = t % 3;
      const char* bt = ldsB + bbase + t * (4 * BHALF);
      const f16x8 b1 = *reinterpret_cast<const f16x8*>(bt);               // P1 = b_hi * 2^11
      const f16x8 b2 = *reinterpret_cast<const f16x8*>(bt + 2 * BHALF);   // P2 = b_lo * 2^11
      f16x8 ahi[MT], alo[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const char* rec = lds + abase + ((m + dy) * HW + dx) * REC;
        ahi[m] = *reinterpret_cast<const f16x8*>(rec);
        alo[m] = *reinterpret_cast<const f16x8*>(rec + 32);
      }
      // the split of the NEXT chunk's staged slots rides in the MFMA shadow of the last taps (its loads were
      // issued a whole chunk of MFMAs earlier); two slots per tap
      constexpr int SPT = 2;                         // slots per tap (3 or 6, i.e. a later start, measured the same)
      constexpr int SPLIT_T0 = TAPS - (NSLOT + SPT - 1) / SPT;
      if (INTERLEAVE && more && t >= SPLIT_T0) {
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
          const int sq = SPT * (t - SPLIT_T0) + q;
          if (sq < NSLOT) HCF_SPLIT_SLOT(sq)
        }
      }
      // term-major order: consecutive MFMAs hit different accumulators (MT independent chains)
#pragma unroll
      for (int m = 0; m < MT; ++m)   // a_hi * (b_hi 2^11)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b1, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)   // a_hi * (b_lo 2^11)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b2, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)   // a_lo * (b_hi 2^11)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[m], b1, acc[m], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!more) break;
    if (!INTERLEAVE) HCF_STAGE_SPLIT();            // (otherwise the split already ran inside the last taps)
    __syncthreads();                               // every wave has finished reading this chunk
    HCF_STAGE_WRITE();
    __syncthreads();
  }
#undef HCF_STAGE_LOAD
#undef HCF_STAGE_SPLIT
#undef HCF_STAGE_WRITE

#ifdef HCF_CONV_TIMERS
  unsigned long long dbg_r2 = 0ull;
  if (dbg_on) {   // shader clock vs 100 MHz reference; {prologue, chunk loop, epilogue} in 100 MHz ticks at [2..4], blocks at [5]
    dbg_r2 = __builtin_amdgcn_s_memrealtime();
    atomicAdd(a.dbg + 0, __builtin_readcyclecounter() - dbg_c0);
    atomicAdd(a.dbg + 1, dbg_r2 - dbg_r0);
    atomicAdd(a.dbg + 2, dbg_r1 - dbg_r0);
    atomicAdd(a.dbg + 3, dbg_r2 - dbg_r1);
    atomicAdd(a.dbg + 5, 1ull);
  }
#define HCF_DBG_EPI() { if (dbg_on) atomicAdd(a.dbg + 4, __builtin_amdgcn_s_memrealtime() - dbg_r2); }
#else
#define HCF_DBG_EPI()
#endif

  // ---- epilogue (same algebra as the fp32 kernel) ---------------------------------------------
  const int cout = a.out.n;
  const int oc = wn * 32 + li;
  const bool ocok = oc < cout;
  const float UNSPLIT = SCALED ? out_s / SPLIT : 1.0f / SPLIT;
  // Range check: an input with |a| >= 65504 becomes inf in the hi plane and turns every accumulator it
  // touches into inf / NaN (inf * 0 = NaN), so testing the RAW accumulators is sufficient, and 12x
  // cheaper than testing every staged element of every chunk.
  float chk = 0.f;
#pragma unroll
  for (int m = 0; m < (N16 ? 1 : MT); ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][r], 0.f, chk);   // stays 0 unless some acc is inf / NaN
  if constexpr (N16) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int hr = 0; hr < 2; ++hr)
#pragma unroll
        for (int j = 0; j < 4; ++j) chk = fmaf(acc16[m][hr][j], 0.f, chk);
  }

  if constexpr (FUSE2) {
    // layer 1 epilogue -> split f16 A operand in LDS: record (px, kc) = [16 hi | 16 lo] halves of channels 16kc..16kc+15
    {
      const float bias1 = a.bias[oc], scale1 = a.scale[oc];
      const float slope1 = act_slope(a.act);
      __syncthreads();                               // every wave is done with the staging buffers
      const int kc = oc >> 4, kpos = oc & 15;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (MT * wm + m) * TW + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = (acc[m][r] * UNSPLIT + bias1) * scale1;
          v = apply_act(v, slope1);
          const _Float16 h = (_Float16)v;
          // 16-byte slot (kc, plane, k-half) of the pixel's 256-byte record, XOR-swizzled with the pixel index: the A-fragment
          // reads below take the SAME slot of 32 consecutive pixels (256 bytes apart = the same four banks); with the swizzle a
          // group of 16 lanes touches 16 distinct slots (conflict-free ds_read_b128) instead of one
          const int sw = px & 15, kh = kpos >> 3, kb = (kpos & 7) * 2;
          *reinterpret_cast<_Float16*>(lds + px * 256 + (((kc * 4 + kh) ^ sw) << 4) + kb) = h;
          *reinterpret_cast<_Float16*>(lds + px * 256 + (((kc * 4 + 2 + kh) ^ sw) << 4) + kb) = (_Float16)(v - (float)h);
        }
      __syncthreads();
    }
    // layer 2: acc = sum over 4 k-chunks; weights [kc][plane][k-half][64 n][8 halves] straight from L1/L2 (8 KB per wave)
    const _Float16* w2 = reinterpret_cast<const _Float16*>(a.w2) + (size_t)half * (64 * 8) + (size_t)(wn * 32 + li) * 8;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const f16x8 b1 = *reinterpret_cast<const f16x8*>(w2 + (size_t)(kc * 2 + 0) * (2 * 64 * 8));
      const f16x8 b2 = *reinterpret_cast<const f16x8*>(w2 + (size_t)(kc * 2 + 1) * (2 * 64 * 8));
      f16x8 ahi[MT], alo[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const char* rec = lds + ((MT * wm + m) * TW + li) * 256;
        ahi[m] = *reinterpret_cast<const f16x8*>(rec + (((kc * 4 + half) ^ (li & 15)) << 4));
        alo[m] = *reinterpret_cast<const f16x8*>(rec + (((kc * 4 + 2 + half) ^ (li & 15)) << 4));
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b1, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b2, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[m], b1, acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][r], 0.f, chk);
  }
  if (__any(chk != chk)) {
    // bit 0: some input left the f16 range; bit 1 + (sample mod 30): which sample's tile saw it (hcf_check_range_samples: the
    // module re-runs only those samples exactly); a strip tile spans several samples: all of them
    if (lane == 0) atomicOr(a.ovf, sw ? 0x7fffffff : (1 | (2 << (b % 30))));
  }

  if constexpr (TAILC > 0) {
    float* hl = reinterpret_cast<float*>(lds);
    __syncthreads();                                 // every wave is done with the staging buffers
    if constexpr (N16) {
      const float bias_t = a.bias[l16], scale_t = a.scale[l16];
      if (l16 < cout) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int hr = 0; hr < 2; ++hr)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int px = (MT * wm + m) * TW + 16 * hr + 4 * kg + j;
              hl[px * HCS + l16] = (acc16[m][hr][j] * UNSPLIT + bias_t) * scale_t;
            }
      }
    } else {
    const float bias_t = a.bias[oc], scale_t = a.scale[oc];
    if (ocok) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (MT * wm + m) * TW + (r & 3) + 8 * (r >> 2) + 4 * half;
          hl[px * HCS + oc] = (acc[m][r] * UNSPLIT + bias_t) * scale_t;       // Conv2dZeros: no activation
        }
    }
    }
    __syncthreads();
    const int ty = tid >> 5, tx = tid & 31;
    const int y = y0 + ty, x = x0 + tx;
    if (ty < TH && y < H && x < W) {                  // (4-row tiles: the upper half of the block's threads has no pixel)
      const size_t pix = (size_t)((size_t)b * H + y) * W + x;
      float z[TAILC], yv[TAILC];
      load_pixel<TAILC>(a.tz, pix, a.tC, z);
      step_tail_inverse_pixel<TAILC>(z, hl + tid * HCS, a.tC, a.tns, a.tmode, a.tmat, a.tbias, a.tmul, yv);
      store_pixel<TAILC>(a.tzo, pix, a.tC, yv);
      if (a.tzpad) store_pad16<TAILC>(a.tzpad, pix, a.tzpad_n, yv);
    }
    return;
  }

  const float bias = FUSE2 ? a.bias2[oc] : a.bias[oc], scale = FUSE2 ? a.scale2[oc] : a.scale[oc];
  const float slope = act_slope(FUSE2 ? a.act2 : a.act);
  bool done_vec = false;
  if constexpr (TH <= 8) {
    if (a.vec_epi) {
      // The tile goes through LDS (pixel-major fp32) so that every lane loads its residuals and stores its outputs as
      // 16 contiguous bytes: 64 scalar dword stores per lane cost ~12 us per block on the 64-channel tile (measured in
      // hcf_conv_f16x3_dma.hip), the transposed form half of that, and the residual reads become coalesced float4s.
      float* const ldsT = reinterpret_cast<float*>(lds);
      __syncthreads();                               // every wave is done with the staging buffers
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (MT * wm + m) * TW + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = (acc[m][r] * UNSPLIT + bias) * scale;
          v = apply_act(v, slope);
          ldsT[px * NPAD + oc] = v;
        }
      __syncthreads();
      constexpr int C4 = NPAD / 4;
      const int n4 = a.out.n >> 2;
      const bool h1 = a.res1.p != nullptr, h2 = a.res2.p != nullptr;
      const bool fb = SCALED && a.fb_y.p != nullptr;        // fused epilogue backward of the conv whose dL/dy this is (ConvArgs::fb_y)
      const float fb_neg = a.fb_act == ACT_RELU ? 0.f : a.fb_act == ACT_LRELU ? 0.2f : 1.f;
      f32x4 fb_sum = {0.f, 0.f, 0.f, 0.f}, fb_szy = {0.f, 0.f, 0.f, 0.f};
      f32x4 fb_sc = {1.f, 1.f, 1.f, 1.f};            // (this thread's channel quad is the same in every iteration: NTHR % C4 == 0)
      if (fb && a.fb_scale && (tid % C4) < n4) fb_sc = *reinterpret_cast<const f32x4*>(a.fb_scale + 4 * (tid % C4));
      float fb_mx = 0.f;
#pragma unroll 2
      for (int k = 0; k < (TH * TW * C4) / NTHR; ++k) {
        const int idx = tid + NTHR * k;
        const int px = idx / C4, c4 = idx - px * C4;
        const int y = y0 + (px >> 5);
        int x = x0 + (px & 31), bo = b;
        bool okx = x < W;
        if (SCALED && sw) {
          const int vc = min(x, svw - 1);
          bo = (int)__umulhi((unsigned)vc, smagic);
          okx = x < svw && (vc - bo * sw) < W;
          x = vc - bo * sw;
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(ldsT + px * NPAD + 4 * c4);
        if (y < H && okx && c4 < n4) {
          const size_t pixo = (size_t)((size_t)bo * H + y) * W + x;
          if (h1) v = v * a.rs1 + *reinterpret_cast<const f32x4*>(a.res1.p + pixo * a.res1.cs + a.res1.c0 + 4 * c4);
          if (h2) v = v * a.rs2 + *reinterpret_cast<const f32x4*>(a.res2.p + pixo * a.res2.cs + a.res2.c0 + 4 * c4);
          if (fb) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(a.fb_y.p + pixo * a.fb_y.cs + a.fb_y.c0 + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool pos = a.fb_act == ACT_RELU ? (yv[e] > 0.f) : (yv[e] >= 0.f);      // as conv_epilogue_bwd_kernel
              const float dz = (a.fb_act == ACT_NONE || pos) ? v[e] : v[e] * fb_neg;
              v[e] = dz * fb_sc[e];
              fb_sum[e] += v[e];
              fb_szy[e] += dz * yv[e];
              fb_mx = fmaxf(fb_mx, fabsf(v[e]));
            }
          }
          *reinterpret_cast<f32x4*>(a.out.p + pixo * a.out.cs + a.out.c0 + 4 * c4) = v;
        }
      }
      if (fb) {
        // every thread owns ONE channel quad (NTHR % C4 == 0) of TH * TW * C4 / NTHR pixels: per-channel sums over the block's
        // threads in a fixed order through LDS, one row of partials per block (reduced later by launch_sum_jobs)
        static_assert(NTHR % C4 == 0, "one channel quad per thread");
        __syncthreads();                               // the tile in LDS has been read
#pragma unroll
        for (int e = 0; e < 4; ++e) { ldsT[tid * 4 + e] = fb_sum[e]; ldsT[NTHR * 4 + tid * 4 + e] = fb_szy[e]; }
        int* const fb_shm = reinterpret_cast<int*>(ldsT + 2 * NTHR * 4);      // (a word of the tile buffer: no extra static LDS)
        static_assert(LDS_BYTES >= (2 * NTHR * 4 + 1) * 4, "the partial sums fit in the LDS buffer");
        if (tid == 0) *fb_shm = 0;
        __syncthreads();
        if (fb_mx > 0.f && fb_mx == fb_mx) atomicMax(fb_shm, __builtin_bit_cast(int, fb_mx));
        const int n = a.out.n;
        if (tid < n) {
          const int q4 = tid >> 2, e = tid & 3;
          float t0 = 0.f, t1 = 0.f;
          for (int j = 0; j < NTHR / C4; ++j) { t0 += ldsT[(j * C4 + q4) * 4 + e]; t1 += ldsT[NTHR * 4 + (j * C4 + q4) * 4 + e]; }
          a.fb_part[((size_t)blockIdx.x * 2 + 0) * n + tid] = t0;
          a.fb_part[((size_t)blockIdx.x * 2 + 1) * n + tid] = a.fb_zy ? t1 : 0.f;
        }
        __syncthreads();
        if (tid == 0) {
          const int m = *fb_shm, m2 = max(m, in_max_bits);      // fb_max2: the running max, carried on from in_max's slot
          if (a.fb_max && m) atomicMax(reinterpret_cast<int*>(a.fb_max), m);
          if (a.fb_max2 && m2) atomicMax(reinterpret_cast<int*>(a.fb_max2), m2);
        }
      }
      done_vec = true;
    }
  }
  if (!done_vec) {
  // Residual reads are issued RB at a time (per lane and residual) before the first one is used: inside the
  // bounds-checked store loop each load sat behind its own s_waitcnt, i.e. 64 serialised L2/HBM latencies per
  // residual (RDB conv5: +8 % with one residual, +40 % with the RRDB skip as well). Out-of-tile lanes read a
  // clamped (valid) address and never store.
  const bool has1 = a.res1.p != nullptr, has2 = a.res2.p != nullptr;
  constexpr int RB = 8;                              // residual values in flight per lane and residual
  const int occ = ocok ? oc : 0;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int y = y0 + MT * wm + m;
    const int yc = y < H ? y : H - 1;
#pragma unroll
    for (int rb = 0; rb < 16; rb += RB) {
      float r1[RB], r2[RB];
      if (has1 || has2) {
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          const int r = rb + q;
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const size_t pixc = (size_t)((size_t)b * H + yc) * W + (x < W ? x : W - 1);
          r1[q] = has1 ? a.res1.p[pixc * a.res1.cs + a.res1.c0 + occ] : 0.f;
          r2[q] = has2 ? a.res2.p[pixc * a.res2.cs + a.res2.c0 + occ] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int r = rb + q;
        const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ocok && y < H && x < W) {
          const size_t pix = (size_t)((size_t)b * H + y) * W + x;
          float v = (acc[m][r] * UNSPLIT + bias) * scale;
          v = apply_act(v, slope);
          if (has1) v = v * a.rs1 + r1[q];
          if (has2) v = v * a.rs2 + r2[q];
          a.out.p[pix * a.out.cs + a.out.c0 + oc] = v;
        }
      }
    }
  }
  }
  HCF_DBG_EPI()
#undef HCF_DBG_EPI
}

int g_f16x3_tall = 0;   // 16-row tile variants measured no better than the 8-row tile (profiles/r01_f16x3_notes.md); bit0 NTB=1, bit1 NTB=2

template <int NTB>
static int launch_t(const ConvArgs& a, hipStream_t st) {
  const bool plain = !(a.tC > 0) && !a.w2 && !a.in_max;
  const bool tall = plain && (((g_f16x3_tall ^ g_f16x3_ablation) >> (NTB - 1)) & 1) && a.H >= 16;   // --ablate 1/2/3 turns it off
  const int THr = tall ? 16 : 8;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + THr - 1) / THr;
  long long nblk = (long long)a.B * tiles_x * tiles_y;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return HCF_ERR_ARG;
  bool vec = true;
  ConvArgs b = a;
  b.any_up = 0;
  b.strip_w = 0; b.strip_magic = 0;
  {
    auto v4 = [](const View& v) { return !v.p || ((((v.cs | v.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0)); };
    b.vec_epi = !tall && !(a.tC > 0) && a.out.p && (a.out.n & 3) == 0 && v4(a.out) && v4(a.res1) && v4(a.res2) &&
                !(g_f16x3_ablation & 64);                                      // --ablate 64: scalar epilogue
  }
  if (a.fb_y.p) {      // fused epilogue backward: the scaled, vector-epilogue variant only, 16-byte addressable y, whole channel quads
    auto v4b = [](const View& v) { return (((v.cs | v.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0); };
    if (!a.in_max || a.fb_max2 == a.in_max || !b.vec_epi || !a.fb_part || !v4b(a.fb_y) || (a.out.n & 3) || a.out.n > 32 * NTB ||
        (a.fb_scale && (reinterpret_cast<uintptr_t>(a.fb_scale) & 15))) return HCF_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < a.nsrc; ++i) {
    vec = vec && (((a.src[i].cs | a.src[i].c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.src[i].p) & 15) == 0);
    if (a.src[i].up) b.any_up = 1;
    // 32-bit element offsets inside the kernel
    if ((long long)a.B * (a.H >> a.src[i].up) * (a.W >> a.src[i].up) * a.src[i].cs >= 0x7fffffffLL) return HCF_ERR_UNSUPPORTED;
  }
  if (a.tC > 0) {  // fused inverse flow-step tail
    if (!vec || b.any_up || a.w2 || a.res1.p || a.res2.p || a.act != ACT_NONE || a.out.n > ((NTB == 1) ? 32 : 48)) return HCF_ERR_ARG;
    const int cm = step_cmax(a.tC);
    if constexpr (NTB == 1) {
      const long long nblk4 = (long long)a.B * tiles_x * ((a.H + 3) / 4);
      const bool th4 = nblk4 <= 256 && getenv("HCF_NO_TH4") == nullptr;      // small grids: 4-row tiles (see the plain variant below)
      static const bool n16_off = getenv("HCF_NO_N16") != nullptr;              // A/B knob: the 16-wide channel tile (<= 16 output channels)
      const bool n16 = a.out.n <= 16 && !n16_off && !(g_f16x3_ablation & 2048);
      if (n16 && th4 && cm == 8) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 8, 4, false, false, true>), dim3((unsigned)nblk4), dim3(256), 0, st, b);
      else if (n16 && th4 && cm == 12) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 12, 4, false, false, true>), dim3((unsigned)nblk4), dim3(256), 0, st, b);
      else if (n16 && !th4 && cm == 8) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 8, 8, false, false, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
      else if (n16 && !th4 && cm == 12) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 12, 8, false, false, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
      else
      if (th4 && cm == 8) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 8, 4>), dim3((unsigned)nblk4), dim3(256), 0, st, b);
      else if (th4 && cm == 12) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 12, 4>), dim3((unsigned)nblk4), dim3(256), 0, st, b);
      else if (th4 && cm == 24) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 24, 4>), dim3((unsigned)nblk4), dim3(256), 0, st, b);
      else
      if (cm == 8) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 8>), dim3((unsigned)nblk), dim3(256), 0, st, b);
      else if (cm == 12) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 12>), dim3((unsigned)nblk), dim3(256), 0, st, b);
      else if (cm == 24) hipLaunchKernelGGL((conv_f16x3_kernel<1, true, false, false, 24>), dim3((unsigned)nblk), dim3(256), 0, st, b);
      else return HCF_ERR_UNSUPPORTED;
    } else {
      return HCF_ERR_UNSUPPORTED;     // 45/48-channel steps (x8 level 2) keep the stand-alone tail kernel
    }
  } else if (a.w2) {      // fused FCN conv1 + conv2
    if constexpr (NTB == 2) {
      if (!vec || a.out.n != 64 || !a.bias2 || !a.scale2 || a.res1.p || a.res2.p) return HCF_ERR_ARG;
      const long long nblk4 = (long long)a.B * tiles_x * ((a.H + 3) / 4);
      if (b.any_up)
        hipLaunchKernelGGL((conv_f16x3_kernel<2, true, true, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
      else if (b.vec_epi && nblk4 <= 256 && getenv("HCF_NO_TH4") == nullptr)      // small grids: 4-row tiles
        hipLaunchKernelGGL((conv_f16x3_kernel<2, true, false, true, 0, 4>), dim3((unsigned)nblk4), dim3(256), 0, st, b);
      else
        hipLaunchKernelGGL((conv_f16x3_kernel<2, true, false, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
    } else {
      return HCF_ERR_ARG;
    }
  } else if (tall && vec && !b.any_up)
    hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false, false, 0, 16>), dim3((unsigned)nblk), dim3(512), 0, st, b);
  else if (tall && vec)
    hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, true, false, 0, 16>), dim3((unsigned)nblk), dim3(512), 0, st, b);
  else if (tall)
    hipLaunchKernelGGL((conv_f16x3_kernel<NTB, false, true, false, 0, 16>), dim3((unsigned)nblk), dim3(512), 0, st, b);
  else if (a.in_max && vec && !b.any_up) {
    int th = 8;
    if (b.vec_epi) {                               // strips on narrow images (the vector epilogue maps pixels back per image)
      nblk = conv_f16x3_scaled_blocks(a.B, a.H, a.W, &b.strip_w, &th);
      b.strip_magic = b.strip_w ? (unsigned)(0x100000000ull / (unsigned)b.strip_w) + 1u : 0u;
    }
    if (th == 4) hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false, false, 0, 4, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
    else
    hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false, false, 0, 8, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  }
  else if (a.in_max)
    return HCF_ERR_UNSUPPORTED;
  else if (vec && !b.any_up) {
    // small grids: 4-row tiles while even they stay within one block per CU (bit-identical to the 8-row form; HCF_NO_TH4: A/B knob)
    const long long nblk4 = (long long)a.B * tiles_x * ((a.H + 3) / 4);
    if (b.vec_epi && nblk4 <= 256 && getenv("HCF_NO_TH4") == nullptr)
      hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false, false, 0, 4>), dim3((unsigned)nblk4), dim3(256), 0, st, b);
    else
    hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  }
  else if (vec)
    hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  else
    hipLaunchKernelGGL((conv_f16x3_kernel<NTB, false, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// stand-alone 1x1 conv (training): 16-byte addressable windows, no upsample, vector epilogue; anything else stays on the exact kernel
template <int NTB>
static int launch_k1(const ConvArgs& a, hipStream_t st) {
  if (a.tC > 0 || a.w2 || getenv("HCF_NO_K1") != nullptr) return HCF_ERR_UNSUPPORTED;      // HCF_NO_K1: 1x1 convs on the exact kernel (A/B)
  ConvArgs b = a;
  b.any_up = 0; b.strip_w = 0; b.strip_magic = 0;
  auto v4 = [](const View& v) { return !v.p || ((((v.cs | v.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0)); };
  b.vec_epi = a.out.p && (a.out.n & 3) == 0 && v4(a.out) && v4(a.res1) && v4(a.res2);
  if (!b.vec_epi) return HCF_ERR_UNSUPPORTED;
  for (int i = 0; i < a.nsrc; ++i) {
    if (a.src[i].up || ((a.src[i].cs | a.src[i].c0) & 3) || (reinterpret_cast<uintptr_t>(a.src[i].p) & 15)) return HCF_ERR_UNSUPPORTED;
    if ((long long)a.B * a.H * a.W * a.src[i].cs >= 0x7fffffffLL) return HCF_ERR_UNSUPPORTED;
  }
  if (a.fb_y.p) {
    auto v4b = [](const View& v) { return (((v.cs | v.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0); };
    if (!a.in_max || a.fb_max2 == a.in_max || !a.fb_part || !v4b(a.fb_y) || a.out.n > 32 * NTB ||
        (a.fb_scale && (reinterpret_cast<uintptr_t>(a.fb_scale) & 15))) return HCF_ERR_UNSUPPORTED;
  }
  const int tiles_x = (a.W + TW - 1) / TW;
  long long nblk = (long long)a.B * tiles_x * ((a.H + 7) / 8);
  if (nblk <= 0 || nblk > 0x7fffffffLL) return HCF_ERR_ARG;
  if (a.in_max) {
    int th = 8;
    nblk = conv_f16x3_scaled_blocks(a.B, a.H, a.W, &b.strip_w, &th);
    b.strip_magic = b.strip_w ? (unsigned)(0x100000000ull / (unsigned)b.strip_w) + 1u : 0u;
    if (th == 4) hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false, false, 0, 4, true, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
    else hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false, false, 0, 8, true, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  } else {
    const long long nblk4 = (long long)a.B * tiles_x * ((a.H + 3) / 4);
    if (nblk4 <= 256 && getenv("HCF_NO_TH4") == nullptr)
      hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false, false, 0, 4, false, true>), dim3((unsigned)nblk4), dim3(256), 0, st, b);
    else
      hipLaunchKernelGGL((conv_f16x3_kernel<NTB, true, false, false, 0, 8, false, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  }
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace f16x3

// a.wpack must point at the f16x3 pack (pack_conv_weights_f16x3), a.ovf at a device int
int launch_conv_f16x3(const ConvArgs& a, int taps, hipStream_t st) {
  if (a.nsrc < 1 || a.nsrc > kMaxSrc || a.H >= 32768 || a.W >= 32768 || a.H < 1 || a.W < 1 || !a.ovf) return HCF_ERR_ARG;
  for (int i = 0; i < a.nsrc; ++i)
    if ((a.H >> a.src[i].up) << a.src[i].up != a.H || (a.W >> a.src[i].up) << a.src[i].up != a.W) return HCF_ERR_ARG;
  const int nt = (a.out.n + 31) / 32;
  if (taps == 9 && nt == 1) return f16x3::launch_t<1>(a, st);
  if (taps == 9 && nt == 2) return f16x3::launch_t<2>(a, st);
  if (taps == 1 && nt == 1) return f16x3::launch_k1<1>(a, st);
  if (taps == 1 && nt == 2) return f16x3::launch_k1<2>(a, st);
  // > 64 output channels stay on the exact kernel
  return HCF_ERR_UNSUPPORTED;
}

}  // namespace hcf
