// fp32-equivalent 3x3 convolution on the f16 matrix cores: every fp32 product a*b is formed as
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,   x_hi = f16(x), x_lo = f16(x - x_hi)
// (Ootomo-Yokota error-corrected splitting) with fp32 accumulation inside v_mfma_f32_32x32x16_f16.
// The dropped a_lo*b_lo term and the rounding of the lo parts are ~2^-22 relative, i.e. fp32 class:
// on the full-depth SR x4 / x8 / rescaling nets the end-to-end deviation from an fp64 evaluation is
// 4.4e-6 .. 5.6e-6 versus 3.1e-6 .. 4.4e-6 for plain fp32 (tools/split_precision_check.py), far inside
// the 1e-4 parity gate of BASELINE.json. f16 MFMA runs at 16x the fp32 MFMA rate, so three of them per
// fp32-equivalent K=16 step lift the compute ceiling 5.3x (157 -> 833 TFLOP/s-equivalent).
//
// Layout choices
//   * HBM tensors stay plain fp32 NHWC (same Views, same epilogue as hcf_conv.hip). The split happens
//     once per staged element, in registers, between the global load and the LDS write (~3 VALU per
//     element, hidden under the other waves' MFMAs); every staged element is then reused by 9 taps x
//     32..64 output channels.
//   * LDS record per halo pixel: [16 hi halves | 16 lo halves | 16 B pad] = 80 B. The 80-byte pixel
//     stride makes the ds_read_b128 fragment reads (lane = pixel, 8 halves each) bank-conflict free.
//   * One LDS stage = one MFMA K (16 channels); 2 stages x 27.2 KB -> up to 3 blocks per CU.
//   * Weights are pre-split on the host into TWO f16 planes per (chunk, tap): P1 = b_hi*2^11 and
//     P2 = b_lo*2^11 (the 2^11 keeps b_lo's mantissa out of the f16 subnormal range). The activation
//     lo part is left unscaled (a_lo = f16(a - a_hi), f16 subnormals are kept by the matrix core: its
//     absolute error is <= 2^-25, i.e. fp32-class for |a| >~ 0.25 and a 3e-8 absolute floor below), so
//     the third term a_lo*P1 re-uses plane 1: acc = 2^11 (a_hi b_hi + a_hi b_lo + a_lo b_hi) in ONE
//     fp32 accumulator per tile, un-scaled in the epilogue. Two planes instead of three matter: the
//     kernel is L1/TA-bandwidth bound on these weight-fragment loads (profiles/r01_f16x3_notes.md).
//   * Waves: NTB = 2 (33..64 out channels): wave = (row half, n tile), 4 row tiles x 1 n tile each,
//     so a wave re-uses each weight fragment over 4 MFMA rows; NTB = 1: 4 waves x 2 rows.
//   * |a| >= 65504 cannot be represented by the hi part: such inputs raise a device flag and the engine
//     re-runs the pass on the exact fp32 kernel (hcf_conv.hip).
#include "hcf_common.h"

namespace hcf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// explicit global address space: a pointer rebuilt from SGPR halves would otherwise be "generic" and
// its loads become flat_load, which also tick lgkmcnt and so serialise with every LDS wait
typedef const float __attribute__((address_space(1)))* gfptr;
typedef const f32x4 __attribute__((address_space(1)))* gf4ptr;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

int g_f16x3_ablation = 0;

namespace f16x3 {

constexpr int KC = 16;
constexpr int TH = 8;
constexpr int TW = 32;
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

// ABL: timing ablations for tools/conv_bench.py (results are WRONG for ABL != 0; never used by the engine)
//   bit0 no weight loads in the loop, bit1 no staging after chunk 0, bit2 no LDS fragment reads, bit3 no barrier,
//   bit4 no MFMA
template <int TAPS, int NTB, bool VEC, int ABL = 0>
__global__ __launch_bounds__(256, 3) void conv_f16x3_kernel(const ConvArgs a) {
  static_assert(TAPS % 3 == 0, "weight prefetch ring is 3 taps deep");
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  constexpr int HH = TH + 2 * PAD, HW = TW + 2 * PAD, HP = HH * HW;
  constexpr int NLOAD = HP * (KC / 4);
  constexpr int NSLOT = (NLOAD + 255) / 256;
  constexpr int NPAD = NTB * 32;
  constexpr int MT = 2 * NTB;                       // 32-pixel row tiles per wave
  constexpr int STAGE = HP * REC;                   // bytes
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
  const unsigned long long dbg_c0 = a.dbg ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long dbg_r0 = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const int wm = (NTB == 2) ? (wave >> 1) : wave;   // which group of MT tile rows
  const int wn = (NTB == 2) ? (wave & 1) : 0;       // which 32-channel n tile
  const int H = a.H, W = a.W;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int txb = bid % tiles_x;
  const int tyb = (bid / tiles_x) % tiles_y;
  const int b = bid / (tiles_x * tiles_y);
  const int x0 = txb * TW, y0 = tyb * TH;

  int pos[NSLOT], pix0[NSLOT];
  unsigned okmask = 0;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int q = tid + 256 * s;
    const int hp = min(q >> 2, HP - 1);
    const int hy = hp / HW, hx = hp - hy * HW;
    const int y = y0 + hy - PAD, x = x0 + hx - PAD;
    const bool ok = y >= 0 && y < H && x >= 0 && x < W;
    okmask |= ok ? (1u << s) : 0u;
    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
    pos[s] = (yc << 16) | xc;
    pix0[s] = (b * H + yc) * W + xc;               // pixel index for sources read at full resolution
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

  f32x4 stg[NSLOT];
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
      if (a.any_up) { /* kernel-uniform: only conv_first reads upsampled windows */               \
        const int y = (pos[s] >> 16) >> ups, x = (pos[s] & 0xffff) >> ups;                        \
        pidx = (b * Hs + y) * Ws + x;                                                             \
      }                                                                                           \
      gfptr p = sp + (unsigned)(pidx * css); /* launcher guarantees < 2^31 elements per tensor */ \
      f32x4 v;                                                                                    \
      if (VEC) {                                                                                  \
        v = *(gf4ptr)(p);                                                                         \
      } else {                                                                                    \
        v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];                                           \
      }                                                                                           \
      const bool ok = (okmask >> s) & 1u;                                                         \
      v.x = (ok && valid > 0) ? v.x : 0.f;                                                        \
      v.y = (ok && valid > 1) ? v.y : 0.f;                                                        \
      v.z = (ok && valid > 2) ? v.z : 0.f;                                                        \
      v.w = (ok && valid > 3) ? v.w : 0.f;                                                        \
      stg[s] = v;                                                                                 \
    }                                                                                             \
  }
  // split in registers, then two 8-byte LDS writes per staged float4 (hi half-plane, lo half-plane)
#define HCF_STAGE_WRITE(BUF)                                                                      \
  {                                                                                               \
    _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                           \
      const int q = tid + 256 * s;                                                                \
      const f32x4 v = stg[s];                                                                     \
      f16x4 hi, lo;                                                                               \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
        const _Float16 h = (_Float16)v[e];                                                        \
        hi[e] = h;                                                                                \
        lo[e] = (_Float16)(v[e] - (float)h);                                                      \
      }                                                                                           \
      if (q < NLOAD) {                                                                            \
        char* rec = lds + (BUF) * STAGE + (q >> 2) * REC + (q & 3) * 8;                           \
        *reinterpret_cast<f16x4*>(rec) = hi;                                                      \
        *reinterpret_cast<f16x4*>(rec + 32) = lo;                                                 \
      }                                                                                           \
    }                                                                                             \
  }

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // B streams: per (chunk, tap) three planes [which][npad][16 halves]; this lane's 8 halves
  const _Float16* wp = reinterpret_cast<const _Float16*>(a.wpack) + (size_t)(wn * 32 + li) * 16 + half * 8;
  constexpr int WSTEP = 2 * NPAD * 16;              // halves per (chunk, tap): planes P1, P2
  // 3-tap register ring: the fragment for tap t+2 is requested while tap t computes, and BEFORE the
  // chunk's staging loads, so the in-order vmcnt never makes a weight wait sit behind HBM latency
  f16x8 bring[3][2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int j = 0; j < 2; ++j) bring[d][j] = *reinterpret_cast<const f16x8*>(wp + j * NPAD * 16);
    wp += WSTEP;
  }

  // A fragment base (bytes) inside a stage for tile row MT*wm + m
  const int abase = ((MT * wm) * HW + li) * REC + half * 16;

  HCF_STAGE_LOAD(0);
  HCF_STAGE_WRITE(0);
  __syncthreads();

  const int nchunk = a.nchunk;
  for (int c = 0; c < nchunk; ++c) {
    const char* A = lds + (c & 1) * STAGE;
    const bool more = (c + 1 < nchunk);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      if (!(ABL & 1)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bring[(t + 2) % 3][j] = *reinterpret_cast<const f16x8*>(wp + j * NPAD * 16);
      }
      wp += WSTEP;
      if (t == 0 && more && !(ABL & 2)) HCF_STAGE_LOAD(c + 1);
      // pin the issue point: left alone, the scheduler sinks these loads next to their first use
      // (two taps later) to save registers, which exposes the full L2/HBM latency on every tap
      __builtin_amdgcn_sched_barrier(0);
      const int dy = (TAPS == 9) ? t / 3 : 0, dx = (TAPS == 9) ? t % 3 : 0;
      f16x8 ahi[MT], alo[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const char* rec = A + abase + ((m + dy) * HW + dx) * REC;
        if (!(ABL & 4)) {
          ahi[m] = *reinterpret_cast<const f16x8*>(rec);
          alo[m] = *reinterpret_cast<const f16x8*>(rec + 32);
        } else {
          ahi[m] = bring[0][0]; alo[m] = bring[1][1];
          asm volatile("" : "+v"(ahi[m]), "+v"(alo[m]));
        }
      }
      if (ABL & 16) {
#pragma unroll
        for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(ahi[m]), "v"(alo[m]), "v"(bring[t % 3][0]), "v"(bring[t % 3][1]));
        continue;
      }
      // term-major order: consecutive MFMAs hit different accumulators (MT independent chains)
#pragma unroll
      for (int m = 0; m < MT; ++m)   // a_hi * (b_hi 2^11)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], bring[t % 3][0], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)   // a_hi * (b_lo 2^11)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], bring[t % 3][1], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)   // a_lo * (b_hi 2^11)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[m], bring[t % 3][0], acc[m], 0, 0, 0);
    }
    if (more && !(ABL & 2)) HCF_STAGE_WRITE((c + 1) & 1);
    if (!(ABL & 8)) __syncthreads();
  }
#undef HCF_STAGE_LOAD
#undef HCF_STAGE_WRITE

  if (a.dbg && (blockIdx.x & 1023) == 512 && tid == 0) {   // a few mid-grid blocks: shader clock vs 100 MHz reference
    atomicAdd(a.dbg + 0, __builtin_readcyclecounter() - dbg_c0);
    atomicAdd(a.dbg + 1, __builtin_amdgcn_s_memrealtime() - dbg_r0);
  }

  // ---- epilogue (same algebra as the fp32 kernel) ---------------------------------------------
  const int cout = a.out.n;
  const int oc = wn * 32 + li;
  const bool ocok = oc < cout;
  const float bias = a.bias[oc], scale = a.scale[oc];
  constexpr float UNSPLIT = 1.0f / SPLIT;
  // Range check: an input with |a| >= 65504 becomes inf in the hi plane and turns every accumulator it
  // touches into inf / NaN (inf * 0 = NaN), so testing the RAW accumulators is sufficient, and 12x
  // cheaper than testing every staged element of every chunk.
  float chk = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][r], 0.f, chk);   // stays 0 unless some acc is inf / NaN
  if (__any(chk != chk)) {
    if (lane == 0) atomicOr(a.ovf, 1);
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int y = y0 + MT * wm + m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (ocok && y < H && x < W) {
        const size_t pix = (size_t)((size_t)b * H + y) * W + x;
        float v = (acc[m][r] * UNSPLIT + bias) * scale;
        if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (a.act == ACT_LRELU) v = (v >= 0.f) ? v : v * 0.2f;
        if (a.res1.p) v = v * a.rs1 + a.res1.p[pix * a.res1.cs + a.res1.c0 + oc];
        if (a.res2.p) v = v * a.rs2 + a.res2.p[pix * a.res2.cs + a.res2.c0 + oc];
        a.out.p[pix * a.out.cs + a.out.c0 + oc] = v;
      }
    }
  }
}

template <int TAPS, int NTB>
static int launch_t(const ConvArgs& a, hipStream_t st) {
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long long nblk = (long long)a.B * tiles_x * tiles_y;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return HCF_ERR_ARG;
  bool vec = true;
  ConvArgs b = a;
  b.any_up = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    vec = vec && (((a.src[i].cs | a.src[i].c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.src[i].p) & 15) == 0);
    if (a.src[i].up) b.any_up = 1;
    // 32-bit element offsets inside the kernel
    if ((long long)a.B * (a.H >> a.src[i].up) * (a.W >> a.src[i].up) * a.src[i].cs >= 0x7fffffffLL) return HCF_ERR_UNSUPPORTED;
  }
  if (vec && g_f16x3_ablation && TAPS == 9 && NTB == 2) {
    switch (g_f16x3_ablation) {
#define HCF_ABL(N) case N: hipLaunchKernelGGL((conv_f16x3_kernel<9, 2, true, N>), dim3((unsigned)nblk), dim3(256), 0, st, b); break;
      HCF_ABL(1) HCF_ABL(2) HCF_ABL(4) HCF_ABL(8) HCF_ABL(16) HCF_ABL(15) HCF_ABL(3) HCF_ABL(7)
#undef HCF_ABL
      default: return HCF_ERR_ARG;
    }
  } else if (vec)
    hipLaunchKernelGGL((conv_f16x3_kernel<TAPS, NTB, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  else
    hipLaunchKernelGGL((conv_f16x3_kernel<TAPS, NTB, false>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace f16x3

// a.wpack must point at the f16x3 pack (pack_conv_weights_f16x3), a.ovf at a device int
int launch_conv_f16x3(const ConvArgs& a, int taps, hipStream_t st) {
  if (a.nsrc < 1 || a.nsrc > kMaxSrc || a.H >= 32768 || a.W >= 32768 || a.H < 1 || a.W < 1 || !a.ovf) return HCF_ERR_ARG;
  for (int i = 0; i < a.nsrc; ++i)
    if ((a.H >> a.src[i].up) << a.src[i].up != a.H || (a.W >> a.src[i].up) << a.src[i].up != a.W) return HCF_ERR_ARG;
  const int nt = (a.out.n + 31) / 32;
  if (taps == 9 && nt == 1) return f16x3::launch_t<9, 1>(a, st);
  if (taps == 9 && nt == 2) return f16x3::launch_t<9, 2>(a, st);
  // 1x1 convs (FCN conv2, 2 % of the time) stay on the exact kernel: the 3-tap weight ring needs TAPS % 3 == 0
  return HCF_ERR_UNSUPPORTED;
}

}  // namespace hcf
