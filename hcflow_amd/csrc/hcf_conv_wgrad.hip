// Weight gradient of the fused 3x3 / 1x1 convolution (training path) for gfx950.
//
//   dW[oc][ic][tap] += sum over (b, y, x) of  X[b, y + dy - pad, x + dx - pad, ic] * G[b, y, x, oc]
//
// X is the forward input (the same <= 3 NHWC source windows, optionally read through a nearest upsample), G the
// gradient w.r.t. the conv's pre-activation output. Per tap this is a GEMM with M = input channels, N = output
// channels, K = pixels, run on v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation):
//   block  = 4 waves, one (32 input channels) x (32 output channels) tile of dW, all taps; each wave owns a 16 x 16
//            quadrant (16x16x4 MFMAs: 4 accumulator registers per tap -- a 32x32 tile per wave needs 144 and spilled);
//   K loop = the block walks `tpb` pixel tiles of 8 x 32; per tile the 10 x 34 halo of X (32 channels) and the
//            8 x 32 tile of G (32 channels) are staged in LDS as [pixel][32 ch] (conflict-free ds_read_b32), every
//            wave runs 64 K-steps of four pixels, 9 MFMAs each (one per tap: the A operand is the same halo tile
//            shifted by the tap); the next tile's global loads are in flight meanwhile (register staging);
//   end    = each wave stores its quadrant of the block's partial dW tile (scratch), and
//            wgrad_reduce_kernel adds the partial tiles of the blocks that share a dW tile in a fixed order
//            (deterministic; the first version used fp32 atomics and spent 10-25x the MFMA time in them).
#include "hcf_common.h"
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace hcf {
namespace wgrad {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;

template <int TAPS, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  constexpr int HH = TH + 2 * PAD, HW = TW + 2 * PAD, HP = HH * HW;
  __shared__ __attribute__((aligned(16))) float xs[HP * 32];
  __shared__ __attribute__((aligned(16))) float gs[TH * TW * 32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int ntiles = a.B * tiles_x * tiles_y;

  // which 32-channel block of which source window
  int blk = blockIdx.y, si = 0;
  for (; si < a.nsrc; ++si) {
    const int nb = (a.src[si].n + 31) >> 5;
    if (blk < nb) break;
    blk -= nb;
  }
  const View sv = a.src[si];
  const int ic0 = blk * 32;                       // first channel of the block inside the window
  const int icn = min(32, sv.n - ic0);            // valid channels
  const int oc0 = blockIdx.z * 32;
  const int ocn = min(32, a.g.n - oc0);
  const int up = sv.up, Hs = H >> up, Ws = W >> up;

  // wave (qi, qj) owns the 16 x 16 quadrant [16 qi, +16) x [16 qj, +16) of the 32 x 32 dW tile for all taps
  // (v_mfma_f32_16x16x4_f32: 4 accumulator registers per tap) and walks all 256 pixels of every tile
  const int qi = wave >> 1, qj = wave & 1, l16 = lane & 15, kk = lane >> 4;
  f32x4 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // register staging: the NEXT tile's global loads are issued before this tile's MFMAs and land under them.
  // Branch-free: coordinates and channel offsets are clamped into the tensor so every load is unconditional
  // (per-slot branches made the compiler wait for each load in turn: 19 serial memory latencies per tile);
  // out-of-image pixels and channel tails are zeroed by masks when the registers go to LDS.
  constexpr int NX = (HP * 8 + 255) / 256, NG = (TH * TW * 8) / 256;
  f32x4 rx[NX], rg[NG];
  unsigned mskx = 0, mskg = 0;                       // bit s: the pixel of slot s is inside the image
  const int c4t = (tid & 7) * 4;                     // this thread's 4-channel unit (same in every slot)
  const int vx = max(0, min(4, icn - c4t)), vg = max(0, min(4, ocn - c4t));      // valid channels of the unit
  const int c4x = min(c4t, max(0, (icn - 1) & ~3)), c4g = min(c4t, max(0, (ocn - 1) & ~3));
  const float* const xbase = sv.p + sv.c0 + ic0;
  const float* const gbase = a.g.p + a.g.c0 + oc0;
#define HCF_WG_LOAD(TILE)                                                                                     \
  {                                                                                                           \
    const int txb_ = (TILE) % tiles_x, tyb_ = ((TILE) / tiles_x) % tiles_y, b_ = (TILE) / (tiles_x * tiles_y); \
    const int x0_ = txb_ * TW, y0_ = tyb_ * TH;                                                               \
    mskx = 0;                                                                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < NX; ++s_) {                                                       \
      const int hp = min((tid + 256 * s_) >> 3, HP - 1);                                                      \
      const int hy = hp / HW, hx = hp - hy * HW;                                                              \
      const int y = y0_ + hy - PAD, x = x0_ + hx - PAD;                                                       \
      mskx |= (y >= 0 && y < H && x >= 0 && x < W) ? (1u << s_) : 0u;                                         \
      const int yc = min(max(y, 0), H - 1) >> up, xc = min(max(x, 0), W - 1) >> up;                           \
      const float* p = xbase + ((size_t)((size_t)b_ * Hs + yc) * Ws + xc) * sv.cs;                            \
      f32x4 v;                                                                                                \
      if (VEC) {                                                                                              \
        v = *reinterpret_cast<const f32x4*>(p + c4x);                                                         \
      } else {                                                                                                \
        v.x = p[min(c4t, icn - 1)]; v.y = p[min(c4t + 1, icn - 1)];                                           \
        v.z = p[min(c4t + 2, icn - 1)]; v.w = p[min(c4t + 3, icn - 1)];                                       \
      }                                                                                                       \
      rx[s_] = v;                                                                                             \
    }                                                                                                         \
  }
#define HCF_WG_LOAD_G(TILE)                                                                                   \
  {                                                                                                           \
    const int txb_ = (TILE) % tiles_x, tyb_ = ((TILE) / tiles_x) % tiles_y, b_ = (TILE) / (tiles_x * tiles_y); \
    const int x0_ = txb_ * TW, y0_ = tyb_ * TH;                                                               \
    mskg = 0;                                                                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < NG; ++s_) {                                                       \
      const int px = (tid + 256 * s_) >> 3;                                                                   \
      const int y = y0_ + (px >> 5), x = x0_ + (px & 31);                                                     \
      mskg |= (y < H && x < W) ? (1u << s_) : 0u;                                                             \
      const float* p = gbase + ((size_t)((size_t)b_ * H + min(y, H - 1)) * W + min(x, W - 1)) * a.g.cs;       \
      f32x4 v;                                                                                                \
      if (VEC) {                                                                                              \
        v = *reinterpret_cast<const f32x4*>(p + c4g);                                                         \
      } else {                                                                                                \
        v.x = p[min(c4t, ocn - 1)]; v.y = p[min(c4t + 1, ocn - 1)];                                           \
        v.z = p[min(c4t + 2, ocn - 1)]; v.w = p[min(c4t + 3, ocn - 1)];                                       \
      }                                                                                                       \
      rg[s_] = v;                                                                                             \
    }                                                                                                         \
  }
  const int t0 = blockIdx.x * a.tpb, t1 = min(ntiles, t0 + a.tpb);
  const int dbg = a.dbg;
  if (t0 < t1 && !(dbg & 4)) HCF_WG_LOAD(t0)
  if (t0 < t1 && !(dbg & 4)) HCF_WG_LOAD_G(t0)
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();                              // the previous tile's fragments have been read
#pragma unroll
    for (int s_ = 0; s_ < NX; ++s_) {
      const int q = tid + 256 * s_;
      f32x4 v = rx[s_];
      const bool ok = (mskx >> s_) & 1u;
      v.x = (ok && vx > 0) ? v.x : 0.f; v.y = (ok && vx > 1) ? v.y : 0.f;
      v.z = (ok && vx > 2) ? v.z : 0.f; v.w = (ok && vx > 3) ? v.w : 0.f;
      if (q < HP * 8) *reinterpret_cast<f32x4*>(xs + (q >> 3) * 32 + c4t) = v;
    }
#pragma unroll
    for (int s_ = 0; s_ < NG; ++s_) {
      const int q = tid + 256 * s_;
      f32x4 v = rg[s_];
      const bool ok = (mskg >> s_) & 1u;
      v.x = (ok && vg > 0) ? v.x : 0.f; v.y = (ok && vg > 1) ? v.y : 0.f;
      v.z = (ok && vg > 2) ? v.z : 0.f; v.w = (ok && vg > 3) ? v.w : 0.f;
      *reinterpret_cast<f32x4*>(gs + (q >> 3) * 32 + c4t) = v;
    }
    __syncthreads();
    if (tile + 1 < t1 && !(dbg & 4)) { HCF_WG_LOAD(tile + 1) HCF_WG_LOAD_G(tile + 1) }
    if (dbg & 1) continue;
    // ---- K loop: 8 rows x 8 groups of 4 pixels; A[m = l16][k = kk] = X[pixel + tap][16 qi + l16], B = G[pixel][16 qj + l16]
#pragma unroll 1
    for (int row = 0; row < TH; ++row) {
#pragma unroll 4
      for (int g = 0; g < TW / 4; ++g) {
        const int xx = 4 * g + kk;
        const float bg = gs[(row * TW + xx) * 32 + qj * 16 + l16];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int dy = (TAPS == 9) ? t / 3 : 0, dx = (TAPS == 9) ? t % 3 : 0;
          const float ax = xs[((row + dy) * HW + xx + dx) * 32 + qi * 16 + l16];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bg, acc[t], 0, 0, 0);
        }
      }
    }
  }
#undef HCF_WG_LOAD
#undef HCF_WG_LOAD_G

  // ---- cross-wave reduction through LDS; the block's partial dW tile goes to scratch (coalesced), a second kernel
  // sums the blocks that share a tile in a fixed order (deterministic; same-address atomics were 10-25x slower)
  if (dbg & 2) return;
  // ---- the block's partial dW tile goes to scratch ([tap][m][n], 64-byte runs); a second kernel sums the blocks that
  // share a tile in a fixed order (deterministic; same-address atomics were 10-25x slower)
  float* part = a.part + ((size_t)((size_t)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (TAPS * 1024);
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)                       // D[row = 4 kk + r][col = l16]
      part[t * 1024 + (qi * 16 + 4 * kk + r) * 32 + qj * 16 + l16] = acc[t][r];
}

// ---------------------------------------------------------------------------------------------------------------------
// The same weight gradient on the f16 matrix cores (f16x3 precision mode, training): both operands are dynamic, so both are
// split into f16 hi / lo in the staging pass (X as is -- the taped forward pass already range-checked it in the same split;
// G multiplied by a power of two that puts its max |g| (a.g_max, written by the epilogue backward) at 2^13..2^14, undone when
// the partial tile is stored) and every product is x_hi g_hi + (x_lo 2^11)(g_hi 2^-11) + x_hi g_lo in
// v_mfma_f32_32x32x16_f16 (fp32 accumulate): per tap a [32 ic] x [32 oc] x [16 pixels] MFMA instead of four 16x16x4 fp32 ones, 16x the matrix rate.
// K = pixels, but the tensors are channel-major (NHWC): the LDS images stay [pixel][32 channels] f16 (64 B per pixel, hi and
// lo planes) and the fragments are fetched with the gfx950 transpose read ds_read_b64_tr_b16 -- a 16-lane group reads a
// [4 pixels][16 channels] block (lane i: pixel i / 4, channels 4 (i % 4) .. + 3) and lane c receives the four pixels of
// channel c, i.e. exactly four consecutive K values of an MFMA operand row; two reads give the eight a lane needs. A tap
// only moves the pixel index, so all 9 taps read the same halo image at different (immediate) offsets, and 32 lanes of one
// read cover 256 contiguous bytes (conflict-free).
// Block = 8 waves (one image row of the 8 x 32 tile each, all 9 taps: 144 accumulator registers), one block per CU: two 76 KB
// LDS buffers in the default form (Wg16::LDS2_BYTES), one 92 KB buffer in the single-buffer form; the eight per-wave partial tiles are added in a fixed tree through LDS, then the same scratch / reduce path.
typedef short v4i16 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16* lds_v4i16;

// eight consecutive K (pixel) values of this lane's channel: pixels +0..3 and +4..7 (4 x 64 B further)
__device__ __forceinline__ f16x8 tr8(const char* p) {
  const v4i16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16)(p));
  const v4i16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16)(p + 256));
  const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(f16x8, r);
}
// f16 hi / lo split of four fp32 values: hi = RNE pairs; lo = f16((v - hi) * 2^11) in one mixed-precision FMA per value
// (fma(hi, -2^11, v * 2^11): hi read as f16, the rest fp32). The 2^11 keeps lo a NORMAL f16 whenever |v| >= 6e-5 -- with an
// unscaled lo, activations below ~0.1 lose their low half to f16 denormals (3e-8 absolute) -- and is undone on the other
// operand: the x_lo term multiplies by g_hi * 2^-11 (third G plane).
__device__ __forceinline__ void split_hl_x(const f32x4 v, u32x2& h, u32x2& l) {
  const f16x2 h01 = {(_Float16)v.x, (_Float16)v.y}, h23 = {(_Float16)v.z, (_Float16)v.w};
  const uint32_t H0 = __builtin_bit_cast(uint32_t, h01), H1 = __builtin_bit_cast(uint32_t, h23);
  const float k = -2048.f;
  const f32x4 v2 = v * 2048.f;
  uint32_t L0, L1;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(L0) : "v"(H0), "v"(k), "v"(v2.x));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L0) : "v"(H0), "v"(k), "v"(v2.y));
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(L1) : "v"(H1), "v"(k), "v"(v2.z));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L1) : "v"(H1), "v"(k), "v"(v2.w));
  h = u32x2{H0, H1};
  l = u32x2{L0, L1};
}
// G (already scaled to max ~2^14): hi, lo = f16(v - hi), and hs = f16(hi * 2^-11) for the x_lo term
__device__ __forceinline__ void split_hl_g(const f32x4 v, u32x2& h, u32x2& l, u32x2& hs) {
  const f16x2 h01 = {(_Float16)v.x, (_Float16)v.y}, h23 = {(_Float16)v.z, (_Float16)v.w};
  const uint32_t H0 = __builtin_bit_cast(uint32_t, h01), H1 = __builtin_bit_cast(uint32_t, h23);
  const float k = 1.f / 2048.f;
  uint32_t L0, L1, S0, S1;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(L0) : "v"(H0), "v"(v.x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L0) : "v"(H0), "v"(v.y));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(L1) : "v"(H1), "v"(v.z));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L1) : "v"(H1), "v"(v.w));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(S0) : "v"(H0), "v"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(S0) : "v"(H0), "v"(k));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(S1) : "v"(H1), "v"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(S1) : "v"(H1), "v"(k));
  h = u32x2{H0, H1};
  l = u32x2{L0, L1};
  hs = u32x2{S0, S1};
}

// G split for the double-buffered form: hi and lo only
__device__ __forceinline__ void split_hl_g2(const f32x4 v, u32x2& h, u32x2& l) {
  const f16x2 h01 = {(_Float16)v.x, (_Float16)v.y}, h23 = {(_Float16)v.z, (_Float16)v.w};
  const uint32_t H0 = __builtin_bit_cast(uint32_t, h01), H1 = __builtin_bit_cast(uint32_t, h23);
  uint32_t L0, L1;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(L0) : "v"(H0), "v"(v.x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L0) : "v"(H0), "v"(v.y));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(L1) : "v"(H1), "v"(v.z));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L1) : "v"(H1), "v"(v.w));
  h = u32x2{H0, H1};
  l = u32x2{L0, L1};
}

template <int TAPS> struct Wg16 {
  static constexpr int PAD = (TAPS == 9) ? 1 : 0;
  static constexpr int HH = TH + 2 * PAD, HW = TW + 2 * PAD, HP = HH * HW;
  static constexpr int XB = HP * 64, GB = TH * TW * 64;          // bytes of one f16 plane of the X halo / the G tile
  static constexpr int LDS_BYTES = 2 * XB + 3 * GB;              // 92 672 (3x3) / 81 920 (1x1): X hi, lo'; G hi, lo, hi * 2^-11
  // double-buffered form (the default): two buffers of {X hi, X lo', G hi, G lo}; g_hi * 2^-11 is formed in registers from the
  // g_hi fragment (four v_pk_mul_f16 per 16 pixels), so its plane is gone and two buffers fit one CU's 160 KB
  static constexpr int BUF_BYTES = 2 * XB + 2 * GB;              // 76 288 (3x3) / 65 536 (1x1)
  static constexpr int LDS2_BYTES = 2 * BUF_BYTES;               // 152 576 (3x3) / 131 072 (1x1)
};

// (bxi, byi, bzi) = (pixel slice, input-channel block, output-channel block) in a grid of (., gdy, gdz): the block coordinates of
// the one-conv launch, or decoded from a linear block number by the batched launch
template <int TAPS, bool VEC, int DB>
__device__ __forceinline__ void wgrad_f16x3_body(const WgradArgs& a, const int bxi, const int byi, const int bzi, const int gdy, const int gdz) {
  typedef Wg16<TAPS> C;
  constexpr int PAD = C::PAD, HW = C::HW, HP = C::HP, XB = C::XB, GB = C::GB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const xh = lds;
  char* const gh = lds + 2 * XB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W;
  const int sw = a.strip_w, svw = a.B * sw;      // strips (WgradArgs::strip_w): virtual row width B (W + 1)
  const unsigned smagic = a.strip_magic;
  const int tiles_x = sw ? (svw + TW - 1) / TW : (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int ntiles = sw ? tiles_x * tiles_y : a.B * tiles_x * tiles_y;

  int blk = byi, si = 0;
  for (; si < a.nsrc; ++si) {
    const int nb = (a.src[si].n + 31) >> 5;
    if (blk < nb) break;
    blk -= nb;
  }
  const View sv = a.src[si];
  const int ic0 = blk * 32;
  const int icn = min(32, sv.n - ic0);
  const int oc0 = bzi * 32;
  const int ocn = min(32, a.g.n - oc0);
  const int up = sv.up, Hs = H >> up, Ws = W >> up;

  float g_s = 1.f, g_inv = 1.f;                  // G * g_s has its max |g| in [2^13, 2^14)
  {
    const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.g_max)));
    if (mx > 0.f && mx < 3.0e38f) {
      int ex = 0;
      (void)frexpf(mx, &ex);
      g_s = ldexpf(1.f, 14 - ex);
      g_inv = ldexpf(1.f, ex - 14);
    }
  }

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // register staging of the next tile, branch-free (see conv_wgrad_kernel): clamped addresses, masks applied at the LDS write
  constexpr int NX = (HP * 8 + 511) / 512, NG = (TH * TW * 8) / 512;
  f32x4 rx[NX], rg[NG];
  unsigned mskx = 0, mskg = 0;
  const int c4t = (tid & 7) * 4;
  const int vx = max(0, min(4, icn - c4t)), vg = max(0, min(4, ocn - c4t));
  const int c4x = min(c4t, max(0, (icn - 1) & ~3)), c4g = min(c4t, max(0, (ocn - 1) & ~3));
  const float* const xbase = sv.p + sv.c0 + ic0;
  const float* const gbase = a.g.p + a.g.c0 + oc0;
#define HCF_WG_LOAD(TILE)                                                                                     \
  {                                                                                                           \
    const int txb_ = (TILE) % tiles_x, tyb_ = ((TILE) / tiles_x) % tiles_y, bt_ = sw ? 0 : (TILE) / (tiles_x * tiles_y); \
    const int x0_ = txb_ * TW, y0_ = tyb_ * TH;                                                               \
    mskx = 0;                                                                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < NX; ++s_) {                                                       \
      const int hp = min((tid + 512 * s_) >> 3, HP - 1);                                                      \
      const int hy = hp / HW, hx = hp - hy * HW;                                                              \
      const int y = y0_ + hy - PAD;                                                                           \
      int x = x0_ + hx - PAD, b_ = bt_;                                                                       \
      bool okx_ = x >= 0 && x < W;                                                                            \
      if (sw) {                      /* virtual column -> (image, column); column W is the zero separator */ \
        const int vc_ = min(max(x, 0), svw - 1);                                                              \
        b_ = (int)__umulhi((unsigned)vc_, smagic);                                                            \
        okx_ = x >= 0 && x < svw && (vc_ - b_ * sw) < W;                                                      \
        x = vc_ - b_ * sw;                                                                                    \
      }                                                                                                       \
      mskx |= (y >= 0 && y < H && okx_) ? (1u << s_) : 0u;                                                    \
      const int yc = min(max(y, 0), H - 1) >> up, xc = min(max(x, 0), W - 1) >> up;                           \
      const float* p = xbase + ((size_t)((size_t)b_ * Hs + yc) * Ws + xc) * sv.cs;                            \
      f32x4 v;                                                                                                \
      if (VEC) {                                                                                              \
        v = *reinterpret_cast<const f32x4*>(p + c4x);                                                         \
      } else {                                                                                                \
        v.x = p[min(c4t, icn - 1)]; v.y = p[min(c4t + 1, icn - 1)];                                           \
        v.z = p[min(c4t + 2, icn - 1)]; v.w = p[min(c4t + 3, icn - 1)];                                       \
      }                                                                                                       \
      rx[s_] = v;                                                                                             \
    }                                                                                                         \
    mskg = 0;                                                                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < NG; ++s_) {                                                       \
      const int px = (tid + 512 * s_) >> 3;                                                                   \
      const int y = y0_ + (px >> 5);                                                                          \
      int x = x0_ + (px & 31), b_ = bt_;                                                                      \
      bool okx_ = x < W;                                                                                      \
      if (sw) {                                                                                               \
        const int vc_ = min(x, svw - 1);                                                                      \
        b_ = (int)__umulhi((unsigned)vc_, smagic);                                                            \
        okx_ = x < svw && (vc_ - b_ * sw) < W;                                                                \
        x = vc_ - b_ * sw;                                                                                    \
      }                                                                                                       \
      mskg |= (y < H && okx_) ? (1u << s_) : 0u;                                                              \
      const float* p = gbase + ((size_t)((size_t)b_ * H + min(y, H - 1)) * W + min(x, W - 1)) * a.g.cs;       \
      f32x4 v;                                                                                                \
      if (VEC) {                                                                                              \
        v = *reinterpret_cast<const f32x4*>(p + c4g);                                                         \
      } else {                                                                                                \
        v.x = p[min(c4t, ocn - 1)]; v.y = p[min(c4t + 1, ocn - 1)];                                           \
        v.z = p[min(c4t + 2, ocn - 1)]; v.w = p[min(c4t + 3, ocn - 1)];                                       \
      }                                                                                                       \
      rg[s_] = v;                                                                                             \
    }                                                                                                         \
  }
  // transpose-read lane offset: pixel (i / 4) + 8 (lane / 32), channels 16 ((lane / 16) & 1) + 4 (i % 4), i = lane % 16
  const int lofs = (((lane & 15) >> 2) + 8 * (lane >> 5)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const char* const xw = xh + lofs + wave * (HW * 64);           // this wave's image row (row + dy through immediates)
  const char* const gw = gh + lofs + wave * (TW * 64);

  const int t0 = bxi * a.tpb, t1 = min(ntiles, t0 + a.tpb);
  if constexpr (DB == 2) {
    // ---- double-buffered, interleaved form (the default): tile i's MFMAs read LDS buffer i & 1 while the staging registers of
    // tile i + 1 (loaded a whole tile earlier) are converted and written to the OTHER buffer ONE SLOT PER TAP (the five slots of
    // a 3x3 tile ride on the first five taps), each slot's registers then reloaded with tile i + 2: the convert / ds_write /
    // address / global_load stream of a slot (~70 VALU instructions) issues in the shadow of the taps' MFMAs (96 matrix-pipe
    // cycles per tap and wave), one barrier per tile. The single-buffer form runs barrier, convert + write (all 16 waves of the
    // CU at once, matrix pipe idle), barrier, MFMAs. Same MFMAs, same fragments, same order: bit-identical results.
    constexpr int BUFB = C::BUF_BYTES;
    // staging slots of this form: FOUR threads per pixel, 8 channels (two 16-byte loads behind ONE address computation, one
    // ds_write_b128 per plane) -- half the slots, i.e. half the per-slot address / mask arithmetic of the 8-threads-per-pixel
    // mapping of the other forms; the same registers (rx[2 s], rx[2 s + 1]), the same values at the same LDS addresses
    constexpr int NX2 = (HP * 4 + 511) / 512, NG2 = (TH * TW * 4) / 512;
    static_assert(2 * NX2 <= NX && 2 * NG2 <= NG, "the slot pairs fit the staging registers");
    constexpr int NS = NX2 + NG2, SPP = (NS + 2 * TAPS - 1) / (2 * TAPS);    // staging slots, slots per (step, tap) position
    const int c8t = (tid & 3) * 8;
    const int vxa = max(0, min(4, icn - c8t)), vxb = max(0, min(4, icn - c8t - 4));
    const int vga = max(0, min(4, ocn - c8t)), vgb = max(0, min(4, ocn - c8t - 4));
    const int c8xa = min(c8t, max(0, (icn - 1) & ~3)), c8xb = min(c8t + 4, max(0, (icn - 1) & ~3));
    const int c8ga = min(c8t, max(0, (ocn - 1) & ~3)), c8gb = min(c8t + 4, max(0, (ocn - 1) & ~3));
    int x0n = 0, y0n = 0, btn = 0;                     // tile whose loads are being issued
#define HCF_WG_COORDS(TILE)                                                                                   \
    {                                                                                                         \
      const int txb_ = (TILE) % tiles_x, tyb_ = ((TILE) / tiles_x) % tiles_y;                                 \
      btn = sw ? 0 : (TILE) / (tiles_x * tiles_y);                                                            \
      x0n = txb_ * TW; y0n = tyb_ * TH;                                                                       \
    }
#define HCF_WG_LOAD_SLOT(SL)                                                                                          \
    if ((SL) < NX2) {                                                                                                 \
      const int s_ = (SL);                                                                                            \
      const int hp = min((tidl_ + 512 * s_) >> 2, HP - 1);                                                            \
      const int hy = hp / HW, hxx_ = hp - hy * HW;                                                                    \
      const int y = y0n + hy - PAD;                                                                                   \
      int x = x0n + hxx_ - PAD, b_ = btn;                                                                             \
      /* strips: virtual column -> (image, column), branch-free (sw is block-uniform: selects, no basic-block split) */ \
      const int vc_ = min(max(x, 0), svw - 1);                                                                        \
      const int bq_ = (int)__umulhi((unsigned)vc_, smagic);                                                           \
      const int xr_ = vc_ - bq_ * sw;                                                                                 \
      const bool okx_ = sw ? (x >= 0 && x < svw && xr_ < W) : (x >= 0 && x < W);                                      \
      b_ = sw ? bq_ : b_;                                                                                             \
      x = sw ? xr_ : x;                                                                                               \
      mskx = (mskx & ~(1u << s_)) | ((y >= 0 && y < H && okx_) ? (1u << s_) : 0u);                                    \
      const int yc = min(max(y, 0), H - 1) >> up, xc = min(max(x, 0), W - 1) >> up;                                   \
      const float* p = xbase + ((size_t)((size_t)b_ * Hs + yc) * Ws + xc) * sv.cs;                                    \
      f32x4 va, vb;                                                                                                   \
      if (VEC) {                                                                                                      \
        va = *reinterpret_cast<const f32x4*>(p + c8xa);                                                               \
        vb = *reinterpret_cast<const f32x4*>(p + c8xb);                                                               \
      } else {                                                                                                        \
        va.x = p[min(c8t, icn - 1)]; va.y = p[min(c8t + 1, icn - 1)];                                                 \
        va.z = p[min(c8t + 2, icn - 1)]; va.w = p[min(c8t + 3, icn - 1)];                                             \
        vb.x = p[min(c8t + 4, icn - 1)]; vb.y = p[min(c8t + 5, icn - 1)];                                             \
        vb.z = p[min(c8t + 6, icn - 1)]; vb.w = p[min(c8t + 7, icn - 1)];                                             \
      }                                                                                                               \
      rx[2 * s_] = va;                                                                                                \
      rx[2 * s_ + 1] = vb;                                                                                            \
    } else if ((SL) < NS) {                                                                                           \
      const int s_ = (SL) - NX2;                                                                                      \
      const int px = (tidl_ + 512 * s_) >> 2;                                                                         \
      const int y = y0n + (px >> 5);                                                                                  \
      int x = x0n + (px & 31), b_ = btn;                                                                              \
      const int vc_ = min(x, svw - 1);                                                                                \
      const int bq_ = (int)__umulhi((unsigned)vc_, smagic);                                                           \
      const int xr_ = vc_ - bq_ * sw;                                                                                 \
      const bool okx_ = sw ? (x < svw && xr_ < W) : (x < W);                                                          \
      b_ = sw ? bq_ : b_;                                                                                             \
      x = sw ? xr_ : x;                                                                                               \
      mskg = (mskg & ~(1u << s_)) | ((y < H && okx_) ? (1u << s_) : 0u);                                              \
      const float* p = gbase + ((size_t)((size_t)b_ * H + min(y, H - 1)) * W + min(x, W - 1)) * a.g.cs;               \
      f32x4 va, vb;                                                                                                   \
      if (VEC) {                                                                                                      \
        va = *reinterpret_cast<const f32x4*>(p + c8ga);                                                               \
        vb = *reinterpret_cast<const f32x4*>(p + c8gb);                                                               \
      } else {                                                                                                        \
        va.x = p[min(c8t, ocn - 1)]; va.y = p[min(c8t + 1, ocn - 1)];                                                 \
        va.z = p[min(c8t + 2, ocn - 1)]; va.w = p[min(c8t + 3, ocn - 1)];                                             \
        vb.x = p[min(c8t + 4, ocn - 1)]; vb.y = p[min(c8t + 5, ocn - 1)];                                             \
        vb.z = p[min(c8t + 6, ocn - 1)]; vb.w = p[min(c8t + 7, ocn - 1)];                                             \
      }                                                                                                               \
      rg[2 * s_] = va;                                                                                                \
      rg[2 * s_ + 1] = vb;                                                                                            \
    }
#define HCF_WG_STORE_SLOT(SL, BUF)                                                                                    \
    if ((SL) < NX2) {                                                                                                 \
      const int s_ = (SL);                                                                                            \
      char* const xb_ = lds + (BUF) * BUFB;                                                                           \
      f32x4 va = rx[2 * s_], vb = rx[2 * s_ + 1];                                                                     \
      const bool ok = (mskx >> s_) & 1u;                                                                              \
      va.x = (ok && vxa > 0) ? va.x : 0.f; va.y = (ok && vxa > 1) ? va.y : 0.f;                                       \
      va.z = (ok && vxa > 2) ? va.z : 0.f; va.w = (ok && vxa > 3) ? va.w : 0.f;                                       \
      vb.x = (ok && vxb > 0) ? vb.x : 0.f; vb.y = (ok && vxb > 1) ? vb.y : 0.f;                                       \
      vb.z = (ok && vxb > 2) ? vb.z : 0.f; vb.w = (ok && vxb > 3) ? vb.w : 0.f;                                       \
      u32x2 ha, la, hb, lb;                                                                                           \
      split_hl_x(va, ha, la);                                                                                         \
      split_hl_x(vb, hb, lb);                                                                                         \
      /* threads past the halo's last pixel hold pixel HP - 1's data (their load was clamped to it): they write the same values */ \
      /* to the same address as its owner instead of branching around the store */                                    \
      const int qp_ = min((tidl_ + 512 * s_) >> 2, HP - 1);                                                           \
      *reinterpret_cast<u32x4*>(xb_ + qp_ * 64 + c8t * 2) = u32x4{ha.x, ha.y, hb.x, hb.y};                            \
      *reinterpret_cast<u32x4*>(xb_ + XB + qp_ * 64 + c8t * 2) = u32x4{la.x, la.y, lb.x, lb.y};                       \
    } else if ((SL) < NS) {                                                                                           \
      const int s_ = (SL) - NX2;                                                                                      \
      char* const gb_ = lds + (BUF) * BUFB + 2 * XB;                                                                  \
      f32x4 va = rg[2 * s_] * g_s, vb = rg[2 * s_ + 1] * g_s;                                                         \
      const bool ok = (mskg >> s_) & 1u;                                                                              \
      va.x = (ok && vga > 0) ? va.x : 0.f; va.y = (ok && vga > 1) ? va.y : 0.f;                                       \
      va.z = (ok && vga > 2) ? va.z : 0.f; va.w = (ok && vga > 3) ? va.w : 0.f;                                       \
      vb.x = (ok && vgb > 0) ? vb.x : 0.f; vb.y = (ok && vgb > 1) ? vb.y : 0.f;                                       \
      vb.z = (ok && vgb > 2) ? vb.z : 0.f; vb.w = (ok && vgb > 3) ? vb.w : 0.f;                                       \
      u32x2 ha, la, hb, lb;                                                                                           \
      split_hl_g2(va, ha, la);                                                                                        \
      split_hl_g2(vb, hb, lb);                                                                                        \
      const int qp_ = (tidl_ + 512 * s_) >> 2;                                                                        \
      *reinterpret_cast<u32x4*>(gb_ + qp_ * 64 + c8t * 2) = u32x4{ha.x, ha.y, hb.x, hb.y};                            \
      *reinterpret_cast<u32x4*>(gb_ + GB + qp_ * 64 + c8t * 2) = u32x4{la.x, la.y, lb.x, lb.y};                       \
    }
    // one tile: ST = tile + 1 exists (its registers go to the other buffer), LD = tile + 2 exists (its loads are issued)
#define HCF_WG_TILE_BODY(ST, LD)                                                                              \
    {                                                                                                         \
      const int cur = (tile - t0) & 1;                                                                        \
      int tidl_ = tid;               /* opaque copy: the slots' pixel coordinates are recomputed per tile (hoisted out of */ \
      asm volatile("" : "+v"(tidl_));   /* the loop they cost ~20 registers, i.e. spills beside 144 accumulators) */  \
      const char* const xwc = xw2 + cur * BUFB;                                                               \
      const char* const gwc = gw2 + cur * BUFB;                                                               \
      if (LD) HCF_WG_COORDS(tile + 2)                                                                         \
      _Pragma("unroll") for (int hx = 0; hx < 2; ++hx) {                                                      \
        const f16x8 bh = tr8(gwc + hx * (16 * 64)), bl = tr8(gwc + GB + hx * (16 * 64));                      \
        const f16x8 bs = bh * (_Float16)0.00048828125f;        /* g_hi 2^-11 (exact up to f16 denormals) */    \
        _Pragma("unroll") for (int t = 0; t < TAPS; ++t) {                                                    \
          const int dy = (TAPS == 9) ? t / 3 : 0, dx = (TAPS == 9) ? t % 3 : 0;                               \
          const int off = (dy * HW + 16 * hx + dx) * 64;                                                      \
          const f16x8 ah = tr8(xwc + off), al = tr8(xwc + XB + off);                                          \
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);                           \
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bs, acc[t], 0, 0, 0);                           \
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);                           \
          if (ST) {                                                                                           \
            _Pragma("unroll") for (int j_ = 0; j_ < SPP; ++j_) {                                              \
              const int sl_ = (hx * TAPS + t) * SPP + j_;       /* a constant after unrolling */               \
              HCF_WG_STORE_SLOT(sl_, cur ^ 1)                                                                 \
              if (LD) { HCF_WG_LOAD_SLOT(sl_) }                                                               \
            }                                                                                                 \
            /* pin the slot to its tap (only MFMAs and LDS reads may cross): left alone, the scheduler hoists every slot's    */ \
            /* conversion to the top of the tile -- one wait for ALL the staging loads, then the VALU stream, then the MFMAs */ \
            __builtin_amdgcn_sched_barrier(0x108);                                                            \
          }                                                                                                   \
        }                                                                                                     \
      }                                                                                                       \
      __syncthreads();                            /* buffer cur read by every wave, buffer cur ^ 1 complete */ \
    }
    const char* const xw2 = lds + lofs + wave * (HW * 64);
    const char* const gw2 = lds + 2 * XB + lofs + wave * (TW * 64);
    if (t0 < t1) {                                 // prologue: tile t0 into buffer 0, tile t0 + 1 into the registers
      const int tidl_ = tid;
      HCF_WG_COORDS(t0)
      _Pragma("unroll") for (int sl_ = 0; sl_ < NS; ++sl_) { HCF_WG_LOAD_SLOT(sl_) }
      _Pragma("unroll") for (int sl_ = 0; sl_ < NS; ++sl_) { HCF_WG_STORE_SLOT(sl_, 0) }
      if (t0 + 1 < t1) {
        HCF_WG_COORDS(t0 + 1)
        _Pragma("unroll") for (int sl_ = 0; sl_ < NS; ++sl_) { HCF_WG_LOAD_SLOT(sl_) }
      }
    }
    __syncthreads();
    int tile = t0;
    for (; tile + 2 < t1; ++tile) HCF_WG_TILE_BODY(1, 1)
    if (tile + 1 < t1) {
      HCF_WG_TILE_BODY(1, 0)
      ++tile;
    }
    if (tile < t1) HCF_WG_TILE_BODY(0, 0)
#undef HCF_WG_TILE_BODY
#undef HCF_WG_STORE_SLOT
#undef HCF_WG_LOAD_SLOT
#undef HCF_WG_COORDS
  } else if constexpr (DB == 1) {
    constexpr int BUFB = C::BUF_BYTES;
#define HCF_WG_STORE(BUF)                                                                                     \
    {                                                                                                         \
      char* const xb_ = lds + (BUF) * BUFB;                                                                   \
      char* const gb_ = xb_ + 2 * XB;                                                                         \
      _Pragma("unroll") for (int s_ = 0; s_ < NX; ++s_) {                                                     \
        const int q = tid + 512 * s_;                                                                         \
        f32x4 v = rx[s_];                                                                                     \
        const bool ok = (mskx >> s_) & 1u;                                                                    \
        v.x = (ok && vx > 0) ? v.x : 0.f; v.y = (ok && vx > 1) ? v.y : 0.f;                                   \
        v.z = (ok && vx > 2) ? v.z : 0.f; v.w = (ok && vx > 3) ? v.w : 0.f;                                   \
        u32x2 h, l;                                                                                           \
        split_hl_x(v, h, l);                                                                                  \
        if (q < HP * 8) {                                                                                     \
          *reinterpret_cast<u32x2*>(xb_ + (q >> 3) * 64 + c4t * 2) = h;                                       \
          *reinterpret_cast<u32x2*>(xb_ + XB + (q >> 3) * 64 + c4t * 2) = l;                                  \
        }                                                                                                     \
      }                                                                                                       \
      _Pragma("unroll") for (int s_ = 0; s_ < NG; ++s_) {                                                     \
        const int q = tid + 512 * s_;                                                                         \
        f32x4 v = rg[s_] * g_s;                                                                               \
        const bool ok = (mskg >> s_) & 1u;                                                                    \
        v.x = (ok && vg > 0) ? v.x : 0.f; v.y = (ok && vg > 1) ? v.y : 0.f;                                   \
        v.z = (ok && vg > 2) ? v.z : 0.f; v.w = (ok && vg > 3) ? v.w : 0.f;                                   \
        u32x2 h, l;                                                                                           \
        split_hl_g2(v, h, l);                                                                                 \
        *reinterpret_cast<u32x2*>(gb_ + (q >> 3) * 64 + c4t * 2) = h;                                         \
        *reinterpret_cast<u32x2*>(gb_ + GB + (q >> 3) * 64 + c4t * 2) = l;                                    \
      }                                                                                                       \
    }
#define HCF_WG_STEP(HX)                                                                                       \
    {                                                                                                         \
      const f16x8 bh = tr8(gwc + (HX) * (16 * 64)), bl = tr8(gwc + GB + (HX) * (16 * 64));                    \
      const f16x8 bs = bh * (_Float16)0.00048828125f;          /* g_hi 2^-11 (exact up to f16 denormals) */    \
      _Pragma("unroll") for (int t = 0; t < TAPS; ++t) {                                                      \
        const int dy = (TAPS == 9) ? t / 3 : 0, dx = (TAPS == 9) ? t % 3 : 0;                                 \
        const int off = (dy * HW + 16 * (HX) + dx) * 64;                                                      \
        const f16x8 ah = tr8(xwc + off), al = tr8(xwc + XB + off);                                            \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);                             \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bs, acc[t], 0, 0, 0);                             \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);                             \
      }                                                                                                       \
    }
    const char* const xw2 = lds + lofs + wave * (HW * 64);
    const char* const gw2 = lds + 2 * XB + lofs + wave * (TW * 64);
    if (t0 < t1) {
      HCF_WG_LOAD(t0)
      HCF_WG_STORE(0)
      if (t0 + 1 < t1) HCF_WG_LOAD(t0 + 1)
    }
    __syncthreads();
    for (int tile = t0; tile < t1; ++tile) {
      const int cur = (tile - t0) & 1;
      const char* const xwc = xw2 + cur * BUFB;
      const char* const gwc = gw2 + cur * BUFB;
      HCF_WG_STEP(0)
      if (tile + 1 < t1) HCF_WG_STORE(cur ^ 1)
      if (tile + 2 < t1) HCF_WG_LOAD(tile + 2)
      HCF_WG_STEP(1)
      __syncthreads();                            // buffer cur has been read by every wave, buffer cur ^ 1 is complete
    }
#undef HCF_WG_STEP
#undef HCF_WG_STORE
  } else {
  if (t0 < t1) HCF_WG_LOAD(t0)
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();                              // the previous tile's fragments have been read
#pragma unroll
    for (int s_ = 0; s_ < NX; ++s_) {
      const int q = tid + 512 * s_;
      f32x4 v = rx[s_];
      const bool ok = (mskx >> s_) & 1u;
      v.x = (ok && vx > 0) ? v.x : 0.f; v.y = (ok && vx > 1) ? v.y : 0.f;
      v.z = (ok && vx > 2) ? v.z : 0.f; v.w = (ok && vx > 3) ? v.w : 0.f;
      u32x2 h, l;
      split_hl_x(v, h, l);
      if (q < HP * 8) {
        *reinterpret_cast<u32x2*>(xh + (q >> 3) * 64 + c4t * 2) = h;
        *reinterpret_cast<u32x2*>(xh + XB + (q >> 3) * 64 + c4t * 2) = l;
      }
    }
#pragma unroll
    for (int s_ = 0; s_ < NG; ++s_) {
      const int q = tid + 512 * s_;
      f32x4 v = rg[s_] * g_s;
      const bool ok = (mskg >> s_) & 1u;
      v.x = (ok && vg > 0) ? v.x : 0.f; v.y = (ok && vg > 1) ? v.y : 0.f;
      v.z = (ok && vg > 2) ? v.z : 0.f; v.w = (ok && vg > 3) ? v.w : 0.f;
      u32x2 h, l, hs;
      split_hl_g(v, h, l, hs);
      *reinterpret_cast<u32x2*>(gh + (q >> 3) * 64 + c4t * 2) = h;
      *reinterpret_cast<u32x2*>(gh + GB + (q >> 3) * 64 + c4t * 2) = l;
      *reinterpret_cast<u32x2*>(gh + 2 * GB + (q >> 3) * 64 + c4t * 2) = hs;
    }
    __syncthreads();
    if (tile + 1 < t1) HCF_WG_LOAD(tile + 1)
    // ---- K loop of this wave: its image row, two steps of 16 pixels, all taps
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
      const f16x8 bh = tr8(gw + hx * (16 * 64)), bl = tr8(gw + GB + hx * (16 * 64)), bs = tr8(gw + 2 * GB + hx * (16 * 64));
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dy = (TAPS == 9) ? t / 3 : 0, dx = (TAPS == 9) ? t % 3 : 0;
        const int off = (dy * HW + 16 * hx + dx) * 64;
        const f16x8 ah = tr8(xw + off), al = tr8(xw + XB + off);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bs, acc[t], 0, 0, 0);      // (x_lo 2^11) (g_hi 2^-11)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
      }
    }
  }
  }
#undef HCF_WG_LOAD

  // ---- fixed-order tree over the eight waves through LDS (two 36 KB slots): ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7))
  float* const red = reinterpret_cast<float*>(lds);
#define HCF_WG_ROUND(WR_LO, ADD_LO)      /* waves WR_LO, WR_LO + 1 publish; waves ADD_LO, ADD_LO + 1 accumulate */ \
  __syncthreads();                                                                                            \
  if (wave == (WR_LO) || wave == (WR_LO) + 1) {                                                               \
    _Pragma("unroll") for (int t = 0; t < TAPS; ++t)                                                          \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) red[(((wave - (WR_LO)) * TAPS + t) * 16 + r) * 64 + lane] = acc[t][r]; \
  }                                                                                                           \
  __syncthreads();                                                                                            \
  if (wave == (ADD_LO) || wave == (ADD_LO) + 1) {                                                             \
    _Pragma("unroll") for (int t = 0; t < TAPS; ++t)                                                          \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[t][r] += red[(((wave - (ADD_LO)) * TAPS + t) * 16 + r) * 64 + lane]; \
  }
  HCF_WG_ROUND(6, 2)
  HCF_WG_ROUND(4, 0)
  HCF_WG_ROUND(2, 0)
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
#undef HCF_WG_ROUND
  if (wave != 0) return;
  float* part = a.part + ((size_t)((size_t)bxi * gdy + byi) * gdz + bzi) * (TAPS * 1024);
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {                // D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31]
      const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      part[t * 1024 + m * 32 + (lane & 31)] = (acc[t][r] + red[(t * 16 + r) * 64 + lane]) * g_inv;
    }
}


template <int TAPS, bool VEC, int DB>
__global__ __launch_bounds__(512, 1) void conv_wgrad_f16x3_kernel(const WgradArgs a) {
  wgrad_f16x3_body<TAPS, VEC, DB>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, gridDim.z);
}

// Several convs' weight gradients as ONE launch (the five convs of a dense block: hcf_engine_train.inc). A one-conv launch is at
// most one round of one-per-CU blocks, i.e. 2-3 tiles per block behind a 13 us fixed part (prologue + the eight-wave reduce tree);
// here the same block budget covers all the convs, every block walks 10+ tiles, and the fixed part is paid once per block instead
// of once per conv and block: 234 blocks x 68 us for a dense block at 16 x 40 x 40 against 5 x (160 blocks x 22-27 us).
template <bool VEC, int DB>
__global__ __launch_bounds__(512, 1) void conv_wgrad_f16x3_batch_kernel(const WgradBatchArgs b) {
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.blk0[j + 1]) ++j;      // block-uniform
  int r = (int)blockIdx.x - b.blk0[j];
  const int nicb = b.nicb[j], nocb = b.nocb[j];
  const int bz = r % nocb; r /= nocb;
  const int by = r % nicb; r /= nicb;
  wgrad_f16x3_body<9, VEC, DB>(b.a[j], r, by, bz, nicb, nocb);
}

// dW[oc][ic][tap] += sum over the nx blocks of part[bx][icb][ocb][tap][m][n]
// (bx, icb, ocb) = this block's coordinates in a grid of (ceil(taps * 1024 / 256), nicb, nocb)
__device__ __forceinline__ void wgrad_reduce_block(const WgradArgs& a, int nx, int taps, int bx, int icb, int ocb, int nicb, int nocb) {
  const int e = bx * 256 + threadIdx.x;          // (tap, m, n)
  if (e >= taps * 1024) return;
  const int t = e >> 10, m = (e >> 5) & 31, n = e & 31;
  // locate the channel block
  int blk = icb, si = 0, ic_base = 0;
  for (; si < a.nsrc; ++si) {
    const int nb = (a.src[si].n + 31) >> 5;
    if (blk < nb) break;
    blk -= nb;
    ic_base += a.src[si].n;
  }
  const int ic = blk * 32 + m, oc = ocb * 32 + n;
  if (ic >= a.src[si].n || oc >= a.g.n) return;
  const size_t stride = (size_t)nicb * nocb * taps * 1024;
  const float* p = a.part + ((size_t)icb * nocb + ocb) * (taps * 1024) + e;
  // fixed summation order (8 interleaved partial sums, then their sum): deterministic, and 8 loads in flight
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= nx; b += 8)
#pragma unroll
    for (int j = 0; j < 8; ++j) s8[j] += p[(size_t)(b + j) * stride];
  for (; b < nx; ++b) s8[b & 7] += p[(size_t)b * stride];
  const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  a.dw[((size_t)oc * a.cin_total + ic_base + ic) * taps + t] += a.negate ? -s : s;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a, int nx, int taps) {
  wgrad_reduce_block(a, nx, taps, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, gridDim.z);
}

// many reduce steps in one launch: block -> job by binary search over the jobs' first-block numbers (blk0, ascending)
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceJob* __restrict__ jobs, int njobs) {
  const long long blk = blockIdx.x;
  int lo = 0, hi = njobs;                       // invariant: jobs[lo].blk0 <= blk < (hi < njobs ? jobs[hi].blk0 : total)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (jobs[mid].blk0 <= blk) lo = mid; else hi = mid;
  }
  const WgradReduceJob& j = jobs[lo];
  int r = (int)(blk - j.blk0);
  const int ocb = r % j.nocb; r /= j.nocb;
  const int icb = r % j.nicb; r /= j.nicb;
  wgrad_reduce_block(j.a, j.nx, j.a.taps, r, icb, ocb, j.nicb, j.nocb);
}

// max |g| over a channel window (float bits ordered like ints for non-negative values); *out zero-initialised by the caller
__global__ __launch_bounds__(256) void absmax_kernel(View g, long long npix, float* out) {
  float mx = 0.f;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256)
    for (int c = 0; c < g.n; ++c) mx = fmaxf(mx, fabsf(g.p[(size_t)p * g.cs + g.c0 + c]));
  __shared__ int shm;
  if (threadIdx.x == 0) shm = 0;
  __syncthreads();
  if (mx > 0.f && mx == mx) atomicMax(&shm, __builtin_bit_cast(int, mx));
  __syncthreads();
  if (threadIdx.x == 0 && shm) atomicMax(reinterpret_cast<int*>(out), shm);
}

// max |B^T d B| over every F(2x2,3x3) input patch (4x4 pixels, stride 2, zero padding 1) of a channel window: the largest value
// the Winograd kernels have to split into f16 hi / lo parts (range probe only, hcf_debug_range_probe)
__global__ __launch_bounds__(256) void wino_vmax_kernel(View g, int B, int H, int W, float* out) {
  const int tw = (W + 1) / 2, th = (H + 1) / 2;
  const long long total = (long long)B * th * tw * g.n;
  float mx = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % g.n);
    long long tix = i / g.n;
    const int tx = (int)(tix % tw);
    tix /= tw;
    const int ty = (int)(tix % th), b = (int)(tix / th);
    float d[4][4], t[4][4];
    for (int r = 0; r < 4; ++r)
      for (int q = 0; q < 4; ++q) {
        const int y = 2 * ty - 1 + r, x = 2 * tx - 1 + q;
        d[r][q] = (y >= 0 && y < H && x >= 0 && x < W) ? g.p[(((size_t)b * H + y) * W + x) * g.cs + g.c0 + c] : 0.f;
      }
    for (int q = 0; q < 4; ++q) {
      t[0][q] = d[0][q] - d[2][q]; t[1][q] = d[1][q] + d[2][q]; t[2][q] = d[2][q] - d[1][q]; t[3][q] = d[1][q] - d[3][q];
    }
    for (int r = 0; r < 4; ++r) {
      mx = fmaxf(mx, fabsf(t[r][0] - t[r][2])); mx = fmaxf(mx, fabsf(t[r][1] + t[r][2]));
      mx = fmaxf(mx, fabsf(t[r][2] - t[r][1])); mx = fmaxf(mx, fabsf(t[r][1] - t[r][3]));
    }
  }
  __shared__ int shm;
  if (threadIdx.x == 0) shm = 0;
  __syncthreads();
  if (mx > 0.f && mx == mx) atomicMax(&shm, __builtin_bit_cast(int, mx));
  __syncthreads();
  if (threadIdx.x == 0 && shm) atomicMax(reinterpret_cast<int*>(out), shm);
}

}  // namespace wgrad

int launch_wino_vmax(const View& g, int B, int H, int W, float* out, hipStream_t st) {
  if (!g.p || !out || g.up) return HCF_ERR_ARG;
  const long long total = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * g.n;
  const unsigned nb = (unsigned)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(wgrad::wino_vmax_kernel, dim3(nb), dim3(256), 0, st, g, B, H, W, out);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

int launch_absmax(const View& g, int B, int H, int W, float* out, hipStream_t st) {
  if (!g.p || !out) return HCF_ERR_ARG;
  const long long npix = (long long)B * H * W;
  const unsigned nb = (unsigned)std::min<long long>((npix + 255) / 256, 2048);
  hipLaunchKernelGGL(wgrad::absmax_kernel, dim3(nb), dim3(256), 0, st, g, npix, out);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// Tiles the kernels walk; *strip_w: WgradArgs::strip_w of the f16x3 kernel (strips when they save >= 10 % of the tiles).
static int wgrad_tiles(const WgradArgs& a0, int* strip_w) {
  const int tiles_y = (a0.H + 7) / 8, per_image = a0.B * ((a0.W + 31) / 32) * tiles_y;
  if (strip_w) *strip_w = 0;
  const bool off = getenv("HCF_NO_WG_STRIP") != nullptr;             // A/B knob
  if (!a0.g_max || off || a0.B < 2 || (long long)a0.B * (a0.W + 1) >= 65536) return per_image;
  const int strips = ((a0.B * (a0.W + 1) + 31) / 32) * tiles_y;
  if (strips * 10 > per_image * 9) return per_image;
  if (strip_w) *strip_w = a0.W + 1;
  return strips;
}

size_t conv_wgrad_scratch_floats(const WgradArgs& a0, int* out_nblk_x, int* out_tpb) {
  int nicb = 0;
  for (int i = 0; i < a0.nsrc; ++i) nicb += (a0.src[i].n + 31) >> 5;
  const int nocb = (a0.g.n + 31) >> 5;
  const int tiles = wgrad_tiles(a0, nullptr);
  const int pairs = nicb * nocb;
  // fp32 kernel: one resident round of 256 CUs x 2 blocks (measured best of 512 / 768 / 1024 / 2048; phase-aligned rounds do
  // not overlap). The f16x3 kernel holds one 8-wave block per CU (92 KB of LDS).
  static const int wg_blocks = getenv("HCF_WG_BLOCKS") ? atoi(getenv("HCF_WG_BLOCKS")) : 0;   // experiment knob, read once
  int nblk_x;
  if (a0.g_max) {                                   // at most one full round of 256 blocks (264 blocks would cost two)
    const int target = wg_blocks > 0 ? wg_blocks : (a0.blocks_hint > 0 ? a0.blocks_hint : 256);
    nblk_x = target / pairs;
  } else {
    nblk_x = ((wg_blocks > 0 ? wg_blocks : 512) + pairs - 1) / pairs;
  }
  if (nblk_x > tiles) nblk_x = tiles;
  if (nblk_x < 1) nblk_x = 1;
  const int tpb = (tiles + nblk_x - 1) / nblk_x;
  nblk_x = (tiles + tpb - 1) / tpb;
  if (out_nblk_x) *out_nblk_x = nblk_x;
  if (out_tpb) *out_tpb = tpb;
  return (size_t)nblk_x * pairs * a0.taps * 1024;
}

// LDS form of the f16x3 weight-gradient kernels: 2 = double-buffered, interleaved, four threads per pixel (the default; same box,
// profiles/r05_ab_wgrad_lds_forms.txt: batched launch 263.4 -> 229.2 us, one-conv 3x3 launch 56.3 -> 49.1 us, 1x1 launch
// 25.3 -> 21.0 us), 1 = double-buffered with one block store per tile (HCF_WG_DB_BLOCK=1), 0 = single buffer (HCF_WG_SINGLE_BUF=1).
// Read per launch: the tests compare the forms inside one process. All three are bit-identical.
static int wgrad_lds_form() {
  const char* const e = getenv("HCF_WG_SINGLE_BUF");
  if (e && atoi(e) != 0) return 0;
  const char* const b = getenv("HCF_WG_DB_BLOCK");
  return (b && atoi(b) != 0) ? 1 : 2;
}

int launch_wgrad_reduce_batch(const WgradReduceJob* jobs_dev, int njobs, long long nblocks, hipStream_t st) {
  if (!jobs_dev || njobs < 1 || nblocks < 1 || nblocks > 0x7fffffffLL) return HCF_ERR_ARG;
  hipLaunchKernelGGL(wgrad::wgrad_reduce_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, jobs_dev, njobs);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

int launch_conv_wgrad(const WgradArgs& a0, hipStream_t st, WgradReduceJob* defer) {
  if (a0.nsrc < 1 || a0.nsrc > kMaxSrc || (a0.taps != 9 && a0.taps != 1) || !a0.dw || !a0.g.p || a0.g.n < 1 || !a0.part)
    return HCF_ERR_ARG;
  WgradArgs a = a0;
  static const int wg_dbg = getenv("HCF_WG_DBG") ? atoi(getenv("HCF_WG_DBG")) : 0;   // read once
  a.dbg = wg_dbg;
  int nicb = 0, cin = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    if ((a.H >> a.src[i].up) << a.src[i].up != a.H || (a.W >> a.src[i].up) << a.src[i].up != a.W) return HCF_ERR_ARG;
    nicb += (a.src[i].n + 31) >> 5;
    cin += a.src[i].n;
  }
  a.cin_total = cin;
  int nblk_x = 0;
  const size_t need = conv_wgrad_scratch_floats(a, &nblk_x, &a.tpb);
  (void)wgrad_tiles(a, &a.strip_w);
  a.strip_magic = a.strip_w ? (unsigned)(0x100000000ull / (unsigned)a.strip_w) + 1u : 0u;
  if (need > a.part_cap) return HCF_ERR_NOMEM;
  const int nocb = (a.g.n + 31) >> 5;
  const dim3 grid((unsigned)nblk_x, (unsigned)nicb, (unsigned)nocb);
  bool vec = (((a.g.cs | a.g.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.g.p) & 15) == 0);
  for (int i = 0; i < a.nsrc; ++i)
    vec = vec && (((a.src[i].cs | a.src[i].c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.src[i].p) & 15) == 0);
  if (a.g_max) {
    // f16 matrix cores (one-time opt-in to > 64 KB of dynamic LDS per instantiation)
    static bool attr_dev[64][12] = {};    // per device: several GPUs in one process (nn.DataParallel replicas)
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return HCF_ERR_HIP;
    bool (&attr)[12] = attr_dev[dev_];
    const int db = wgrad_lds_form();
    auto go = [&](auto fn, int idx, int ldsb) {
      if (!attr[idx]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) != hipSuccess) return false;
        attr[idx] = true;
      }
      hipLaunchKernelGGL(fn, grid, dim3(512), ldsb, st, a);
      return true;
    };
    bool ok;
    if (db == 2) {
      if (a.taps == 9 && vec) ok = go(wgrad::conv_wgrad_f16x3_kernel<9, true, 2>, 8, wgrad::Wg16<9>::LDS2_BYTES);
      else if (a.taps == 9) ok = go(wgrad::conv_wgrad_f16x3_kernel<9, false, 2>, 9, wgrad::Wg16<9>::LDS2_BYTES);
      else if (vec) ok = go(wgrad::conv_wgrad_f16x3_kernel<1, true, 2>, 10, wgrad::Wg16<1>::LDS2_BYTES);
      else ok = go(wgrad::conv_wgrad_f16x3_kernel<1, false, 2>, 11, wgrad::Wg16<1>::LDS2_BYTES);
    } else if (db == 1) {
      if (a.taps == 9 && vec) ok = go(wgrad::conv_wgrad_f16x3_kernel<9, true, 1>, 4, wgrad::Wg16<9>::LDS2_BYTES);
      else if (a.taps == 9) ok = go(wgrad::conv_wgrad_f16x3_kernel<9, false, 1>, 5, wgrad::Wg16<9>::LDS2_BYTES);
      else if (vec) ok = go(wgrad::conv_wgrad_f16x3_kernel<1, true, 1>, 6, wgrad::Wg16<1>::LDS2_BYTES);
      else ok = go(wgrad::conv_wgrad_f16x3_kernel<1, false, 1>, 7, wgrad::Wg16<1>::LDS2_BYTES);
    } else
    if (a.taps == 9 && vec) ok = go(wgrad::conv_wgrad_f16x3_kernel<9, true, 0>, 0, wgrad::Wg16<9>::LDS_BYTES);
    else if (a.taps == 9) ok = go(wgrad::conv_wgrad_f16x3_kernel<9, false, 0>, 1, wgrad::Wg16<9>::LDS_BYTES);
    else if (vec) ok = go(wgrad::conv_wgrad_f16x3_kernel<1, true, 0>, 2, wgrad::Wg16<1>::LDS_BYTES);
    else ok = go(wgrad::conv_wgrad_f16x3_kernel<1, false, 0>, 3, wgrad::Wg16<1>::LDS_BYTES);
    if (!ok) return HCF_ERR_HIP;
  } else
  if (a.taps == 9 && vec) hipLaunchKernelGGL((wgrad::conv_wgrad_kernel<9, true>), grid, dim3(256), 0, st, a);
  else if (a.taps == 9) hipLaunchKernelGGL((wgrad::conv_wgrad_kernel<9, false>), grid, dim3(256), 0, st, a);
  else if (vec) hipLaunchKernelGGL((wgrad::conv_wgrad_kernel<1, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad::conv_wgrad_kernel<1, false>), grid, dim3(256), 0, st, a);
  if (defer) {                                    // the caller batches the reduce steps (launch_wgrad_reduce_batch)
    defer->a = a; defer->nx = nblk_x; defer->nicb = nicb; defer->nocb = nocb; defer->nbx = (a.taps * 1024 + 255) / 256; defer->blk0 = 0;
    return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
  }
  const dim3 rgrid((unsigned)((a.taps * 1024 + 255) / 256), (unsigned)nicb, (unsigned)nocb);
  hipLaunchKernelGGL(wgrad::wgrad_reduce_kernel, rgrid, dim3(256), 0, st, a, nblk_x, a.taps);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// f16x3 3x3 weight gradients of n <= kWgBatchMax convs as one launch; jobs[i].part / part_cap / blocks_hint set by the caller
// (conv_wgrad_scratch_floats with the same blocks_hint sizes part). HCF_ERR_UNSUPPORTED: some job does not qualify (no g_max,
// 1x1, unaligned views) -- the caller launches them one by one. defer[i] receives job i's reduce step.
int launch_conv_wgrad_batch(const WgradArgs* jobs, int n, hipStream_t st, WgradReduceJob* defer) {
  if (!jobs || n < 1 || n > kWgBatchMax || !defer) return HCF_ERR_ARG;
  WgradBatchArgs b;
  memset(&b, 0, sizeof(b));
  b.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    WgradArgs a = jobs[i];
    if (a.nsrc < 1 || a.nsrc > kMaxSrc || !a.dw || !a.g.p || a.g.n < 1 || !a.part) return HCF_ERR_ARG;
    if (a.taps != 9 || !a.g_max) return HCF_ERR_UNSUPPORTED;
    int nicb = 0, cin = 0;
    bool vec = (((a.g.cs | a.g.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.g.p) & 15) == 0);
    for (int k = 0; k < a.nsrc; ++k) {
      if ((a.H >> a.src[k].up) << a.src[k].up != a.H || (a.W >> a.src[k].up) << a.src[k].up != a.W) return HCF_ERR_ARG;
      nicb += (a.src[k].n + 31) >> 5;
      cin += a.src[k].n;
      vec = vec && (((a.src[k].cs | a.src[k].c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.src[k].p) & 15) == 0);
    }
    if (!vec) return HCF_ERR_UNSUPPORTED;
    a.cin_total = cin;
    a.dbg = 0;
    int nblk_x = 0;
    const size_t need = conv_wgrad_scratch_floats(a, &nblk_x, &a.tpb);
    if (need > a.part_cap) return HCF_ERR_NOMEM;
    (void)wgrad_tiles(a, &a.strip_w);
    a.strip_magic = a.strip_w ? (unsigned)(0x100000000ull / (unsigned)a.strip_w) + 1u : 0u;
    const int nocb = (a.g.n + 31) >> 5;
    b.a[i] = a;
    b.nicb[i] = nicb; b.nocb[i] = nocb;
    b.blk0[i] = total;
    total += nblk_x * nicb * nocb;
    defer[i].a = a; defer[i].nx = nblk_x; defer[i].nicb = nicb; defer[i].nocb = nocb; defer[i].nbx = (9 * 1024 + 255) / 256; defer[i].blk0 = 0;
  }
  b.blk0[n] = total;
  static bool attr_dev[64][3] = {};
  int dev_ = 0;
  if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return HCF_ERR_HIP;
  const int db = wgrad_lds_form();
  auto fn = db == 2 ? wgrad::conv_wgrad_f16x3_batch_kernel<true, 2>
          : db == 1 ? wgrad::conv_wgrad_f16x3_batch_kernel<true, 1> : wgrad::conv_wgrad_f16x3_batch_kernel<true, 0>;
  const int ldsb = db ? wgrad::Wg16<9>::LDS2_BYTES : wgrad::Wg16<9>::LDS_BYTES;
  if (!attr_dev[dev_][db]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) != hipSuccess)
      return HCF_ERR_HIP;
    attr_dev[dev_][db] = true;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)total), dim3(512), ldsb, st, b);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace hcf
