// Weight gradient of the fused 3x3 / 1x1 convolution (training path) for gfx950.
//
//   dW[oc][ic][tap] += sum over (b, y, x) of  X[b, y + dy - pad, x + dx - pad, ic] * G[b, y, x, oc]
//
// X is the forward input (the same <= 3 NHWC source windows, optionally read through a nearest upsample), G the
// gradient w.r.t. the conv's pre-activation output. Per tap this is a GEMM with M = input channels, N = output
// channels, K = pixels, run on v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation):
//   block  = 4 waves, one (32 input channels) x (32 output channels) tile of dW, all taps;
//   K loop = the block walks `tpb` pixel tiles of 8 x 32; per tile the 10 x 34 halo of X (32 channels) and the
//            8 x 32 tile of G (32 channels) are staged in LDS as [pixel][32 ch] (conflict-free ds_read_b32: lane i
//            reads channel i), wave w owns tile rows {2w, 2w+1} = 32 K-steps of two pixels, 9 MFMAs each (one per
//            tap: the A operand is the same halo tile shifted by the tap);
//   end    = the 4 waves' accumulators are summed through LDS into the block's partial dW tile (scratch), and
//            wgrad_reduce_kernel adds the partial tiles of the blocks that share a dW tile in a fixed order
//            (deterministic; the first version used fp32 atomics and spent 10-25x the MFMA time in them).
#include "hcf_common.h"

namespace hcf {
namespace wgrad {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;

template <int TAPS>
__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  constexpr int HH = TH + 2 * PAD, HW = TW + 2 * PAD, HP = HH * HW;
  __shared__ __attribute__((aligned(16))) float xs[HP * 32];
  __shared__ __attribute__((aligned(16))) float gs[TH * TW * 32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
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
  const bool vecx = (((sv.cs | (sv.c0 + ic0)) & 3) == 0) && ((reinterpret_cast<uintptr_t>(sv.p) & 15) == 0);
  const bool vecg = (((a.g.cs | (a.g.c0 + oc0)) & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.g.p) & 15) == 0);

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // register staging: the NEXT tile's global loads are issued before this tile's MFMAs and land under them
  constexpr int NX = (HP * 8 + 255) / 256, NG = (TH * TW * 8) / 256;
  f32x4 rx[NX], rg[NG];
#define HCF_WG_LOAD(TILE)                                                                                     \
  {                                                                                                           \
    const int txb_ = (TILE) % tiles_x, tyb_ = ((TILE) / tiles_x) % tiles_y, b_ = (TILE) / (tiles_x * tiles_y); \
    const int x0_ = txb_ * TW, y0_ = tyb_ * TH;                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < NX; ++s_) {                                                       \
      const int q = tid + 256 * s_;                                                                           \
      const int hp = min(q >> 3, HP - 1), c4 = (q & 7) * 4;                                                   \
      const int hy = hp / HW, hx = hp - hy * HW;                                                              \
      const int y = y0_ + hy - PAD, x = x0_ + hx - PAD;                                                       \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                         \
      if (y >= 0 && y < H && x >= 0 && x < W && c4 < icn) {                                                   \
        const float* p = sv.p + ((size_t)((size_t)b_ * Hs + (y >> up)) * Ws + (x >> up)) * sv.cs + sv.c0 + ic0 + c4; \
        if (vecx && c4 + 4 <= icn) {                                                                          \
          v = *reinterpret_cast<const f32x4*>(p);                                                             \
        } else {                                                                                              \
          v.x = p[0];                                                                                         \
          if (c4 + 1 < icn) v.y = p[1];                                                                       \
          if (c4 + 2 < icn) v.z = p[2];                                                                       \
          if (c4 + 3 < icn) v.w = p[3];                                                                       \
        }                                                                                                     \
      }                                                                                                       \
      rx[s_] = v;                                                                                             \
    }                                                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < NG; ++s_) {                                                       \
      const int q = tid + 256 * s_;                                                                           \
      const int px = q >> 3, c4 = (q & 7) * 4;                                                                \
      const int y = y0_ + (px >> 5), x = x0_ + (px & 31);                                                     \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                         \
      if (y < H && x < W && c4 < ocn) {                                                                       \
        const float* p = a.g.p + ((size_t)((size_t)b_ * H + y) * W + x) * a.g.cs + a.g.c0 + oc0 + c4;        \
        if (vecg && c4 + 4 <= ocn) {                                                                          \
          v = *reinterpret_cast<const f32x4*>(p);                                                             \
        } else {                                                                                              \
          v.x = p[0];                                                                                         \
          if (c4 + 1 < ocn) v.y = p[1];                                                                       \
          if (c4 + 2 < ocn) v.z = p[2];                                                                       \
          if (c4 + 3 < ocn) v.w = p[3];                                                                       \
        }                                                                                                     \
      }                                                                                                       \
      rg[s_] = v;                                                                                             \
    }                                                                                                         \
  }
  const int t0 = blockIdx.x * a.tpb, t1 = min(ntiles, t0 + a.tpb);
  if (t0 < t1) HCF_WG_LOAD(t0)
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();                              // the previous tile's fragments have been read
#pragma unroll
    for (int s_ = 0; s_ < NX; ++s_) {
      const int q = tid + 256 * s_;
      if (q < HP * 8) *reinterpret_cast<f32x4*>(xs + (q >> 3) * 32 + (q & 7) * 4) = rx[s_];
    }
#pragma unroll
    for (int s_ = 0; s_ < NG; ++s_) {
      const int q = tid + 256 * s_;
      *reinterpret_cast<f32x4*>(gs + (q >> 3) * 32 + (q & 7) * 4) = rg[s_];
    }
    __syncthreads();
    if (tile + 1 < t1) HCF_WG_LOAD(tile + 1)
    // ---- 2 rows x 16 pixel pairs per wave
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr;
#pragma unroll 4
      for (int pp = 0; pp < 16; ++pp) {
        const int xx = 2 * pp + half;
        const float bg = gs[(row * TW + xx) * 32 + li];                  // B[k = half][n = li]
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int dy = (TAPS == 9) ? t / 3 : 0, dx = (TAPS == 9) ? t % 3 : 0;
          const float ax = xs[((row + dy) * HW + xx + dx) * 32 + li];    // A[m = li][k = half]
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bg, acc[t], 0, 0, 0);
        }
      }
    }
  }
#undef HCF_WG_LOAD

  // ---- cross-wave reduction through LDS; the block's partial dW tile goes to scratch (coalesced), a second kernel
  // sums the blocks that share a tile in a fixed order (deterministic; same-address atomics were 10-25x slower)
  float* red = xs;                                 // 4 waves x 1024 floats
  float* part = a.part + ((size_t)((size_t)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (TAPS * 1024);
#pragma unroll 1
  for (int t = 0; t < TAPS; ++t) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * half;                  // input channel within the block
      red[wave * 1024 + m * 32 + li] = acc[t][r];
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) part[t * 1024 + e] = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
  }
}

// dW[oc][ic][tap] += sum over the nx blocks of part[bx][icb][ocb][tap][m][n]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a, int nx, int taps) {
  const int icb = blockIdx.y, ocb = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;          // (tap, m, n)
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
  const size_t stride = (size_t)gridDim.y * gridDim.z * taps * 1024;
  const float* p = a.part + ((size_t)icb * gridDim.z + ocb) * (taps * 1024) + e;
  // fixed summation order (8 interleaved partial sums, then their sum): deterministic, and 8 loads in flight
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= nx; b += 8)
#pragma unroll
    for (int j = 0; j < 8; ++j) s8[j] += p[(size_t)(b + j) * stride];
  for (; b < nx; ++b) s8[b & 7] += p[(size_t)b * stride];
  const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  a.dw[((size_t)oc * a.cin_total + ic_base + ic) * taps + t] += s;
}

}  // namespace wgrad

size_t conv_wgrad_scratch_floats(const WgradArgs& a0, int* out_nblk_x, int* out_tpb) {
  int nicb = 0;
  for (int i = 0; i < a0.nsrc; ++i) nicb += (a0.src[i].n + 31) >> 5;
  const int nocb = (a0.g.n + 31) >> 5;
  const int tiles = a0.B * ((a0.W + 31) / 32) * ((a0.H + 7) / 8);
  const int pairs = nicb * nocb;
  int nblk_x = (1024 + pairs - 1) / pairs;          // ~4 blocks per CU in total
  if (nblk_x > tiles) nblk_x = tiles;
  if (nblk_x < 1) nblk_x = 1;
  const int tpb = (tiles + nblk_x - 1) / nblk_x;
  nblk_x = (tiles + tpb - 1) / tpb;
  if (out_nblk_x) *out_nblk_x = nblk_x;
  if (out_tpb) *out_tpb = tpb;
  return (size_t)nblk_x * pairs * a0.taps * 1024;
}

int launch_conv_wgrad(const WgradArgs& a0, hipStream_t st) {
  if (a0.nsrc < 1 || a0.nsrc > kMaxSrc || (a0.taps != 9 && a0.taps != 1) || !a0.dw || !a0.g.p || a0.g.n < 1 || !a0.part)
    return HCF_ERR_ARG;
  WgradArgs a = a0;
  int nicb = 0, cin = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    if ((a.H >> a.src[i].up) << a.src[i].up != a.H || (a.W >> a.src[i].up) << a.src[i].up != a.W) return HCF_ERR_ARG;
    nicb += (a.src[i].n + 31) >> 5;
    cin += a.src[i].n;
  }
  a.cin_total = cin;
  int nblk_x = 0;
  const size_t need = conv_wgrad_scratch_floats(a, &nblk_x, &a.tpb);
  if (need > a.part_cap) return HCF_ERR_NOMEM;
  const int nocb = (a.g.n + 31) >> 5;
  const dim3 grid((unsigned)nblk_x, (unsigned)nicb, (unsigned)nocb);
  if (a.taps == 9) hipLaunchKernelGGL(wgrad::conv_wgrad_kernel<9>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wgrad::conv_wgrad_kernel<1>, grid, dim3(256), 0, st, a);
  const dim3 rgrid((unsigned)((a.taps * 1024 + 255) / 256), (unsigned)nicb, (unsigned)nocb);
  hipLaunchKernelGGL(wgrad::wgrad_reduce_kernel, rgrid, dim3(256), 0, st, a, nblk_x, a.taps);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace hcf
