// FCN coupling net, layers 1 + 2 for SMALL input widths (Basic.py:441-444: conv3x3 -> ActNorm -> ReLU -> conv1x1 -> ActNorm ->
// ReLU) as ONE persistent launch: h2 = relu(AN2(W2 relu(AN1(conv3x3(z1) [+ pre])))).
//
// Why a dedicated kernel: after the conditional features' share of conv1 has been hoisted out of the flow steps (it does not
// depend on z: hcf_engine.hip run_coupling_net, `pre`), every FCN of the SR nets sees 3..12 input channels = ONE 16-channel K
// chunk. The generic f16x3 kernel (hcf_conv_f16x3.hip, FUSE2) then spends its life outside the matrix cores: every 8 x 32 tile
// re-stages 36.8 KB of conv1 weights through registers into LDS, exposes two L2 round trips and four barriers for 156 MFMAs per
// wave -- 263 us per launch at 16 x 320 x 320 against an MFMA floor of 83 us and an HBM floor of 70 us
// (profiles/r03_notes.md). Here
//   * the grid is persistent (2 blocks per CU) and BOTH layers' weight fragments stay in registers for the life of the block,
//   * the next tile's z1 halo travels in registers while the current tile computes (no exposed HBM latency),
//   * tiles are 4 rows x 32 columns (128 pixels): the layer-1 tile (split f16, A operand of the 1x1 layer) and the fp32 output
//     tile (transposed for 16-byte stores) are 34 / 32 KB and alias the halo buffer -> 34 KB of LDS per block,
//   * `pre` (optional): fp32 [B,H,W,64] added to the conv1 accumulators before the ActNorm -- the hoisted W_u * u term.
// Numerics are those of hcf_conv_f16x3.hip (3-term f16 split, fp32 accumulation, weights pre-scaled by 2^11): the conv1 /
// conv2 packs are the same host packs (pack_conv_weights_f16x3), the epilogue algebra is copied term by term.
#include "hcf_common.h"
#include <cstdlib>

namespace hcf {
namespace fcn12 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(1)))* gfptr;
typedef const f32x4 __attribute__((address_space(1)))* gf4ptr;

constexpr int TW = 32, TH = 4, HW = TW + 2, HH = TH + 2, HP = HH * HW;       // 204 halo pixels
constexpr int REC = 80;                      // bytes per halo pixel: [16 hi | 16 lo | pad], conflict-free ds_read_b128
constexpr int NLOAD = HP * 4, NSLOT = (NLOAD + 255) / 256;                     // 816 float4 units, 4 per thread
constexpr int BHALF = 64 * 16;               // bytes of one (tap, plane, k-half) of a 64-channel pack: [64 n][8 halves]
constexpr int F2S = 272;                     // bytes per pixel of the layer-1 tile: 4 x [16 hi | 16 lo] + 16 pad. An odd multiple of 16 B:
                                             // the A-fragment reads of 32 consecutive pixels (ds_read_b128) are conflict-free; with
                                             // the natural 256 every lane of a group lands in the same four banks
constexpr int X_BYTES = TH * TW * F2S;                                         // 34 816: layer-1 tile (split f16) / fp32 output tile (32 KB)
constexpr int W2_BYTES = 4 * 2 * 2 * BHALF;                                    // 16 384: conv2 pack [kc][plane][k-half][64 n][8 halves]
#ifndef FCN12_W1_LDS
#define FCN12_W1_LDS 0   // 1: conv1's pack in LDS too (36 KB) instead of registers (A/B knob of tools/micro/fcn12_bench.hip)
#endif
constexpr int W1_BYTES = 9 * 2 * 2 * BHALF;                                    // 36 864
constexpr int LDS_BYTES = X_BYTES + (FCN12_W1_LDS ? W1_BYTES : W2_BYTES);      // (with conv1's pack in LDS conv2's fragments come from L1)  // 51 200 (the conv1 weights live in registers)
constexpr float SPLIT = 2048.f;
#ifndef FCN12_ABL
#define FCN12_ABL 0      // timing ablations of tools/micro/fcn12_bench.hip (results invalid); the library is built with 0
#endif
static_assert(HP * REC <= X_BYTES && TH * TW * 64 * 4 <= X_BYTES, "halo buffer and output tile alias the exchange region");

__device__ __forceinline__ f32x4 split4(const f32x4 v) {      // hcf_conv_f16x3.hip: {hi.xy, hi.zw, lo.xy, lo.zw}
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
__device__ __forceinline__ gfptr uniform_ptr(const float* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gfptr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int xcd_remap(int orig, int n) {
  const int xcd = orig & 7, q = n >> 3, r = n & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// a.src[0]: z1 window (n <= 16 channels, 16-byte addressable), a.wpack: f16x3 pack of conv1 restricted to it (ONE chunk),
// a.bias / scale / act: ActNorm + ReLU of layer 1; a.w2 / bias2 / scale2 / act2: layer 2; a.out: h2 (64 channels);
// a.res1 (PRE): the pre-activation term, fp32 NHWC window of 64 channels.
template <bool PRE>
__global__ __launch_bounds__(256, 2) void fcn12_kernel(const ConvArgs a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;            // rows {2 wm, 2 wm + 1} of the tile, 32-channel n tile wn
  const int H = a.H, W = a.W;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int uq = tid & 3;                             // this thread's 4-channel unit of the 16-channel chunk
  const int n0 = __builtin_amdgcn_readfirstlane(a.src[0].n), cs0 = __builtin_amdgcn_readfirstlane(a.src[0].cs);
  const int valid = n0 - 4 * uq;                      // channels of the unit that exist (<= 0: all zero)
  const gfptr sp0 = uniform_ptr(a.src[0].p + a.src[0].c0) + 4 * ((valid > 0) ? uq : 0);
  const gfptr zpage = uniform_ptr(a.zeros);

  const int oc = wn * 32 + li;
  const float bias1 = a.bias[oc], scale1 = a.scale[oc], slope1 = act_slope(a.act);
  const float bias2 = a.bias2[oc], scale2 = a.scale2[oc], slope2 = act_slope(a.act2);
  const float UNSPLIT = 1.0f / SPLIT;
  const int abase = ((2 * wm) * HW + li) * REC + half * 16;
  // conv1's B fragments of this wave's n tile (9 taps x 2 planes x 4 VGPRs) live in REGISTERS for the life of the block (the
  // block is persistent): the LDS pipe is what bounds this kernel, and re-reading them per tile was a quarter of its traffic.
  // conv2's 16 KB pack stays in LDS (8 fragment reads per tile; in registers too the kernel spills).
#if FCN12_W1_LDS
  {
    const gf4ptr wq4 = (gf4ptr)uniform_ptr(a.wpack);
#pragma unroll
    for (int s = 0; s < W1_BYTES / 16 / 256; ++s) *reinterpret_cast<f32x4*>(lds + X_BYTES + (tid + 256 * s) * 16) = wq4[tid + 256 * s];
  }
  const char* const w1l = lds + X_BYTES + half * BHALF + oc * 16;
#define FCN12_W1A(T) (*reinterpret_cast<const f16x8*>(w1l + (T) * (4 * BHALF)))
#define FCN12_W1B(T) (*reinterpret_cast<const f16x8*>(w1l + (T) * (4 * BHALF) + 2 * BHALF))
#else
#define FCN12_W1A(T) w1a[T]
#define FCN12_W1B(T) w1b[T]
#endif
  f16x8 w1a[9], w1b[9];
  {
    const char __attribute__((address_space(1)))* const wq =
        (const char __attribute__((address_space(1)))*)uniform_ptr(a.wpack) + half * BHALF + oc * 16;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      if (!FCN12_W1_LDS) {
        w1a[tp] = *(const f16x8 __attribute__((address_space(1)))*)(wq + tp * (4 * BHALF));
        w1b[tp] = *(const f16x8 __attribute__((address_space(1)))*)(wq + tp * (4 * BHALF) + 2 * BHALF);
      }
    }
    // conv2 pack: 16 KB, copied into LDS once (behind the exchange region)
    const gf4ptr vq = (gf4ptr)uniform_ptr(a.w2);
    if (!FCN12_W1_LDS) {
#pragma unroll
      for (int s = 0; s < W2_BYTES / 16 / 256; ++s) *reinterpret_cast<f32x4*>(lds + X_BYTES + (tid + 256 * s) * 16) = vq[tid + 256 * s];
    }
  }
  const char* const w2l = lds + X_BYTES + half * BHALF + oc * 16;
  const char __attribute__((address_space(1)))* const w2g =
      (const char __attribute__((address_space(1)))*)uniform_ptr(a.w2) + half * BHALF + oc * 16;

  f32x4 stg[NSLOT];
  int tb = 0, ty0 = 0, tx0 = 0;                       // coordinates of the tile whose halo sits in stg
#define FCN_TILE_COORDS(T, B_, Y0_, X0_)                                     \
  {                                                                          \
    const int v_ = xcd_remap((T), ntiles);                                   \
    X0_ = (v_ % tiles_x) * TW;                                               \
    Y0_ = ((v_ / tiles_x) % tiles_y) * TH;                                   \
    B_ = v_ / (tiles_x * tiles_y);                                           \
  }
#define FCN_LOAD(B_, Y0_, X0_)                                               \
  {                                                                          \
    _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                      \
      const int q = tid + 256 * s;                                           \
      const int hp = min(q >> 2, HP - 1);                                    \
      const int hy = hp / HW, hx = hp - hy * HW;                             \
      const int y = (Y0_) + hy - 1, x = (X0_) + hx - 1;                      \
      const bool ok = y >= 0 && y < H && x >= 0 && x < W && valid > 0;       \
      const int pix = ((B_) * H + min(max(y, 0), H - 1)) * W + min(max(x, 0), W - 1); \
      gfptr p = ok ? sp0 + (unsigned)(pix * cs0) : zpage;                    \
      if (FCN12_ABL & 8) p = zpage;                                          \
      stg[s] = *(gf4ptr)(p);                                                 \
    }                                                                        \
  }
  int t = blockIdx.x;
  if (t >= ntiles) return;
  FCN_TILE_COORDS(t, tb, ty0, tx0)
  FCN_LOAD(tb, ty0, tx0)
  float chk = 0.f;
  unsigned omask = 0;            // range flag of this lane: bit 0 + one bit per sample slot (as hcf_conv_f16x3.hip)

  for (; t < ntiles; t += gridDim.x) {
    const int b = tb, y0 = ty0, x0 = tx0;
    f32x16 acc[2];
    if (PRE) {          // acc starts at 2^11 * pre: the hoisted conditional term joins the conv1 sum before the ActNorm
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int y = y0 + 2 * wm + m;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const size_t pix = (size_t)((size_t)b * H + min(y, H - 1)) * W + min(x, W - 1);
          acc[m][r] = a.res1.p[pix * a.res1.cs + a.res1.c0 + oc];
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    }
    // ---- stage this tile's halo: split in registers, [16 hi | 16 lo] records
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      f32x4 v = stg[s];
      if (valid < 4) {
        v.x = (valid > 0) ? v.x : 0.f; v.y = (valid > 1) ? v.y : 0.f; v.z = (valid > 2) ? v.z : 0.f; v.w = 0.f;
      }
      stg[s] = split4(v);
    }
    __syncthreads();                                   // the previous tile's output reads of the exchange region are done
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int q = tid + 256 * s;
      if (q < NLOAD) {
        char* rec = lds + (q >> 2) * REC + (q & 3) * 8;
        union { f16x4 h[2]; f32x4 f; } u_;
        u_.f = stg[s];
        *reinterpret_cast<f16x4*>(rec) = u_.h[0];
        *reinterpret_cast<f16x4*>(rec + 32) = u_.h[1];
      }
    }
    __syncthreads();
    // ---- the next tile's halo flies under this tile's MFMAs
    const int tn = t + gridDim.x;
    if (tn < ntiles) {
      FCN_TILE_COORDS(tn, tb, ty0, tx0)
      FCN_LOAD(tb, ty0, tx0)
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PRE) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] *= SPLIT;
    }
    // ---- layer 1: 9 taps x one 16-channel chunk, dx-major: the four halo rows of a column offset are read once and serve
    // both output rows (24 fragment reads per tile instead of 36)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      f16x8 rhi[4], rlo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char* rec = lds + abase + (j * HW + dx) * REC;
        rhi[j] = *reinterpret_cast<const f16x8*>(rec);
        rlo[j] = *reinterpret_cast<const f16x8*>(rec + 32);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const f16x8 b1 = FCN12_W1A(dy * 3 + dx), b2 = FCN12_W1B(dy * 3 + dx);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rhi[m + dy], b1, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rhi[m + dy], b2, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rlo[m + dy], b1, acc[m], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][r], 0.f, chk);       // inf / NaN: an input left the f16 range
    // ---- layer-1 epilogue -> split f16 A operand of the 1x1 layer: record (px, kc) = [16 hi | 16 lo] of channels 16 kc ..
    __syncthreads();                                   // every wave is done with the halo records
    {
      // A lane holds ONE channel of 16 pixels; neighbouring lanes hold neighbouring channels. Lanes 2j / 2j+1 swap half of
      // their values (one DPP move per pixel pair carries the f16 hi and lo parts) so that each writes two channels of one
      // pixel as a dword: 32 ds_write_b32 per lane instead of 64 ds_write_b16.
      const int kc = oc >> 4, kpe = oc & 14;
      const bool odd = (li & 1) != 0;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          uint32_t own[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float v = (acc[m][r + e] * UNSPLIT + bias1) * scale1;
            v = apply_act(v, slope1);
            const _Float16 h = (_Float16)v;
            const f16x2 hl = {h, (_Float16)(v - (float)h)};
            own[e] = __builtin_bit_cast(uint32_t, hl);
          }
          const uint32_t give = odd ? own[0] : own[1];                  // the pixel the partner writes
          const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
          const uint32_t mine = odd ? own[1] : own[0];
          const uint32_t lo_ch = odd ? got : mine, hi_ch = odd ? mine : got;   // channels (oc & ~1), (oc | 1) of this lane's pixel
          const int px = (2 * wm + m) * TW + ((r + (odd ? 1 : 0)) & 3) + 8 * (r >> 2) + 4 * half;
          char* rec = lds + px * F2S + kc * 64 + kpe * 2;
          *reinterpret_cast<uint32_t*>(rec) = (lo_ch & 0xffffu) | (hi_ch << 16);
          *reinterpret_cast<uint32_t*>(rec + 32) = (lo_ch >> 16) | (hi_ch & 0xffff0000u);
        }
    }
    __syncthreads();
    // ---- layer 2: 64 x 64 GEMM per pixel row
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const f16x8 b1 = FCN12_W1_LDS ? *(const f16x8 __attribute__((address_space(1)))*)(w2g + (kc * 2 + 0) * (2 * BHALF))
                                    : *reinterpret_cast<const f16x8*>(w2l + (kc * 2 + 0) * (2 * BHALF));
      const f16x8 b2 = FCN12_W1_LDS ? *(const f16x8 __attribute__((address_space(1)))*)(w2g + (kc * 2 + 1) * (2 * BHALF))
                                    : *reinterpret_cast<const f16x8*>(w2l + (kc * 2 + 1) * (2 * BHALF));
      f16x8 ahi[2], alo[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const char* rec = lds + ((2 * wm + m) * TW + li) * F2S + kc * 64 + half * 16;
        ahi[m] = *reinterpret_cast<const f16x8*>(rec);
        alo[m] = *reinterpret_cast<const f16x8*>(rec + 32);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b1, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b2, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[m], b1, acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][r], 0.f, chk);
    omask |= (chk != chk) ? (1u | (2u << (b % 30))) : 0u;      // per tile: a block walks tiles of several samples
    chk = 0.f;
    // ---- output tile: pixel-major fp32 through LDS, 16-byte stores
    __syncthreads();                                   // every wave has read its layer-1 records
    float* const ldsT = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = (2 * wm + m) * TW + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = (acc[m][r] * UNSPLIT + bias2) * scale2;
        ldsT[px * 64 + oc] = apply_act(v, slope2);
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (TH * TW * 16) / 256; ++k) {
      const int idx = tid + 256 * k;
      const int px = idx >> 4, c4 = idx & 15;
      const int y = y0 + (px >> 5), x = x0 + (px & 31);
      const f32x4 v = *reinterpret_cast<const f32x4*>(ldsT + px * 64 + 4 * c4);
      if (y < H && x < W && (!(FCN12_ABL & 1) || v.x == 123.456f)) {
        const size_t pixo = (size_t)((size_t)b * H + y) * W + x;
        *reinterpret_cast<f32x4*>(a.out.p + pixo * a.out.cs + a.out.c0 + 4 * c4) = v;
      }
    }
  }
  if (omask) atomicOr(a.ovf, (int)omask);      // (rare path: every flagged lane reports its own samples)
#undef FCN_TILE_COORDS
#undef FCN_LOAD
}

}  // namespace fcn12

// HCF_ERR_UNSUPPORTED: this call does not fit the small-K persistent form (the generic FUSE2 kernel takes it)
int launch_fcn12(const ConvArgs& a, hipStream_t st) {
  static const bool off = getenv("HCF_NO_FCN12") != nullptr;         // A/B knob, read once
  if (off) return HCF_ERR_UNSUPPORTED;
  auto v16 = [](const View& v) { return v.p && (((v.cs | v.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0); };
  if (a.nsrc != 1 || a.nchunk != 1 || a.src[0].n > 16 || a.src[0].up || !a.w2 || !a.bias2 || !a.scale2 || a.out.n != 64 ||
      !v16(a.src[0]) || !v16(a.out) || a.res2.p || a.tC > 0 || a.in_max || !a.ovf || !a.zeros)
    return HCF_ERR_UNSUPPORTED;
  if (a.res1.p && (a.res1.n != 64 || a.res1.up)) return HCF_ERR_UNSUPPORTED;
  if ((long long)a.B * a.H * a.W * a.src[0].cs >= 0x7fffffffLL) return HCF_ERR_UNSUPPORTED;
  const int tiles_x = (a.W + fcn12::TW - 1) / fcn12::TW, tiles_y = (a.H + fcn12::TH - 1) / fcn12::TH;
  const long long ntiles = (long long)a.B * tiles_x * tiles_y;
  if (ntiles < 1 || ntiles > 0x7fffffffLL) return HCF_ERR_ARG;
  static int ncu_dev[64] = {0};
  static bool attr_dev[64][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return HCF_ERR_HIP;
  if (!ncu_dev[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    ncu_dev[dev] = v;
  }
  const unsigned grid = (unsigned)std::min<long long>(ntiles, 2LL * ncu_dev[dev]);
  const int pre = a.res1.p ? 1 : 0;
  auto go = [&](auto fn) {
    if (!attr_dev[dev][pre]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, fcn12::LDS_BYTES) != hipSuccess)
        return HCF_ERR_HIP;
      attr_dev[dev][pre] = true;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), fcn12::LDS_BYTES, st, a, (int)ntiles);
    return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
  };
  return pre ? go(fcn12::fcn12_kernel<true>) : go(fcn12::fcn12_kernel<false>);
}

}  // namespace hcf
