// 3x3 convolution on split16p tensors, staged by LDS-DMA (gfx950). fp32-equivalent products as in
// hcf_conv_f16x3.hip: a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation.
//
// Why this kernel exists (profiles/r02_notes.md): with operands resident in LDS every register tiling of the K loop
// runs at 1.45-1.55 PFLOP/s executed (95-99 % MFMA busy at the ~1.48 GHz this part clocks to under dense random-data
// f16 MFMA) -- the round-1 kernels reach 0.95-1.05 because of what happens AROUND the K loop: per-block prologue (first
// chunk's HBM latency) and epilogue (LDS transposition + barriers) that 2-3 co-resident blocks cannot cover on the short-K
// convs of an RDB, and the in-loop staging (global load -> registers -> split VALU -> ds_write -> two barriers). Here:
//   * activations are stored pre-split by their PRODUCER ("split16p": every aligned group of 16 channels of a pixel is a
//     64-byte record [hi k-half 0 | hi k-half 1 | lo k-half 0 | lo k-half 1] of f16, same bytes as fp32), so a consumer
//     needs no staging registers, no split VALU and no ds_write: 16-byte pieces travel HBM/L2 -> LDS by
//     global_load_lds_dwordx4, weights likewise; LDS is double-buffered (2 x 40 192 B -> two blocks per CU);
//   * blocks are persistent and walk (tile, 32-channel n-tile) units; the DMA of the next unit's first chunk is issued
//     during the last chunk of the current one: no prologue after the first unit;
//   * the matrix-core operand roles are swapped (A = weights: M = 32 output channels, B = pixels: N = 32 pixels of an image
//     row), so an accumulator lane holds 16 channels of ONE pixel: the epilogue stores 16-byte vectors straight from
//     registers (fp32 and / or split16p) -- no LDS transposition, no barrier; the other block on the CU computes meanwhile;
//   * dx-major sliding rows: per dx the three dy taps' weights are held in registers and every activation row fragment
//     feeds the (up to) three output rows it contributes to: 14 activation + 18 weight ds_read_b128 per 54 MFMAs.
// The k order inside a record follows the producer's accumulator layout: element e of k-half h is channel
// rec_channel(h, e) of the group; the weight pack uses the same order, so no data is ever permuted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include <vector>

namespace hcf {
namespace s16 {

__host__ __device__ constexpr int rec_channel(int h, int e) { return e < 4 ? 4 * h + e : 8 + 4 * h + (e - 4); }

constexpr int TW = 32, TH = 8, HWP = TW + 2, HHP = TH + 2;
constexpr int A_BYTES = HHP * HWP * 64;          // 21 760: 340 records
constexpr int A_PIECES = A_BYTES / 16;           // 1 360
constexpr int B_BYTES = 9 * 2 * 2 * 32 * 16;     // 18 432: [tap][plane][k-half][32 n][8 halves]
constexpr int STAGE = A_BYTES + B_BYTES;         // 40 192
constexpr int PIECES = STAGE / 16;               // 2 512 -> 40 wave-instructions of 64 pieces (the last one: 16)
constexpr int NSLOT = 10;                        // DMA instructions per wave and chunk
constexpr int STAGE_PITCH = 40 * 1024;           // every DMA instruction is a full 1 KB; 768-byte gap after each stage
constexpr int LDS_BYTES = 2 * STAGE_PITCH;       // 81 920 -> 2 blocks per CU use the whole 160 KB
constexpr float UNSPLIT = 1.f / 2048.f;

struct Args {
  const char* src;        // split16p tensor [B][src_planes][H][W][64 B]; the conv reads planes [src_rec0, src_rec0 + nchunk)
  int src_planes, src_rec0, nchunk;
  const char* wpack;      // [ntile_n][nchunk][B_BYTES] (pack_weights_s16)
  int ntile_n;            // 32-channel output tiles (1 or 2)
  const float* bias;      // [32 * ntile_n]
  const float* scale;     // [32 * ntile_n]
  int act;                // 0 none, 1 relu, 2 leaky relu 0.2
  // y = res2 + rs2 * (res1 + rs1 * act((acc + bias) * scale)); written as split16p and / or fp32
  char* out16; int out16_planes, out16_rec0;   // [B][out16_planes][H][W][64 B], planes [out16_rec0 + 2 nt, + 2)
  float* out32; int out32_cs, out32_c0;
  const float* res1; int res1_cs, res1_c0; float rs1;
  const float* res2; int res2_cs, res2_c0; float rs2;
  int B, H, W;
  int* ovf;               // raised when an accumulator is inf / NaN (|a| >= 65504 cannot be split)
  const char* zeros;      // >= 64 bytes of zeros in device memory (conv zero padding)
  int variant;            // experiment switches (tools/micro/conv_s16.hip)
  unsigned long long* dbg; // optional timing counters (S16_PROF builds): [0] vmcnt wait, [1] barrier wait, [2] block life, [3] epilogue, [4] samples
};

// w: PyTorch [cout][cin][3][3]; cin multiple of 16. Plane 0 = f16(w) * 2^11, plane 1 = f16((w - f16(w)) * 2^11).
// chunk_major = false: [ntile][chunk] blocks (conv_s16_kernel, one 32-channel tile per unit); true: [chunk][ntile] blocks
// (conv_s16w_kernel: both tiles of a chunk are one contiguous 36 KB piece).
static inline bool pack_weights_s16(const float* w, int cin, int cout, std::vector<uint16_t>& pk, bool chunk_major = false) {
  const int nchunk = cin / 16, ntn = (cout + 31) / 32;
  pk.assign(((size_t)ntn * nchunk + 1) * (B_BYTES / 2), 0);      // + one zero chunk: the DMA cursor runs one chunk ahead
  for (int nt = 0; nt < ntn; ++nt)
    for (int c = 0; c < nchunk; ++c)
      for (int t = 0; t < 9; ++t)
        for (int h = 0; h < 2; ++h)
          for (int n = 0; n < 32; ++n)
            for (int e = 0; e < 8; ++e) {
              const int oc = nt * 32 + n, ic = 16 * c + rec_channel(h, e);
              if (oc >= cout) continue;
              const float x = w[((size_t)oc * cin + ic) * 9 + t];
              if (!(fabsf(x) * 2048.f < 60000.f)) return false;
              const _Float16 hi = (_Float16)x;
              const _Float16 p0 = (_Float16)((float)hi * 2048.f), p1 = (_Float16)((x - (float)hi) * 2048.f);
              const size_t blk = chunk_major ? (size_t)c * ntn + nt : (size_t)nt * nchunk + c;
              const size_t o = (blk * B_BYTES) / 2 + (size_t)((t * 2 + 0) * 2 + h) * 256 + (size_t)n * 8 + e;
              memcpy(&pk[o], &p0, 2);
              memcpy(&pk[o + 512], &p1, 2);
            }
  return true;
}

#if defined(__HIPCC__)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef const char __attribute__((address_space(1)))* gcptr;
typedef __attribute__((address_space(3))) void* lptr;

__device__ __forceinline__ int xcd_remap(int orig, int n) {
  const int xcd = orig & 7, q = n >> 3, r = n & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}
__device__ __forceinline__ gcptr uniform_ptr(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gcptr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void glds16(gcptr g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lptr)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ float act1(float v, float slope, float lo) { return !(v <= 0.f) ? v : slope * fmaxf(v, lo); }

// Experiment switches (compile time, tools/micro/conv_s16.hip): -DS16_NO_DMA (no DMA inside the chunk loop: stale LDS),
// -DS16_NO_EPI (no epilogue stores), -DS16_ABL=n (DMA source ablations), -DS16_PROF (wait-time counters).
#if !defined(S16_ABL)
#define S16_ABL 0
#endif
constexpr int TAB_OFF = STAGE;                   // bias / scale table in the 768-byte gap behind stage 0: [64 bias][64 scale]

// OUT32 / OUT16: which output forms are written; RES: number of fp32 residual inputs (0..2)
template <bool OUT32, bool OUT16, int RES>
__global__ __launch_bounds__(256, 2) void conv_s16_kernel(const Args a, const int nunits) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, ntn = a.ntile_n, nchunk = a.nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int plane_b = H * W * 64;                          // bytes of one 16-channel plane of one image
  const int splanes = __builtin_amdgcn_readfirstlane(a.src_planes);

  // ---- DMA slots: instruction I = 4 j + wave moves pieces [64 I, 64 I + 64) of the stage image; pieces < 1360 are
  // activation pieces (record = piece >> 2 = halo pixel hy * 34 + hx, LDS slot = piece & 3 holds logical slot
  // (piece & 3) ^ ((hx >> 2) & 3): conflict-free ds_read_b128 fragments without padding), the rest weight pieces.
  // Instruction 39 (wave 3, j = 9) carries 16 pieces: its other lanes are masked off.
  int hyx[6];                  // (hy << 16) | (hx << 8) | byte offset of the fetched slot; -1: weight slot (j = 5 only)
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int p = (4 * j + wave) * 64 + lane;
    const int rec = p >> 2, hy = rec / HWP, hx = rec - hy * HWP;
    hyx[j] = (p < A_PIECES) ? ((hy << 16) | (hx << 8) | (((p & 3) ^ ((hx >> 2) & 3)) << 4)) : -1;
  }
  const gcptr srcp = uniform_ptr(a.src + (size_t)a.src_rec0 * plane_b);
  const gcptr wq = uniform_ptr(a.wpack);
  const gcptr zpage = uniform_ptr(a.zeros);
  const int boff0 = ((4 * 5 + wave) * 64 + lane - A_PIECES) * 16;      // weight-block byte offset of slot 5 (if it is a weight slot)
  const bool tail_ok = (wave != 3) || (lane < 16);

  gcptr gp[6], gpb;            // DMA cursors (run one chunk ahead of the MFMAs): slots 0..5, weight slots 6..9 = gpb + 4096 (j - 5)
  int ginc[6];
  int binc = B_BYTES;
  int ub = 0, uy0 = 0, ux0 = 0, unt = 0;      // unit the DMA cursor points into

  auto setup_unit = [&](int U) {
    const int v_ = xcd_remap(U, nunits);
    unt = __builtin_amdgcn_readfirstlane(v_ % ntn);
    const int t_ = v_ / ntn;
    ux0 = __builtin_amdgcn_readfirstlane((t_ % tiles_x) * TW);
    uy0 = __builtin_amdgcn_readfirstlane(((t_ / tiles_x) % tiles_y) * TH);
    ub = __builtin_amdgcn_readfirstlane(t_ / (tiles_x * tiles_y));
    const gcptr wb_ = wq + (size_t)unt * nchunk * B_BYTES;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int y = uy0 + (hyx[j] >> 16) - 1, x = ux0 + ((hyx[j] >> 8) & 255) - 1, als = hyx[j] & 255;
      const bool in = y >= 0 && y < H && x >= 0 && x < W;
      const gcptr pa = in ? srcp + (size_t)ub * splanes * plane_b + (unsigned)((y * W + x) * 64 + als) : zpage + als;
      gp[j] = (hyx[j] >= 0) ? pa : wb_ + boff0;
      ginc[j] = (hyx[j] >= 0) ? (in ? plane_b : 0) : B_BYTES;
    }
    gpb = wb_ + boff0;
    binc = B_BYTES;
  };
  auto setup_dead = [&]() {      // no unit left for this block: the cursor idles on the zero page
#pragma unroll
    for (int j = 0; j < 6; ++j) { gp[j] = zpage + ((lane & 3) << 4); ginc[j] = 0; }
    gpb = wq - 4096 + (lane << 4);                  // slots 6..9 re-read the first KBs of the weight pack (harmless: stage is never read)
    binc = 0;
  };
  auto issue_slot = [&](auto jc, int stg) {
    constexpr int J = decltype(jc)::value;
    char* const d_ = lds + stg * STAGE_PITCH + wave * 1024 + J * 4096;
    if (!(((S16_ABL) & 2) && J < 5) && !(((S16_ABL) & 4) && J > 5)) {
      const gcptr g_ = ((S16_ABL) & 1) ? wq + J * 1024 + lane * 16 : (J < 6) ? gp[J < 6 ? J : 0] : gpb + (J - 5) * 4096;
      if (J < NSLOT - 1) glds16(g_, d_);
      else if (tail_ok) glds16(g_, d_);
    }
    if (J < 6) gp[J < 6 ? J : 0] += ginc[J < 6 ? J : 0];
    if (J == NSLOT - 1) gpb += binc;
  };
#define S16_ISSUE(J, STG) issue_slot(std::integral_constant<int, (J)>{}, (STG));

  // ---- fragment read offsets (bytes inside a stage) -----------------------------------------------------------------
  const int wm = wave;                                     // this wave's output rows: 2 wm, 2 wm + 1
  int fa_hi[3], fa_lo[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int x = li + dx, key = (x >> 2) & 3;
    const int rec = (2 * wm) * HWP + x;
    fa_hi[dx] = rec * 64 + ((half ^ key) << 4);
    fa_lo[dx] = rec * 64 + (((2 + half) ^ key) << 4);
  }
  const int fb = A_BYTES + half * 512 + li * 16;

  // bias / scale table (this launch's 32 * ntn channels) -> LDS, once per block
  if (tid < 64) {
    // y = act((acc / 2^11 + bias) * scale) = act(acc * ms + bs)
    const float sc_ = (tid < 32 * ntn) ? a.scale[tid] : 1.f, bi_ = (tid < 32 * ntn) ? a.bias[tid] : 0.f;
    reinterpret_cast<float*>(lds + TAB_OFF)[tid] = bi_ * sc_;
    reinterpret_cast<float*>(lds + TAB_OFF)[64 + tid] = sc_ * UNSPLIT;
  }

#if defined(S16_PROF)
  unsigned long long pw_vm = 0, pw_bar = 0, pw_epi = 0;
  const unsigned long long pw_t0 = __builtin_readcyclecounter();
#endif
  int u = blockIdx.x;
  if (u >= nunits) return;
  setup_unit(u);
  S16_ISSUE(0, 0) S16_ISSUE(1, 0) S16_ISSUE(2, 0) S16_ISSUE(3, 0) S16_ISSUE(4, 0)
  S16_ISSUE(5, 0) S16_ISSUE(6, 0) S16_ISSUE(7, 0) S16_ISSUE(8, 0) S16_ISSUE(9, 0)
  int g = 0;
  const float slope = a.act == 1 ? 0.f : a.act == 2 ? 0.2f : 1.f;
  const float alo_ = (slope == 0.f) ? -3.0e38f : -INFINITY;

  f32x16 acc[2], pacc[2];                         // running / previous unit's accumulators (epilogue deferred into the next unit)
  f32x4 rl1[2][4], rl2[2][4];                     // residual loads (issued in a unit's last chunk); rl1 <- rs2 * res1 + res2 at the next chunk
  int pb = 0, py0 = 0, px0 = 0, pnt = 0;
  bool pvalid = false;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[m][r] = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) rl1[m][q] = rl2[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // deferred epilogue of (m, gI): 8 channels 16 gI + {4 half .. 4 half + 3, 8 + 4 half ..} of pixel (py0 + 2 wm + m, px0 + li)
  auto epilogue_piece = [&](auto mc, auto gc) {
    constexpr int m = decltype(mc)::value, gI = decltype(gc)::value;
    const int x = px0 + li, y = py0 + 2 * wm + m;
    const bool ok = pvalid && y < H && x < W;
    const size_t pix = (size_t)((size_t)pb * H + (y < H ? y : H - 1)) * W + (x < W ? x : W - 1);
    const int cbase = pnt * 32 + 4 * half;
    f32x4 v[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = 2 * gI + qq;
      const f32x4 bs = *reinterpret_cast<const f32x4*>(lds + TAB_OFF + (cbase + 8 * q) * 4);
      const f32x4 ms = *reinterpret_cast<const f32x4*>(lds + TAB_OFF + 256 + (cbase + 8 * q) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = fmaf(pacc[m][4 * q + e], ms[e], bs[e]);
        t = act1(t, slope, alo_);
        if (RES == 1) t = t * a.rs1 + rl1[m][q][e];
        if (RES == 2) t = t * (a.rs1 * a.rs2) + rl1[m][q][e];
        v[qq][e] = t;
      }
    }
#if !defined(S16_NO_EPI)
    if (ok) {
      if (OUT32) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
          *reinterpret_cast<f32x4*>(a.out32 + pix * a.out32_cs + a.out32_c0 + cbase + 8 * (2 * gI + qq)) = v[qq];
      }
      if (OUT16) {
        char* const o = a.out16 + ((size_t)pb * a.out16_planes + a.out16_rec0 + 2 * pnt + gI) * plane_b + (size_t)((y * W + x) * 64 + half * 16);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = v[e >> 2][e & 3];
          hi[e] = (_Float16)t;
          lo[e] = (_Float16)(t - (float)hi[e]);
        }
        *reinterpret_cast<f16x8*>(o) = hi;
        *reinterpret_cast<f16x8*>(o + 32) = lo;
      }
    }
#else
    if (ok && v[0][0] == 123.456f) a.out32[0] = v[1][1];
#endif
  };

  // One 16-channel chunk. KIND 0: first chunk of a unit (runs the previous unit's deferred epilogue), 1: middle, 2: last
  // (prefetches this unit's residuals, points the DMA cursor at the next unit). Steps s = 3 dx + dy: 6 MFMAs each
  // (3 terms x 2 output rows) on W(dx, dy) and halo rows dy, dy + 1; the fragments of step s + 1 are read during step s.
  auto do_chunk = [&](auto kind, const int un) {
    constexpr int KIND = decltype(kind)::value;
    const int stg = g & 1;
#if defined(S16_PROF)
    const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
    // this wave's pieces of the chunk have landed (vmcnt), everyone's (barrier); every wave is done with the other stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(S16_PROF)
    const unsigned long long t1_ = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();
#if defined(S16_PROF)
    pw_vm += t1_ - t0_; pw_bar += __builtin_readcyclecounter() - t1_;
#endif
    if (KIND == 0 && RES == 2) {            // residual loads of the previous unit landed a chunk ago
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) rl1[m][q] = rl1[m][q] * a.rs2 + rl2[m][q];
    }
    if (KIND == 2) {
      if (RES > 0) {
        const int x = ux0 + li, xc = x < W ? x : W - 1;
        const int cbase = unt * 32 + 4 * half;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int y = uy0 + 2 * wm + m;
          const size_t pix = (size_t)((size_t)ub * H + (y < H ? y : H - 1)) * W + xc;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            rl1[m][q] = *reinterpret_cast<const f32x4*>(a.res1 + pix * a.res1_cs + a.res1_c0 + cbase + 8 * q);
            if (RES == 2) rl2[m][q] = *reinterpret_cast<const f32x4*>(a.res2 + pix * a.res2_cs + a.res2_c0 + cbase + 8 * q);
          }
        }
      }
      // remember this unit for its deferred epilogue, then move the cursor on
      pb = ub; py0 = uy0; px0 = ux0; pnt = unt;
      if (un < nunits) setup_unit(un); else setup_dead();
    }
    const char* const sb = lds + stg * STAGE_PITCH;
    const int so = stg ^ 1;
    __builtin_amdgcn_sched_barrier(0);
#define S16_W(DX, DY, PL) (*reinterpret_cast<const f16x8*>(sb + fb + ((DY) * 3 + (DX)) * 2048 + (PL) * 1024))
#define S16_P(DX, R, PL) (*reinterpret_cast<const f16x8*>(sb + ((PL) ? fa_lo[DX] : fa_hi[DX]) + (R) * (HWP * 64)))
    f16x8 w0 = S16_W(0, 0, 0), p0h = S16_P(0, 0, 0), p1h = S16_P(0, 1, 0);
    f16x8 w1 = S16_W(0, 0, 1), p0l = S16_P(0, 0, 1), p1l = S16_P(0, 1, 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int dx = s / 3, dy = s % 3;
      // fragments of the next step
      f16x8 nw0, nw1, nah, nal, nbh, nbl;
      int nld = 0;
      if (s < 8) {
        const int ndx = (s + 1) / 3, ndy = (s + 1) % 3;
        nw0 = S16_W(ndx, ndy, 0);
        nw1 = S16_W(ndx, ndy, 1);
        if (ndy == 0) {
          nah = S16_P(ndx, 0, 0); nbh = S16_P(ndx, 1, 0); nal = S16_P(ndx, 0, 1); nbl = S16_P(ndx, 1, 1);
          nld = 6;
        } else {
          nbh = S16_P(ndx, ndy + 1, 0); nbl = S16_P(ndx, ndy + 1, 1);
          nld = 4;
        }
      }
#if !defined(S16_NO_DMA)
      // 10 DMA instructions of the next chunk over the 9 steps
      if (s == 0) { S16_ISSUE(0, so) S16_ISSUE(1, so) }
      if (s == 1) S16_ISSUE(2, so)
      if (s == 2) S16_ISSUE(3, so)
      if (s == 3) S16_ISSUE(4, so)
      if (s == 4) S16_ISSUE(5, so)
      if (s == 5) S16_ISSUE(6, so)
      if (s == 6) S16_ISSUE(7, so)
      if (s == 7) S16_ISSUE(8, so)
      if (s == 8) S16_ISSUE(9, so)
#endif
      if (KIND == 0) {          // the previous unit's epilogue rides in steps 2, 4, 6, 8
        if (s == 2) epilogue_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if (s == 4) epilogue_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        if (s == 6) epilogue_piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        if (s == 8) epilogue_piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      }
      // 6 MFMAs: consecutive ones alternate the accumulators and share the weight fragment
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, p0h, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, p1h, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, p0h, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, p1h, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, p0l, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, p1l, acc[1], 0, 0, 0);
      // pipeline order: an LDS read behind each of the first MFMAs, the DMA behind the second one
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (k < nld) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (KIND == 0 && k == 5 && (s == 2 || s == 4 || s == 6 || s == 8)) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#if !defined(S16_NO_DMA)
        if (k == 1) { if (s == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#endif
      }
      if (s < 8) {
        w0 = nw0; w1 = nw1;
        if ((s + 1) % 3 == 0) { p0h = nah; p0l = nal; p1h = nbh; p1l = nbl; }
        else { p0h = p1h; p0l = p1l; p1h = nbh; p1l = nbl; }
      }
    }
#undef S16_W
#undef S16_P
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    ++g;
  };

  while (true) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const int un = u + gridDim.x;
    do_chunk(std::integral_constant<int, 0>{}, un);
    for (int c = 1; c + 1 < nchunk; ++c) do_chunk(std::integral_constant<int, 1>{}, un);
    do_chunk(std::integral_constant<int, 2>{}, un);
    {   // range check: an |a| >= 65504 input turns the accumulators it touches into inf / NaN
      float chk = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][r], 0.f, chk);
      if (__any(chk != chk)) {
        if (lane == 0) atomicOr(a.ovf, 1);
      }
    }
    pacc[0] = acc[0]; pacc[1] = acc[1];
    pvalid = true;
    u = un;
    if (u >= nunits) break;
  }
  // the last unit's epilogue
  {
#if defined(S16_PROF)
    const unsigned long long te0_ = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (RES == 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) rl1[m][q] = rl1[m][q] * a.rs2 + rl2[m][q];
    }
    epilogue_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    epilogue_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    epilogue_piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    epilogue_piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
#if defined(S16_PROF)
    pw_epi += __builtin_readcyclecounter() - te0_;
#endif
  }
#if defined(S16_PROF)
  if (a.dbg && lane == 0 && (blockIdx.x & 63) == 17) {
    atomicAdd(a.dbg + 0, pw_vm); atomicAdd(a.dbg + 1, pw_bar); atomicAdd(a.dbg + 2, __builtin_readcyclecounter() - pw_t0);
    atomicAdd(a.dbg + 3, pw_epi); atomicAdd(a.dbg + 4, 1ull);
  }
#endif
#undef S16_ISSUE
}

template <bool OUT32, bool OUT16, int RES>
static inline int launch_t(const Args& a, int ncu, long long nunits, hipStream_t st) {
  static bool attr = false;
  auto fn = conv_s16_kernel<OUT32, OUT16, RES>;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -2;
    attr = true;
  }
  const long long cap = 2LL * ncu;
  const unsigned grid = (unsigned)(nunits < cap ? nunits : cap);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(256), LDS_BYTES, st, a, (int)nunits);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

static inline int launch(const Args& a, int ncu, hipStream_t st) {
  if (!a.src || !a.wpack || !a.bias || !a.scale || !a.ovf || !a.zeros || a.nchunk < 2 || a.ntile_n < 1 || a.ntile_n > 2) return -1;
  if ((reinterpret_cast<uintptr_t>(a.src) & 63) || a.src_rec0 + a.nchunk > a.src_planes) return -1;
  if (a.out16 && ((reinterpret_cast<uintptr_t>(a.out16) & 63) || a.out16_rec0 + 2 * a.ntile_n > a.out16_planes)) return -1;
  if ((long long)a.H * a.W * 64 >= 0x7fffffffLL) return -6;
  if (a.out32 && (((a.out32_cs | a.out32_c0) & 3) || (reinterpret_cast<uintptr_t>(a.out32) & 15))) return -1;
  if (a.res1 && (((a.res1_cs | a.res1_c0) & 3) || (reinterpret_cast<uintptr_t>(a.res1) & 15))) return -1;
  if (a.res2 && (((a.res2_cs | a.res2_c0) & 3) || (reinterpret_cast<uintptr_t>(a.res2) & 15))) return -1;
  if (a.res2 && !a.res1) return -1;
  if (!a.out16 && !a.out32) return -1;
  if ((long long)a.B * a.H * a.W >= 0x7fffffffLL) return -6;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long long nunits = (long long)a.B * tiles_x * tiles_y * a.ntile_n;
  if (nunits < 1 || nunits > 0x7fffffffLL) return -1;
  const int res = a.res2 ? 2 : a.res1 ? 1 : 0;
  const bool o32 = a.out32 != nullptr, o16 = a.out16 != nullptr;
  if (o16 && !o32 && res == 0) return launch_t<false, true, 0>(a, ncu, nunits, st);      // RDB conv1..4
  if (o16 && o32 && res == 1) return launch_t<true, true, 1>(a, ncu, nunits, st);        // RDB conv5
  if (o16 && o32 && res == 2) return launch_t<true, true, 2>(a, ncu, nunits, st);        // RDB conv5 + RRDB skip
  if (!o16 && o32 && res == 0) return launch_t<true, false, 0>(a, ncu, nunits, st);
  if (!o16 && o32 && res == 1) return launch_t<true, false, 1>(a, ncu, nunits, st);
  if (!o16 && o32 && res == 2) return launch_t<true, false, 2>(a, ncu, nunits, st);
  if (o16 && o32 && res == 0) return launch_t<true, true, 0>(a, ncu, nunits, st);
  return -6;
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide variant for 64 output channels (RDB conv5): one 512-thread block per CU, unit = 16 rows x 32 px x 64 channels, wave w
// owns output rows 2 w, 2 w + 1 and BOTH 32-channel tiles (4 accumulators), so every staged activation piece feeds twice
// the MFMAs of the narrow kernel (45 instead of 98 activation bytes per MFMA) and the weights are shared by 16 rows.
// LDS: 2 stages x (39 KB activations (18 x 34 records, padded to 39 DMA instructions) + 36 KB weights) + table = 150.5 KB.
// The epilogue is not deferred (the accumulators already take 64 registers); residuals are prefetched in the last chunk.
namespace wide_consts {
constexpr int WTH = 16, WHH = WTH + 2;
constexpr int WA_REAL = WHH * HWP * 64;               // 39 168
constexpr int WA_PITCH = 39 * 1024;                  // 39 936: 2 496 pieces = 39 instructions (48 dead lanes read the zero page)
constexpr int WA_PIECES = WA_REAL / 16;               // 2 448
constexpr int WB2_BYTES = 2 * B_BYTES;               // 36 864 = 36 instructions
constexpr int WSTAGE = WA_PITCH + WB2_BYTES;           // 76 800 = 75 instructions
constexpr int WTAB_OFF = 2 * WSTAGE;                  // [64 bs][64 ms]
constexpr int WLDS_BYTES = 2 * WSTAGE + 512;          // 154 112
constexpr int WNSLOT = 10;                           // I = 8 j + wave < 75: j = 9 only for waves 0..2
}  // namespace wide_consts
using namespace wide_consts;

template <bool OUT32, bool OUT16, int RES>
__global__ __launch_bounds__(512, 2) void conv_s16w_kernel(const Args a, const int nunits) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, nchunk = a.nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + WTH - 1) / WTH;
  const int plane_b = H * W * 64;
  const int splanes = __builtin_amdgcn_readfirstlane(a.src_planes);

  // DMA slots: instruction I = 8 j + wave; I < 39 activation pieces, 39 <= I < 75 weight pieces. Waves 0..6: j = 0..4
  // activation, wave 7: j = 0..3; the rest weights (wave-uniform: no mixed instruction)
  const int na = (wave == 7) ? 4 : 5;                 // activation slots of this wave
  int hyx[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int p = (8 * j + wave) * 64 + lane;
    const int rec = p >> 2, hy = rec / HWP, hx = rec - hy * HWP;
    hyx[j] = (p < WA_PIECES) ? ((hy << 16) | (hx << 8) | (((p & 3) ^ ((hx >> 2) & 3)) << 4)) : (-256 | ((p & 3) << 4));   // dead lane: zero page
  }
  const gcptr srcp = uniform_ptr(a.src + (size_t)a.src_rec0 * plane_b);
  const gcptr wq = uniform_ptr(a.wpack);
  const gcptr zpage = uniform_ptr(a.zeros);
  const int boff = ((8 * 4 + wave) * 64 + lane) * 16 - WA_PITCH;          // weight-stage byte offset of slot 4 seen as a weight slot

  gcptr gp[5], gpb;
  int ginc[5];
  int binc = WB2_BYTES;
  int ub = 0, uy0 = 0, ux0 = 0;

  auto setup_unit = [&](int U) {
    const int t_ = xcd_remap(U, nunits);
    ux0 = __builtin_amdgcn_readfirstlane((t_ % tiles_x) * TW);
    uy0 = __builtin_amdgcn_readfirstlane(((t_ / tiles_x) % tiles_y) * WTH);
    ub = __builtin_amdgcn_readfirstlane(t_ / (tiles_x * tiles_y));
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int y = uy0 + (hyx[j] >> 16) - 1, x = ux0 + ((hyx[j] >> 8) & 255) - 1, als = hyx[j] & 255;
      const bool in = hyx[j] >= 0 && y >= 0 && y < H && x >= 0 && x < W;
      gp[j] = in ? srcp + (size_t)ub * splanes * plane_b + (unsigned)((y * W + x) * 64 + als) : zpage + als;
      ginc[j] = in ? plane_b : 0;
    }
    gpb = wq + boff;
    binc = WB2_BYTES;
  };
  auto setup_dead = [&]() {
#pragma unroll
    for (int j = 0; j < 5; ++j) { gp[j] = zpage + ((lane & 3) << 4); ginc[j] = 0; }
    gpb = wq + boff;                                   // re-reads the first chunk's weights into a stage nobody reads
    binc = 0;
  };
  // slot J of the next chunk -> stage STG. Weight slots: byte offset (8 (J - 4) * 1024) from slot 4's weight position
  auto issue_slot = [&](auto jc, int stg) {
    constexpr int J = decltype(jc)::value;
    char* const d_ = lds + stg * WSTAGE + (8 * J + wave) * 1024;
    if (J < 4) { glds16(gp[J < 5 ? J : 0], d_); gp[J < 5 ? J : 0] += ginc[J < 5 ? J : 0]; }
    else if (J == 4) {
      if (na == 5) { glds16(gp[4], d_); gp[4] += ginc[4]; }
      else glds16(gpb, d_);
    } else if (J < 9) glds16(gpb + (J - 4) * 8192, d_);
    else { if (wave < 3) glds16(gpb + 5 * 8192, d_); gpb += binc; }
  };
#define S16W_ISSUE(J, STG) issue_slot(std::integral_constant<int, (J)>{}, (STG));

  const int wm = wave;
  int fa_hi[3], fa_lo[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int x = li + dx, key = (x >> 2) & 3;
    const int rec = (2 * wm) * HWP + x;
    fa_hi[dx] = rec * 64 + ((half ^ key) << 4);
    fa_lo[dx] = rec * 64 + (((2 + half) ^ key) << 4);
  }
  const int fb = WA_PITCH + half * 512 + li * 16;

  if (tid < 64) {
    const float sc_ = a.scale[tid], bi_ = a.bias[tid];
    reinterpret_cast<float*>(lds + WTAB_OFF)[tid] = bi_ * sc_;
    reinterpret_cast<float*>(lds + WTAB_OFF)[64 + tid] = sc_ * UNSPLIT;
  }

  int u = blockIdx.x;
  if (u >= nunits) return;
  setup_unit(u);
  S16W_ISSUE(0, 0) S16W_ISSUE(1, 0) S16W_ISSUE(2, 0) S16W_ISSUE(3, 0) S16W_ISSUE(4, 0)
  S16W_ISSUE(5, 0) S16W_ISSUE(6, 0) S16W_ISSUE(7, 0) S16W_ISSUE(8, 0) S16W_ISSUE(9, 0)
  int g = 0;
  const float slope = a.act == 1 ? 0.f : a.act == 2 ? 0.2f : 1.f;
  const float alo_ = (slope == 0.f) ? -3.0e38f : -INFINITY;

  f32x16 acc[2][2];                               // [output row][32-channel tile]
  f32x4 rl1[2][8];                                // residual 1 of this unit (prefetched in its last chunk)
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 8; ++q) rl1[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one 16-channel chunk; LAST: prefetch this unit's residual 1, point the DMA cursor at the next unit
  auto do_chunk = [&](auto lastc, const int un, int& eb, int& ey0, int& ex0) {
    constexpr bool LAST = decltype(lastc)::value;
    const int stg = g & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (LAST) {
      eb = ub; ey0 = uy0; ex0 = ux0;
      if (RES > 0) {
        const int x = ux0 + li, xc = x < W ? x : W - 1;
        const int y = uy0 + 2 * wm;                 // row 0 only: row 1's residual is read at the start of the epilogue (registers)
        const size_t pix = (size_t)((size_t)ub * H + (y < H ? y : H - 1)) * W + xc;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          rl1[0][q] = *reinterpret_cast<const f32x4*>(a.res1 + pix * a.res1_cs + a.res1_c0 + 4 * half + 8 * q);
      }
      if (un < nunits) setup_unit(un); else setup_dead();
    }
    const char* const sb = lds + stg * WSTAGE;
    const int so = stg ^ 1;
    __builtin_amdgcn_sched_barrier(0);
#define S16W_W(DX, DY, NT, PL) (*reinterpret_cast<const f16x8*>(sb + fb + (NT) * B_BYTES + ((DY) * 3 + (DX)) * 2048 + (PL) * 1024))
#define S16W_P(DX, R, PL) (*reinterpret_cast<const f16x8*>(sb + ((PL) ? fa_lo[DX] : fa_hi[DX]) + (R) * (HWP * 64)))
    f16x8 w00 = S16W_W(0, 0, 0, 0), p0h = S16W_P(0, 0, 0), p1h = S16W_P(0, 1, 0), w10 = S16W_W(0, 0, 1, 0);
    f16x8 w01 = S16W_W(0, 0, 0, 1), w11 = S16W_W(0, 0, 1, 1), p0l = S16W_P(0, 0, 1), p1l = S16W_P(0, 1, 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      f16x8 n00, n01, n10, n11, nah, nal, nbh, nbl;
      int nld = 0;
      if (s < 8) {
        const int ndx = (s + 1) / 3, ndy = (s + 1) % 3;
        n00 = S16W_W(ndx, ndy, 0, 0);
        if (ndy == 0) { nah = S16W_P(ndx, 0, 0); nbh = S16W_P(ndx, 1, 0); }
        else nbh = S16W_P(ndx, ndy + 1, 0);
        n10 = S16W_W(ndx, ndy, 1, 0);
        n01 = S16W_W(ndx, ndy, 0, 1);
        n11 = S16W_W(ndx, ndy, 1, 1);
        if (ndy == 0) { nal = S16W_P(ndx, 0, 1); nbl = S16W_P(ndx, 1, 1); nld = 8; }
        else { nbl = S16W_P(ndx, ndy + 1, 1); nld = 6; }
      }
#if !defined(S16_NO_DMA)
      if (s == 0) { S16W_ISSUE(0, so) S16W_ISSUE(1, so) }
      if (s == 1) S16W_ISSUE(2, so)
      if (s == 2) S16W_ISSUE(3, so)
      if (s == 3) S16W_ISSUE(4, so)
      if (s == 4) S16W_ISSUE(5, so)
      if (s == 5) S16W_ISSUE(6, so)
      if (s == 6) S16W_ISSUE(7, so)
      if (s == 7) S16W_ISSUE(8, so)
      if (s == 8) S16W_ISSUE(9, so)
#endif
      // 12 MFMAs: (term, tile, row); consecutive ones share the weight fragment and hit different accumulators
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w00, p0h, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w00, p1h, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w10, p0h, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w10, p1h, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w01, p0h, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w01, p1h, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w11, p0h, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w11, p1h, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w00, p0l, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w00, p1l, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w10, p0l, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w10, p1l, acc[1][1], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (k < nld) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#if !defined(S16_NO_DMA)
        if (k == 8) { if (s == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#endif
      }
      if (s < 8) {
        w00 = n00; w01 = n01; w10 = n10; w11 = n11;
        if ((s + 1) % 3 == 0) { p0h = nah; p0l = nal; p1h = nbh; p1l = nbl; }
        else { p0h = p1h; p0l = p1l; p1h = nbh; p1l = nbl; }
      }
    }
#undef S16W_W
#undef S16W_P
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    ++g;
  };

  while (true) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int un = u + gridDim.x;
    int eb = 0, ey0 = 0, ex0 = 0;
    for (int c = 0; c + 1 < nchunk; ++c) do_chunk(std::false_type{}, un, eb, ey0, ex0);
    do_chunk(std::true_type{}, un, eb, ey0, ex0);
    // ---- epilogue: lane (pixel li, k-half `half`) holds channels 32 nt + (r&3) + 8 (r>>2) + 4 half of rows 2 wm, 2 wm + 1 ----
    {
      float chk = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][n][r], 0.f, chk);
      if (__any(chk != chk)) {
        if (lane == 0) atomicOr(a.ovf, 1);
      }
    }
    const int x = ex0 + li, xc = x < W ? x : W - 1;
    if (RES >= 1) {
      const int y = ey0 + 2 * wm + 1;
      const size_t pix = (size_t)((size_t)eb * H + (y < H ? y : H - 1)) * W + xc;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        rl1[1][q] = *reinterpret_cast<const f32x4*>(a.res1 + pix * a.res1_cs + a.res1_c0 + 4 * half + 8 * q);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y = ey0 + 2 * wm + m;
      const bool ok = y < H && x < W;
      const size_t pix = (size_t)((size_t)eb * H + (y < H ? y : H - 1)) * W + xc;
      f32x4 r2[8];
      if (RES == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) r2[q] = *reinterpret_cast<const f32x4*>(a.res2 + pix * a.res2_cs + a.res2_c0 + 4 * half + 8 * q);
      }
#pragma unroll
      for (int gI = 0; gI < 4; ++gI) {        // 16-channel group: tile gI >> 1, half-tile gI & 1
        f32x4 v[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = 2 * gI + qq;          // float4 unit q: channels 8 q + 4 half .. + 3
          const f32x4 bs = *reinterpret_cast<const f32x4*>(lds + WTAB_OFF + (8 * q + 4 * half) * 4);
          const f32x4 ms = *reinterpret_cast<const f32x4*>(lds + WTAB_OFF + 256 + (8 * q + 4 * half) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = fmaf(acc[m][gI >> 1][4 * (q & 3) + e], ms[e], bs[e]);
            t = act1(t, slope, alo_);
            if (RES >= 1) t = t * a.rs1 + rl1[m][q][e];
            if (RES == 2) t = t * a.rs2 + r2[q][e];
            v[qq][e] = t;
          }
        }
#if !defined(S16_NO_EPI)
        if (ok) {
          if (OUT32) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
              *reinterpret_cast<f32x4*>(a.out32 + pix * a.out32_cs + a.out32_c0 + 4 * half + 8 * (2 * gI + qq)) = v[qq];
          }
          if (OUT16) {
            char* const o = a.out16 + ((size_t)eb * a.out16_planes + a.out16_rec0 + gI) * plane_b + (size_t)((y * W + x) * 64 + half * 16);
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t = v[e >> 2][e & 3];
              hi[e] = (_Float16)t;
              lo[e] = (_Float16)(t - (float)hi[e]);
            }
            *reinterpret_cast<f16x8*>(o) = hi;
            *reinterpret_cast<f16x8*>(o + 32) = lo;
          }
        }
#else
        if (ok && v[0][0] == 123.456f) a.out32[0] = v[1][1];
#endif
      }
    }
    u = un;
    if (u >= nunits) break;
  }
#undef S16W_ISSUE
}

template <bool OUT32, bool OUT16, int RES>
static inline int launch_w(const Args& a, int ncu, long long nunits, hipStream_t st) {
  static bool attr = false;
  auto fn = conv_s16w_kernel<OUT32, OUT16, RES>;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, WLDS_BYTES) != hipSuccess) return -2;
    attr = true;
  }
  const unsigned grid = (unsigned)(nunits < ncu ? nunits : ncu);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(512), WLDS_BYTES, st, a, (int)nunits);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// 64 output channels, weights packed chunk-major (pack_weights_s16(..., true)); same argument checks as launch()
static inline int launch_wide(const Args& a, int ncu, hipStream_t st) {
  if (!a.src || !a.wpack || !a.bias || !a.scale || !a.ovf || !a.zeros || a.nchunk < 1 || a.ntile_n != 2) return -1;
  if ((reinterpret_cast<uintptr_t>(a.src) & 63) || a.src_rec0 + a.nchunk > a.src_planes) return -1;
  if (a.out16 && ((reinterpret_cast<uintptr_t>(a.out16) & 63) || a.out16_rec0 + 4 > a.out16_planes)) return -1;
  if ((long long)a.H * a.W * 64 >= 0x7fffffffLL) return -6;
  if (a.out32 && (((a.out32_cs | a.out32_c0) & 3) || (reinterpret_cast<uintptr_t>(a.out32) & 15))) return -1;
  if (a.res1 && (((a.res1_cs | a.res1_c0) & 3) || (reinterpret_cast<uintptr_t>(a.res1) & 15))) return -1;
  if (a.res2 && (((a.res2_cs | a.res2_c0) & 3) || (reinterpret_cast<uintptr_t>(a.res2) & 15))) return -1;
  if ((a.res2 && !a.res1) || (!a.out16 && !a.out32)) return -1;
  if ((long long)a.B * a.H * a.W >= 0x7fffffffLL) return -6;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + WTH - 1) / WTH;
  const long long nunits = (long long)a.B * tiles_x * tiles_y;
  if (nunits < 1 || nunits > 0x7fffffffLL) return -1;
  const int res = a.res2 ? 2 : a.res1 ? 1 : 0;
  const bool o32 = a.out32 != nullptr, o16 = a.out16 != nullptr;
  if (o16 && o32 && res == 1) return launch_w<true, true, 1>(a, ncu, nunits, st);
  if (o16 && o32 && res == 2) return launch_w<true, true, 2>(a, ncu, nunits, st);
  if (o16 && o32 && res == 0) return launch_w<true, true, 0>(a, ncu, nunits, st);
  if (!o16 && o32 && res == 0) return launch_w<true, false, 0>(a, ncu, nunits, st);
  if (!o16 && o32 && res == 1) return launch_w<true, false, 1>(a, ncu, nunits, st);
  if (o16 && !o32 && res == 0) return launch_w<false, true, 0>(a, ncu, nunits, st);
  return -6;
}
#endif  // __HIPCC__

}  // namespace s16
}  // namespace hcf
