// Fused implicit-GEMM convolution (3x3 pad 1 / 1x1) on fp32 MFMA for gfx950.
//
// This one kernel carries ~99 % of the FLOPs of the HCFlow forward/inverse pass: the RRDB trunks
// (Basic.py:360-398), the coupling networks FCN / DenseBlock (Basic.py:329-356,426-447), conv_first /
// trunk_conv1 / prior head of ConditionalFlow (ConditionalFlow.py:28-41).
//
// GEMM view per tap: M = pixels, N = output channels, K = input channels.
//   * block = 256 threads = 4 waves, output tile 8 rows x 32 cols of pixels x (32*NT) channels;
//     wave w owns tile rows {2w, 2w+1} -> 2 x NT accumulators of v_mfma_f32_32x32x2_f32
//     (exact fp32, 64 cycles each: the kernel is MFMA-bound by design).
//   * K is walked in chunks of 16 "virtual" channels. The input halo tile (10 x 34 pixels x 16 ch,
//     21.8 KB) is staged through LDS once per chunk and reused by all 9 taps; double buffered
//     -> one barrier per chunk (18 K-steps x 16 MFMAs per wave between barriers at NT = 2).
//   * A fragments: lane (i = l&31, half = l>>5) reads 4 consecutive channels with one ds_read_b128
//     (the K order inside an 8-channel group is permuted so each lane's 4 k's are contiguous).
//   * B fragments (weights) skip LDS: host-packed [chunk][tap][kg][n][8] so that a wave's
//     ds-free global_load_dwordx4 covers 1 KB contiguous; identical for every block -> L1/L2 hits.
//   * inputs are up to 3 channel windows (dense concat is never materialised: RDB growth slab,
//     cat(z1, u), cat(z, up2(cf), up4(cf)) are just extra sources; nearest upsample = index shift).
//   * epilogue fuses bias, ActNorm / exp(3 logs) scale, ReLU / LeakyReLU(0.2) and two scaled
//     residual adds, and writes at a channel offset of the destination slab.
#include "hcf_common.h"

namespace hcf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// explicit global address space: a pointer rebuilt from SGPR halves would otherwise be "generic" and
// its loads become flat_load, which also tick lgkmcnt and so serialise with every LDS wait
typedef const float __attribute__((address_space(1)))* gfptr;
typedef const f32x4 __attribute__((address_space(1)))* gf4ptr;

constexpr int KC = 16;   // virtual channels per LDS stage
constexpr int TH = 8;
constexpr int TW = 32;

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  // contiguous tile ranges per XCD (block b runs on XCD b % 8): neighbouring tiles share halo
  // rows/cols through the same private L2. Bijective for any nwg (cdna guide 5.5 T1).
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

template <int TAPS, int NT, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  constexpr int HH = TH + 2 * PAD, HW = TW + 2 * PAD, HP = HH * HW;
  constexpr int NLOAD = HP * (KC / 4);
  constexpr int NSLOT = (NLOAD + 255) / 256;
  constexpr int NPAD = NT * 32;
  constexpr int KSTEPS = TAPS * 2;                 // (tap, 8-channel group) steps per chunk
  // the epilogue re-uses the staging LDS for the transposed output tile (16-byte stores, float4 residual reads) where
  // that costs no occupancy: 32-channel tiles fit as is, the 3x3 64-channel tile grows 43.5 -> 64 KB (2 blocks/CU either way)
  constexpr bool VEC_EPI = (NT == 1) || (NT == 2 && TAPS == 9);
  constexpr int E_FLOATS = VEC_EPI ? TH * TW * NPAD : 0;
  constexpr int LDS_FLOATS = (2 * HP * KC > E_FLOATS) ? 2 * HP * KC : E_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds_raw[LDS_FLOATS];
  float (*const lds)[HP * KC] = reinterpret_cast<float (*)[HP * KC]>(lds_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
#ifdef HCF_CONV_TIMERS      // clock probe of tools/conv_bench.py: measurement builds only (make TIMERS=1)
  const unsigned long long dbg_c0 = a.dbg ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long dbg_r0 = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
  const int H = a.H, W = a.W;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int txb = bid % tiles_x;
  const int tyb = (bid / tiles_x) % tiles_y;
  const int b = bid / (tiles_x * tiles_y);
  const int x0 = txb * TW, y0 = tyb * TH;

  // ---- per-thread staging slots: which halo pixel each of my float4 loads belongs to.
  // Addresses are clamped into the image so every load is unconditional (no divergent branches
  // around VMEM); out-of-image halo pixels (the conv's zero padding) are masked afterwards.
  int pos[NSLOT];
  unsigned okmask = 0;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int q = tid + 256 * s;
    const int hp = min(q >> 2, HP - 1);
    const int hy = hp / HW, hx = hp - hy * HW;
    const int y = y0 + hy - PAD, x = x0 + hx - PAD;
    const bool ok = y >= 0 && y < H && x >= 0 && x < W;
    okmask |= ok ? (1u << s) : 0u;
    pos[s] = (min(max(y, 0), H - 1) << 16) | min(max(x, 0), W - 1);
  }
  const int uq = tid & 3;                          // my 4-channel unit inside every chunk
  const int u0 = (a.src[0].n + 3) >> 2;
  const int u1 = u0 + ((a.nsrc > 1) ? ((a.src[1].n + 3) >> 2) : 0);
  const int u2 = u1 + ((a.nsrc > 2) ? ((a.src[2].n + 3) >> 2) : 0);

  // Source windows as SGPR values. readfirstlane makes them opaque SSA scalars: otherwise LLVM
  // folds select(load kernarg A, load kernarg B) into a per-lane load(select(&A, &B)), i.e. a
  // dependent VMEM access (and a vmcnt(0) drain) in front of every chunk's staging loads.
  const gfptr sp0 = uniform_ptr(a.src[0].p + a.src[0].c0);
  const gfptr sp1 = uniform_ptr((a.nsrc > 1) ? a.src[1].p + a.src[1].c0 : a.src[0].p);
  const gfptr sp2 = uniform_ptr((a.nsrc > 2) ? a.src[2].p + a.src[2].c0 : a.src[0].p);
  const int cs0 = __builtin_amdgcn_readfirstlane(a.src[0].cs), cs1 = __builtin_amdgcn_readfirstlane(a.src[1].cs),
            cs2 = __builtin_amdgcn_readfirstlane(a.src[2].cs);
  const int up0 = __builtin_amdgcn_readfirstlane(a.src[0].up), up1 = __builtin_amdgcn_readfirstlane(a.src[1].up),
            up2 = __builtin_amdgcn_readfirstlane(a.src[2].up);
  const int n0 = __builtin_amdgcn_readfirstlane(a.src[0].n), n1 = __builtin_amdgcn_readfirstlane(a.src[1].n),
            n2 = __builtin_amdgcn_readfirstlane(a.src[2].n);

  int stg_valid = 0;
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
    const int valid = uok ? (nn - 4 * ul) : 0; /* >=4 whole unit, 1..3 partial, 0 beyond K */     \
    sp = uok ? sp + 4 * ul : sp0;                                                                 \
    const int Hs = H >> ups, Ws = W >> ups;                                                       \
    _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                           \
      const int y = (pos[s] >> 16) >> ups, x = (pos[s] & 0xffff) >> ups;                          \
      gfptr p = sp + ((size_t)((size_t)b * Hs + y) * Ws + x) * css;                               \
      f32x4 v;                                                                                    \
      if (VEC) {                                                                                  \
        v = *(gf4ptr)(p);                                                                         \
      } else { /* windows that are not 16-byte aligned */                                         \
        v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];                                           \
      }                                                                                           \
      stg[s] = v; /* RAW load result: consuming it here would make the wave wait for HBM now */   \
    }                                                                                             \
    stg_valid = valid;                                                                            \
  }
#define HCF_STAGE_WRITE(BUF)                                                                      \
  {                                                                                               \
    _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                           \
      const int q = tid + 256 * s;                                                                \
      f32x4 v = stg[s];                                                                           \
      const bool ok = (okmask >> s) & 1u; /* conv zero padding + channel tail of the window */    \
      v.x = (ok && stg_valid > 0) ? v.x : 0.f;                                                    \
      v.y = (ok && stg_valid > 1) ? v.y : 0.f;                                                    \
      v.z = (ok && stg_valid > 2) ? v.z : 0.f;                                                    \
      v.w = (ok && stg_valid > 3) ? v.w : 0.f;                                                    \
      if (q < NLOAD) *reinterpret_cast<f32x4*>(&lds[(BUF)][q * 4]) = v;                           \
    }                                                                                             \
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // B stream: one K-step = NPAD*8 floats; this lane's float4 sits at (n*8 + half*4)
  const float* wp = a.wpack + (size_t)li * 8 + half * 4;
  f32x4 bcur[NT], bnxt[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) bcur[n] = *reinterpret_cast<const f32x4*>(wp + n * 32 * 8);
  wp += NPAD * 8;

  // A fragment base (floats) inside a stage buffer for tile row 2*wave + m
  const int abase = ((2 * wave) * HW + li) * KC + half * 4;

  HCF_STAGE_LOAD(0);
  HCF_STAGE_WRITE(0);
  __syncthreads();

  const int nchunk = a.nchunk;
  for (int c = 0; c < nchunk; ++c) {
    const float* A = lds[c & 1];
    const bool more = (c + 1 < nchunk);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      // prefetch the next K-step's weights (wpack is padded by one zero step at the end)
#pragma unroll
      for (int n = 0; n < NT; ++n) bnxt[n] = *reinterpret_cast<const f32x4*>(wp + n * 32 * 8);
      wp += NPAD * 8;
      if (s == 0 && more) {
        HCF_STAGE_LOAD(c + 1);                     // global loads fly under this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);         // ... provided they are ISSUED here and not sunk to the LDS write
      }
      const int tap = s >> 1, kg = s & 1;
      const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap % 3 : 0;
      f32x4 af[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        af[m] = *reinterpret_cast<const f32x4*>(A + abase + ((m + dy) * HW + dx) * KC + kg * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][j], bcur[n][j], acc[m][n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) bcur[n] = bnxt[n];
    }
    if (more) HCF_STAGE_WRITE((c + 1) & 1);
    __syncthreads();
  }

#ifdef HCF_CONV_TIMERS
  if (a.dbg && (blockIdx.x & 1023) == 512 && tid == 0) {   // a few mid-grid blocks: shader clock vs 100 MHz reference
    atomicAdd(a.dbg + 0, __builtin_readcyclecounter() - dbg_c0);
    atomicAdd(a.dbg + 1, __builtin_amdgcn_s_memrealtime() - dbg_r0);
  }
#endif

  // ---- epilogue --------------------------------------------------------------------------------
  // Residual reads go out RB at a time before the first is consumed (see hcf_conv_f16x3.hip: one s_waitcnt per value
  // serialised 32 NT memory latencies per residual); out-of-tile lanes read a clamped address and never store.
  const int cout = a.out.n;
  const bool has1 = a.res1.p != nullptr, has2 = a.res2.p != nullptr;
  constexpr int RB = 8;
  const float slope = act_slope(a.act);
  if constexpr (VEC_EPI) {
    if (a.vec_epi) {
      __syncthreads();                               // (the K loop ends on a barrier; kept explicit for the LDS re-use)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int oc = n * 32 + li;
        const float bias = a.bias[oc], scale = a.scale[oc];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = (2 * wave + m) * TW + (r & 3) + 8 * (r >> 2) + 4 * half;
            lds_raw[px * NPAD + oc] = apply_act((acc[m][n][r] + bias) * scale, slope);
          }
      }
      __syncthreads();
      constexpr int C4 = NPAD / 4;
      const int n4 = cout >> 2;
#pragma unroll 2
      for (int k = 0; k < (TH * TW * C4) / 256; ++k) {
        const int idx = tid + 256 * k;
        const int px = idx / C4, c4 = idx - px * C4;
        const int y = y0 + (px >> 5), x = x0 + (px & 31);
        f32x4 v = *reinterpret_cast<const f32x4*>(lds_raw + px * NPAD + 4 * c4);
        if (y < H && x < W && c4 < n4) {
          const size_t pixo = (size_t)((size_t)b * H + y) * W + x;
          if (has1) v = v * a.rs1 + *reinterpret_cast<const f32x4*>(a.res1.p + pixo * a.res1.cs + a.res1.c0 + 4 * c4);
          if (has2) v = v * a.rs2 + *reinterpret_cast<const f32x4*>(a.res2.p + pixo * a.res2.cs + a.res2.c0 + 4 * c4);
          *reinterpret_cast<f32x4*>(a.out.p + pixo * a.out.cs + a.out.c0 + 4 * c4) = v;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int oc = n * 32 + li;
    const bool ocok = oc < cout;
    const int occ = ocok ? oc : 0;
    const float bias = a.bias[oc], scale = a.scale[oc];     // arrays are padded to NPAD
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y = y0 + 2 * wave + m;
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
            float v = (acc[m][n][r] + bias) * scale;
            v = apply_act(v, slope);
            if (has1) v = v * a.rs1 + r1[q];
            if (has2) v = v * a.rs2 + r2[q];
            a.out.p[pix * a.out.cs + a.out.c0 + oc] = v;
          }
        }
      }
    }
  }
}

template <int TAPS, int NT>
static int launch_t(const ConvArgs& a, hipStream_t st) {
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long long nblk = (long long)a.B * tiles_x * tiles_y;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return HCF_ERR_ARG;
  bool vec = true;
  for (int i = 0; i < a.nsrc; ++i) {
    vec = vec && (((a.src[i].cs | a.src[i].c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.src[i].p) & 15) == 0);
  }
  ConvArgs b = a;
  auto v4 = [](const View& v) { return !v.p || ((((v.cs | v.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0)); };
  b.vec_epi = a.out.p && (a.out.n & 3) == 0 && v4(a.out) && v4(a.res1) && v4(a.res2);
  if (vec)
    hipLaunchKernelGGL((conv_mfma_kernel<TAPS, NT, true>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  else
    hipLaunchKernelGGL((conv_mfma_kernel<TAPS, NT, false>), dim3((unsigned)nblk), dim3(256), 0, st, b);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

int launch_conv(const ConvArgs& a, int taps, hipStream_t st) {
  if (a.nsrc < 1 || a.nsrc > kMaxSrc || a.H >= 32768 || a.W >= 32768 || a.H < 1 || a.W < 1) return HCF_ERR_ARG;
  for (int i = 0; i < a.nsrc; ++i)
    if ((a.H >> a.src[i].up) << a.src[i].up != a.H || (a.W >> a.src[i].up) << a.src[i].up != a.W) return HCF_ERR_ARG;
  const int nt = (a.out.n + 31) / 32;
  if (taps == 9) {
    switch (nt) {
      case 1: return launch_t<9, 1>(a, st);
      case 2: return launch_t<9, 2>(a, st);
      case 3: return launch_t<9, 3>(a, st);
    }
  } else if (taps == 1) {
    switch (nt) {
      case 1: return launch_t<1, 1>(a, st);
      case 2: return launch_t<1, 2>(a, st);
      case 3: return launch_t<1, 3>(a, st);
    }
  }
  return HCF_ERR_UNSUPPORTED;
}

}  // namespace hcf
