// f16x3 3x3 convolution whose activation sources are ALREADY stored in split-f16 form, staged by LDS-DMA.
//
// Split tensor format ("split16"): same NHWC addressing and byte size as fp32 (4 bytes per element), but every
// aligned group of 16 channels of a pixel holds 64 bytes [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] of f16, with
// hi = f16(x), lo = f16(x - hi) exactly as hcf_conv_f16x3.hip forms them in registers. A producer writes it once in
// its epilogue; every consumer then needs NO VGPR staging, NO split VALU work and NO ds_write: a 16-byte piece
// goes from HBM/L2 straight into LDS with global_load_lds_dwordx4, weights (already packed) likewise.
//
// Block = 512 threads (8 waves), output tile 16 rows x 32 px x 32*NTB channels, ONE persistent block per CU that
// walks tiles blockIdx, blockIdx + gridDim, ...; both the activation halo tile and the weight chunk are
// double-buffered in LDS (2 x (41.5 + 18.4*NTB) KB): the DMA of chunk c+1 -- or of the NEXT tile's first chunk --
// is issued before the 27*MT MFMAs of chunk c and has that whole phase (and the epilogue) to land; one barrier
// per chunk. Block launch, the first-chunk latency and the store drain are paid once per block, not per tile.
//   * LDS activation image: 18 rows x 36 px (34 used: the DMA destination is lane-linear, holes are simply
//     skipped) x 4 pieces of 16 B; piece j of pixel (r, x) sits at slot (36 r + x) * 4 + (j ^ ((x >> 2) & 3)).
//     The XOR is applied on the SOURCE side (which piece a lane fetches) and on the fragment read address, so
//     the ds_read_b128 fragment reads of a 32-pixel row are bank-conflict free without padding; the key depends
//     on x only, so all fragment addresses are "per-lane base + immediate".
//   * Waves: NTB = 2: wave = (4-row group, 32-channel half), MT = 4; NTB = 1: 8 waves x 2 rows, MT = 2.
//   * Same arithmetic, same accumulation order per output as hcf_conv_f16x3.hip: results are bit-identical to it.
#include "hcf_common.h"

namespace hcf {
namespace f16x3dma {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef const char __attribute__((address_space(1)))* gcptr;
typedef __attribute__((address_space(3))) void* lptr;

constexpr int TW = 32, TH = 16, HH = TH + 2, HW = TW + 2, PITCH = 36;
constexpr int NTHR = 512;
constexpr int A_BYTES = HH * PITCH * 64;                  // 41472
constexpr int A_PIECES = A_BYTES / 16;                    // 2592
constexpr int A_SLOTS = (A_PIECES + NTHR - 1) / NTHR;     // 6 (the last one: 32 lanes)
constexpr float UNSPLIT = 1.f / 2048.f;

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

__device__ __forceinline__ gcptr uniform_ptr(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gcptr)(((uint64_t)hi << 32) | lo);
}

// one 16-byte piece per lane: global (per-lane address) -> LDS at wave-uniform base + lane * 16
__device__ __forceinline__ void glds16(gcptr g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lptr)lds_wave_base, 16, 0, 0);
}

// 4 consecutive channels (float4 unit c4) of a residual view at pixel pixo: fp32, or rebuilt from a split16 tensor
__device__ __forceinline__ f32x4 load_res4(const View& r, size_t pixo, int c4, int soff) {
  const float* const p = r.p + pixo * r.cs + r.c0;
  if (r.fmt) {
    const f16x4 hi = *reinterpret_cast<const f16x4*>(p + soff), lo = *reinterpret_cast<const f16x4*>(p + soff + 8);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)hi[e] + (float)lo[e];
    return v;
  }
  return *reinterpret_cast<const f32x4*>(p + 4 * c4);
}

template <int NTB>
__global__ __launch_bounds__(NTHR, 1) void conv_f16x3_dma_kernel(const ConvArgs a, const int ntiles) {
  constexpr int NPAD = 32 * NTB, MT = (NTB == 2) ? 4 : 2;
  constexpr int BHALF = NPAD * 16;
  constexpr int B_BYTES = 9 * 4 * BHALF;                  // [tap][plane][k-half][n][8 halves]
  constexpr int B_PIECES = B_BYTES / 16;
  constexpr int B_SLOTS = (B_PIECES + NTHR - 1) / NTHR;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int C4 = NPAD / 4;                            // float4 units per pixel of the output tile
  static_assert(STAGE >= (TH / 2) * TW * NPAD * 4, "half an output tile (fp32) must fit one staging stage");
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (NTB == 2) ? (wave >> 1) : wave;
  const int wn = (NTB == 2) ? (wave & 1) : 0;
  const int H = a.H, W = a.W;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;

  // ---- DMA source bookkeeping: slot q = tid + 512 s of the LDS image <- piece pj of halo pixel (hy, hx) --------
  int hyx[A_SLOTS];            // (hy << 8) | hx of the slot's halo pixel, -1: hole in the 36-px pitch / past the image (no DMA)
  int pj[A_SLOTS];             // byte offset of the fetched piece inside the pixel's 64-byte chunk record
  int pix[A_SLOTS];            // current DMA tile: pixel index in the image, -1: zero padding
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int q = tid + NTHR * s;
    const int p = q >> 2;
    const int hy = p / PITCH, hx = p - hy * PITCH;
    hyx[s] = (q < A_PIECES && hx < HW) ? ((hy << 8) | hx) : -1;
    pj[s] = (((q & 3) ^ ((hx >> 2) & 3)) << 4);
    pix[s] = -1;
  }
  const int k0 = __builtin_amdgcn_readfirstlane(a.src[0].n >> 4);
  const int k1 = k0 + __builtin_amdgcn_readfirstlane((a.nsrc > 1) ? (a.src[1].n >> 4) : 0);
  const gcptr sp0 = uniform_ptr(a.src[0].p + a.src[0].c0);
  const gcptr sp1 = uniform_ptr((a.nsrc > 1) ? a.src[1].p + a.src[1].c0 : a.src[0].p);
  const gcptr sp2 = uniform_ptr((a.nsrc > 2) ? a.src[2].p + a.src[2].c0 : a.src[0].p);
  const int csb0 = __builtin_amdgcn_readfirstlane(a.src[0].cs) * 4, csb1 = __builtin_amdgcn_readfirstlane(a.src[1].cs) * 4,
            csb2 = __builtin_amdgcn_readfirstlane(a.src[2].cs) * 4;
  const gcptr wq = uniform_ptr(a.wpack);
  const gcptr zpage = uniform_ptr(a.zeros);
  const int nchunk = a.nchunk;
  const int dbgbits = __builtin_amdgcn_readfirstlane(a.dbg_bits);   // timing ablations: 1 no DMA in the loop, 2 fragments read at tap 0 only, 4 no global stores, 8 no LDS transposition writes

  // tile id -> (image, tile row, tile col); XCD-aware: the tiles one XCD works on are neighbours in memory
#define HCF_TILE_COORDS(T, TB, TY0, TX0)                                                          \
  {                                                                                               \
    const int bid_ = xcd_remap((T), ntiles);                                                      \
    const int txb_ = bid_ % tiles_x;                                                              \
    const int tyb_ = (bid_ / tiles_x) % tiles_y;                                                  \
    TB = __builtin_amdgcn_readfirstlane(bid_ / (tiles_x * tiles_y));                              \
    TY0 = __builtin_amdgcn_readfirstlane(tyb_ * TH);                                              \
    TX0 = __builtin_amdgcn_readfirstlane(txb_ * TW);                                              \
  }
#define HCF_DMA_SET_TILE(TB, TY0, TX0)                                                            \
  {                                                                                               \
    _Pragma("unroll") for (int s = 0; s < A_SLOTS; ++s) {                                         \
      const int y = (TY0) + (hyx[s] >> 8) - 1, x = (TX0) + (hyx[s] & 255) - 1;                    \
      const bool in = y >= 0 && y < H && x >= 0 && x < W;                                         \
      pix[s] = in ? ((TB) * H + y) * W + x : -1;                                                  \
    }                                                                                             \
  }
#define HCF_DMA_ISSUE(CHUNK, STG)                                                                 \
  {                                                                                               \
    const int c_ = (CHUNK);                                                                       \
    const bool in0 = c_ < k0, in1 = c_ < k1;                                                      \
    const gcptr sp = (in0 ? sp0 : in1 ? sp1 : sp2) + (size_t)(in0 ? c_ : in1 ? (c_ - k0) : (c_ - k1)) * 64; \
    const unsigned csb = in0 ? csb0 : in1 ? csb1 : csb2;                                          \
    char* const dstA = lds + (STG) * STAGE + wave * 1024;                                         \
    _Pragma("unroll") for (int s = 0; s < A_SLOTS; ++s) {                                         \
      if (hyx[s] >= 0) {                                                                          \
        const gcptr g = (pix[s] >= 0) ? sp + (size_t)((unsigned)pix[s]) * csb + pj[s] : zpage + pj[s]; \
        glds16(g, dstA + s * (NTHR * 16));                                                        \
      }                                                                                           \
    }                                                                                             \
    char* const dstB = lds + (STG) * STAGE + A_BYTES + wave * 1024;                               \
    const gcptr wb = wq + (size_t)c_ * B_BYTES + tid * 16;                                        \
    _Pragma("unroll") for (int s = 0; s < B_SLOTS; ++s) {                                         \
      if (tid + NTHR * s < B_PIECES) glds16(wb + s * (NTHR * 16), dstB + s * (NTHR * 16));        \
    }                                                                                             \
  }

  // fragment read bases (bytes into a stage): pixel (row MT*wm, x = li + dx), piece hi = half, lo = 2 + half
  int abase_hi[3], abase_lo[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int x = li + dx, key = (x >> 2) & 3;
    const int pl = (MT * wm) * PITCH + x;
    abase_hi[dx] = (pl * 4 + (half ^ key)) * 16;
    abase_lo[dx] = (pl * 4 + ((2 + half) ^ key)) * 16;
  }
  const int bbase = A_BYTES + half * BHALF + (wn * 32 + li) * 16;

  const int oc = wn * 32 + li;
  const float bias = a.bias[oc], scale = a.scale[oc];
  const float slope = act_slope(a.act);
  const int n4 = a.out.n >> 2;
  const bool h1 = a.res1.p != nullptr, h2 = a.res2.p != nullptr;

  const bool prof = a.dbg && (blockIdx.x & 31) == 16 && tid == 0;
  const unsigned long long pr0 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull, pc0 = prof ? __builtin_readcyclecounter() : 0ull;
  unsigned long long pr_epi = 0ull;
  int ntile_done = 0;

  int t = blockIdx.x;                     // the launcher guarantees gridDim.x <= ntiles
  int tb, ty0, tx0;                       // the tile being accumulated
  HCF_TILE_COORDS(t, tb, ty0, tx0)
  HCF_DMA_SET_TILE(tb, ty0, tx0)
  HCF_DMA_ISSUE(0, 0)
  int g = 0;                              // chunks consumed so far by this block: stage = g & 1

  while (true) {
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    int tn = t + gridDim.x;               // next tile of this block
    const int eb = tb, ey0 = ty0, ex0 = tx0;

    for (int c = 0; c < nchunk; ++c, ++g) {
      const int stg = g & 1;
      // chunk (t, c) has landed (this wave's pieces: vmcnt; everyone's: the barrier), and every wave is done reading
      // the other stage (previous chunk's fragments, or the previous tile's transposed output)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (!(dbgbits & 1)) {
        if (c + 1 < nchunk) {
          if (stg) HCF_DMA_ISSUE(c + 1, 0) else HCF_DMA_ISSUE(c + 1, 1)
        } else if (tn < ntiles) {          // first chunk of the NEXT tile: its latency hides behind this chunk + the epilogue
          HCF_TILE_COORDS(tn, tb, ty0, tx0)
          HCF_DMA_SET_TILE(tb, ty0, tx0)
          if (stg) HCF_DMA_ISSUE(0, 0) else HCF_DMA_ISSUE(0, 1)
        }
      }
      const char* const sbase = lds + stg * STAGE;
      __builtin_amdgcn_s_setprio(1);
      f16x8 ahi[MT], alo[MT];
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int dy = tp / 3, dx = tp % 3;
        const char* bt = sbase + bbase + tp * (4 * BHALF);
        const f16x8 b1 = *reinterpret_cast<const f16x8*>(bt);               // b_hi * 2^11
        const f16x8 b2 = *reinterpret_cast<const f16x8*>(bt + 2 * BHALF);   // b_lo * 2^11
        if (tp == 0 || !(dbgbits & 2)) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            ahi[m] = *reinterpret_cast<const f16x8*>(sbase + abase_hi[dx] + (m + dy) * (PITCH * 64));
            alo[m] = *reinterpret_cast<const f16x8*>(sbase + abase_lo[dx] + (m + dy) * (PITCH * 64));
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b1, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b2, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[m], b1, acc[m], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue of tile (eb, ey0, ex0) -----------------------------------------------------------------------------
    // Range check as in hcf_conv_f16x3.hip: an |a| >= 65504 input turns the accumulators it touches into inf / NaN.
    const unsigned long long pe0 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
    {
      float chk = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][r], 0.f, chk);
      if (__any(chk != chk)) {
        if (lane == 0) atomicOr(a.ovf, 1);
      }
    }
    // The tile is transposed through the stage that was just consumed (the other one is receiving the next tile's first
    // chunk), 8 rows at a time, so that every lane reads residuals / stores outputs as 16 contiguous bytes.
    float* const ldsT = reinterpret_cast<float*>(lds + ((g - 1) & 1) * STAGE);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      __syncthreads();                    // pass 0: every wave is done with the fragments; pass 1: with pass 0's tile half
      if ((MT * wm) / (TH / 2) == ps && !(dbgbits & 8)) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = ((MT * wm) % (TH / 2) + m) * TW + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = (acc[m][r] * UNSPLIT + bias) * scale;
            v = apply_act(v, slope);
            ldsT[px * NPAD + oc] = v;
          }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < ((TH / 2) * TW * C4) / NTHR; ++k) {
        const int idx = tid + NTHR * k;
        const int px = idx / C4, c4 = idx - px * C4;
        const int y = ey0 + ps * (TH / 2) + (px >> 5), x = ex0 + (px & 31);
        f32x4 v = *reinterpret_cast<const f32x4*>(ldsT + px * NPAD + 4 * c4);
        if (y < H && x < W && c4 < n4 && !(dbgbits & 4)) {
          const size_t pixo = (size_t)((size_t)eb * H + y) * W + x;
          const int soff = (c4 >> 2) * 16 + (c4 & 3) * 2;          // split16: float offset of the 4 hi halves of this unit
          if (h1) v = v * a.rs1 + load_res4(a.res1, pixo, c4, soff);
          if (h2) v = v * a.rs2 + load_res4(a.res2, pixo, c4, soff);
          float* const o = a.out.p + pixo * a.out.cs + a.out.c0;
          if (a.out.fmt) {
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              hi[e] = (_Float16)v[e];
              lo[e] = (_Float16)(v[e] - (float)hi[e]);
            }
            *reinterpret_cast<f16x4*>(o + soff) = hi;
            *reinterpret_cast<f16x4*>(o + soff + 8) = lo;
          } else {
            *reinterpret_cast<f32x4*>(o + 4 * c4) = v;
          }
        }
      }
    }
    if (prof) { pr_epi += __builtin_amdgcn_s_memrealtime() - pe0; ++ntile_done; }
    t = tn;
    if (t >= ntiles) break;
  }
#undef HCF_DMA_ISSUE
#undef HCF_DMA_SET_TILE
#undef HCF_TILE_COORDS
  if (prof) {   // {-, everything but epilogues, epilogues} in 100 MHz ticks, shader cycles of the block's life, tiles sampled
    const unsigned long long pr3 = __builtin_amdgcn_s_memrealtime();
    atomicAdd(a.dbg + 1, pr3 - pr0 - pr_epi);
    atomicAdd(a.dbg + 2, pr_epi);
    atomicAdd(a.dbg + 3, __builtin_readcyclecounter() - pc0);
    atomicAdd(a.dbg + 4, (unsigned long long)ntile_done);
  }
}

// fp32 NHWC -> split16 (same addressing): one thread per (pixel, 16-channel group); n and cs multiples of 16
__global__ void to_split16_kernel(View src, View dst, long long npix) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = src.n >> 4;
  if (i >= npix * groups) return;
  const long long p = i / groups;
  const int g = (int)(i - p * groups);
  const float* s = src.p + p * src.cs + src.c0 + 16 * g;
  _Float16* d = reinterpret_cast<_Float16*>(dst.p + p * dst.cs + dst.c0 + 16 * g);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float v = s[k];
    const _Float16 h = (_Float16)v;
    d[k] = h;
    d[16 + k] = (_Float16)(v - (float)h);
  }
}

// split16 -> fp32 (same addressing)
__global__ void from_split16_kernel(View src, View dst, long long npix) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = src.n >> 4;
  if (i >= npix * groups) return;
  const long long p = i / groups;
  const int g = (int)(i - p * groups);
  const _Float16* s = reinterpret_cast<const _Float16*>(src.p + p * src.cs + src.c0 + 16 * g);
  float* d = dst.p + p * dst.cs + dst.c0 + 16 * g;
#pragma unroll
  for (int k = 0; k < 16; ++k) d[k] = (float)s[k] + (float)s[16 + k];
}

__global__ void max_abs_diff_kernel(const float* a, const float* b, size_t n, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = fabsf(a[i] - b[i]);
    m = (d == d) ? fmaxf(m, d) : 3.0e38f;
  }
  atomicMax(out, __builtin_bit_cast(unsigned, m));
}

}  // namespace f16x3dma

int launch_to_split16(const View& src, const View& dst, int B, int H, int W, hipStream_t st) {
  if ((src.n & 15) || (src.cs & 15) || (src.c0 & 15) || (dst.cs & 15) || (dst.c0 & 15)) return HCF_ERR_ARG;
  const long long npix = (long long)B * H * W, tot = npix * (src.n >> 4);
  hipLaunchKernelGGL(f16x3dma::to_split16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, dst, npix);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

int launch_from_split16(const View& src, const View& dst, int B, int H, int W, hipStream_t st) {
  if ((src.n & 15) || (src.cs & 15) || (src.c0 & 15) || (dst.cs & 3) || (dst.c0 & 3)) return HCF_ERR_ARG;
  const long long npix = (long long)B * H * W, tot = npix * (src.n >> 4);
  hipLaunchKernelGGL(f16x3dma::from_split16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, dst, npix);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

int launch_max_abs_diff(const float* a, const float* b, size_t n, unsigned* out_bits, hipStream_t st) {
  hipLaunchKernelGGL(f16x3dma::max_abs_diff_kernel, dim3(2048), dim3(256), 0, st, a, b, n, out_bits);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// Every source split16 (n, cs, c0 multiples of 16, 64-byte aligned base, no upsampling), 3x3, <= 64 output channels whose
// view (and the residual views) are 16-byte addressable; split16 outputs / residuals need c0, cs multiples of 16.
// HCF_ERR_UNSUPPORTED when the launch does not qualify (the caller keeps the register-staged kernel for fp32 sources).
int launch_conv_f16x3_dma(const ConvArgs& a, hipStream_t st) {
  if (a.nsrc < 1 || a.nsrc > kMaxSrc || !a.wpack || !a.zeros || !a.ovf || a.tC > 0 || a.w2 || a.in_max) return HCF_ERR_UNSUPPORTED;
  int ktot = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    if (a.src[i].fmt != 1 || (a.src[i].n & 15) || (a.src[i].cs & 15) || (a.src[i].c0 & 15) || a.src[i].up) return HCF_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a.src[i].p) & 63) != 0) return HCF_ERR_UNSUPPORTED;
    ktot += a.src[i].n >> 4;
  }
  if ((long long)a.B * a.H * a.W >= 0x7fffffffLL) return HCF_ERR_UNSUPPORTED;
  if (ktot != a.nchunk) return HCF_ERR_ARG;
  auto ok_view = [](const View& v) {
    if (!v.p) return true;
    const int m = v.fmt ? 15 : 3;
    return ((v.cs | v.c0) & m) == 0 && (reinterpret_cast<uintptr_t>(v.p) & (v.fmt ? 63 : 15)) == 0;
  };
  if (!a.out.p || (a.out.n & (a.out.fmt ? 15 : 3)) || !ok_view(a.out) || !ok_view(a.res1) || !ok_view(a.res2)) return HCF_ERR_UNSUPPORTED;
  const int nt = (a.out.n + 31) / 32;
  if (nt < 1 || nt > 2) return HCF_ERR_UNSUPPORTED;
  const int tiles_x = (a.W + f16x3dma::TW - 1) / f16x3dma::TW, tiles_y = (a.H + f16x3dma::TH - 1) / f16x3dma::TH;
  const long long ntiles = (long long)a.B * tiles_x * tiles_y;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) return HCF_ERR_ARG;
  static int ncu = 0;
  const int lds1 = 2 * (f16x3dma::A_BYTES + 9 * 4 * 32 * 16), lds2 = 2 * (f16x3dma::A_BYTES + 9 * 4 * 64 * 16);
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      return HCF_ERR_HIP;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&f16x3dma::conv_f16x3_dma_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds1) != hipSuccess) return HCF_ERR_HIP;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&f16x3dma::conv_f16x3_dma_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds2) != hipSuccess) return HCF_ERR_HIP;
    ncu = n;
  }
  const unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);      // one persistent block per CU (LDS admits only one)
  if (nt == 1)
    hipLaunchKernelGGL((f16x3dma::conv_f16x3_dma_kernel<1>), dim3(grid), dim3(f16x3dma::NTHR), lds1, st, a, (int)ntiles);
  else
    hipLaunchKernelGGL((f16x3dma::conv_f16x3_dma_kernel<2>), dim3(grid), dim3(f16x3dma::NTHR), lds2, st, a, (int)ntiles);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace hcf
