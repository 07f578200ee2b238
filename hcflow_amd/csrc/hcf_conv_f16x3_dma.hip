// f16x3 3x3 convolution whose activation sources are ALREADY stored in split-f16 form, staged by LDS-DMA.
//
// Split tensor format ("split16"): same NHWC addressing and byte size as fp32 (4 bytes per element), but every
// aligned group of 16 channels of a pixel holds 64 bytes [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] of f16, with
// hi = f16(x), lo = f16(x - hi) exactly as hcf_conv_f16x3.hip forms them in registers. A producer writes it once in
// its epilogue; every consumer then needs NO VGPR staging, NO split VALU work and NO ds_write: a 16-byte piece
// goes from HBM/L2 straight into LDS with global_load_lds_dwordx4, weights (already packed) likewise.
//
// Block = 512 threads (8 waves), output tile 16 rows x 32 px x 32*NTB channels, ONE block per CU, both the
// activation halo tile and the weight chunk double-buffered in LDS (2 x (41.5 + 18.4*NTB) KB): the DMA of chunk
// c+1 is issued before the 27*MT MFMAs of chunk c and has that whole phase to land; one barrier per chunk.
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

template <int NTB>
__global__ __launch_bounds__(NTHR, 1) void conv_f16x3_dma_kernel(const ConvArgs a) {
  constexpr int NPAD = 32 * NTB, MT = (NTB == 2) ? 4 : 2;
  constexpr int BHALF = NPAD * 16;
  constexpr int B_BYTES = 9 * 4 * BHALF;                  // [tap][plane][k-half][n][8 halves]
  constexpr int B_PIECES = B_BYTES / 16;
  constexpr int B_SLOTS = (B_PIECES + NTHR - 1) / NTHR;
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (NTB == 2) ? (wave >> 1) : wave;
  const int wn = (NTB == 2) ? (wave & 1) : 0;
  const int H = a.H, W = a.W;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int txb = bid % tiles_x;
  const int tyb = (bid / tiles_x) % tiles_y;
  const int b = bid / (tiles_x * tiles_y);
  const int x0 = txb * TW, y0 = tyb * TH;

  // ---- DMA source bookkeeping: slot q = tid + 512 s of the LDS image <- piece pj of halo pixel (hy, hx) --------
  int pix[A_SLOTS];            // pixel index in the image, -1: zero padding, -2: hole / beyond the image (no DMA)
  int pj[A_SLOTS];             // byte offset of the fetched piece inside the pixel's 64-byte chunk record
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int q = tid + NTHR * s;
    const int p = q >> 2;
    const int hy = p / PITCH, hx = p - hy * PITCH;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    const bool used = q < A_PIECES && hx < HW;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    pix[s] = !used ? -2 : in ? (b * H + y) * W + x : -1;
    pj[s] = (((q & 3) ^ ((hx >> 2) & 3)) << 4);
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
  const int dbgbits = __builtin_amdgcn_readfirstlane(a.stagger);   // timing ablations: 1 no DMA in the loop, 2 fragments read at tap 0 only

#define HCF_DMA_ISSUE(CHUNK, STG)                                                                 \
  {                                                                                               \
    const int c_ = (CHUNK);                                                                       \
    const bool in0 = c_ < k0, in1 = c_ < k1;                                                      \
    const gcptr sp = (in0 ? sp0 : in1 ? sp1 : sp2) + (size_t)(in0 ? c_ : in1 ? (c_ - k0) : (c_ - k1)) * 64; \
    const unsigned csb = in0 ? csb0 : in1 ? csb1 : csb2;                                          \
    char* const dstA = lds + (STG) * STAGE + wave * 1024;                                         \
    _Pragma("unroll") for (int s = 0; s < A_SLOTS; ++s) {                                         \
      if (pix[s] != -2) {                                                                         \
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

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

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

  if ((dbgbits >> 2) && blockIdx.x < 256) {     // experiment: de-phase the CUs (first round only; equal-length blocks stay staggered)
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, all 32 bits
    const int n = (int)((hwid >> 8) & 7) * (dbgbits >> 2);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
  }
  const bool prof = a.dbg && (blockIdx.x & 127) == 64 && tid == 0;
  const unsigned long long pr0 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull, pc0 = prof ? __builtin_readcyclecounter() : 0ull;
  HCF_DMA_ISSUE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long pr1 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull;

  for (int c = 0; c < nchunk; ++c) {
    const int stg = c & 1;
    const bool more = c + 1 < nchunk;
    if (more && !(dbgbits & 1)) {
      if (stg) HCF_DMA_ISSUE(c + 1, 0) else HCF_DMA_ISSUE(c + 1, 1)
    }
    const char* const sbase = lds + stg * STAGE;
    __builtin_amdgcn_s_setprio(1);
    f16x8 ahi[MT], alo[MT];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t % 3;
      const char* bt = sbase + bbase + t * (4 * BHALF);
      const f16x8 b1 = *reinterpret_cast<const f16x8*>(bt);               // b_hi * 2^11
      const f16x8 b2 = *reinterpret_cast<const f16x8*>(bt + 2 * BHALF);   // b_lo * 2^11
      if (t == 0 || !(dbgbits & 2)) {
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
    if (!more) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk c+1 have landed ...
    __syncthreads();                                   // ... everyone's have, and everyone is done reading chunk c
  }
#undef HCF_DMA_ISSUE
  const unsigned long long pr2 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // ---- epilogue: the tile is transposed through LDS (free now) so that every lane stores 16 contiguous bytes -------
  // (64 scalar dword stores per lane took 11.7 us per block, longer than 2 of the 12 K chunks of an RDB conv5)
  const int oc = wn * 32 + li;
  const float bias = a.bias[oc], scale = a.scale[oc];
  const int act = a.act;
  float* const ldsT = reinterpret_cast<float*>(lds);
  __syncthreads();                                     // every wave is done with the staging buffers
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = (MT * wm + m) * TW + (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = (acc[m][r] * UNSPLIT + bias) * scale;
      if (act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == ACT_LRELU) v = (v >= 0.f) ? v : v * 0.2f;
      ldsT[px * NPAD + oc] = v;
    }
  }
  __syncthreads();
  constexpr int C4 = NPAD / 4;                          // float4 units per pixel
  const int n4 = a.out.n >> 2;
#pragma unroll
  for (int k = 0; k < (TH * TW * C4) / NTHR; ++k) {
    const int idx = tid + NTHR * k;
    const int px = idx / C4, c4 = idx - px * C4;
    const int y = y0 + (px >> 5), x = x0 + (px & 31);
    const f32x4 v = *reinterpret_cast<const f32x4*>(ldsT + px * NPAD + 4 * c4);
    if (y < H && x < W && c4 < n4) {
      const size_t pixo = (size_t)((size_t)b * H + y) * W + x;
      *reinterpret_cast<f32x4*>(a.out.p + pixo * a.out.cs + a.out.c0 + 4 * c4) = v;
    }
  }
  if (prof) {   // {prologue, chunk loop, epilogue} in 100 MHz ticks, shader cycles of the whole block, sample count
    const unsigned long long pr3 = __builtin_amdgcn_s_memrealtime();
    atomicAdd(a.dbg + 0, pr1 - pr0);
    atomicAdd(a.dbg + 1, pr2 - pr1);
    atomicAdd(a.dbg + 2, pr3 - pr2);
    atomicAdd(a.dbg + 3, __builtin_readcyclecounter() - pc0);
    atomicAdd(a.dbg + 4, 1ull);
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

int launch_max_abs_diff(const float* a, const float* b, size_t n, unsigned* out_bits, hipStream_t st) {
  hipLaunchKernelGGL(f16x3dma::max_abs_diff_kernel, dim3(2048), dim3(256), 0, st, a, b, n, out_bits);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// all sources split16 (n, cs, c0 multiples of 16), 3x3, <= 64 output channels, no upsampled source
int launch_conv_f16x3_dma(const ConvArgs& a, hipStream_t st) {
  if (a.nsrc < 1 || a.nsrc > kMaxSrc || !a.wpack || !a.zeros || a.tC > 0 || a.w2 || a.in_max) return HCF_ERR_UNSUPPORTED;
  int ktot = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    if ((a.src[i].n & 15) || (a.src[i].cs & 15) || (a.src[i].c0 & 15) || a.src[i].up) return HCF_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a.src[i].p) & 63) != 0) return HCF_ERR_UNSUPPORTED;
    if ((long long)a.B * a.H * a.W >= 0x7fffffffLL) return HCF_ERR_UNSUPPORTED;
    ktot += a.src[i].n >> 4;
  }
  if (ktot != a.nchunk) return HCF_ERR_ARG;
  const int nt = (a.out.n + 31) / 32;
  if (nt < 1 || nt > 2) return HCF_ERR_UNSUPPORTED;
  const int tiles_x = (a.W + f16x3dma::TW - 1) / f16x3dma::TW, tiles_y = (a.H + f16x3dma::TH - 1) / f16x3dma::TH;
  const long long nblk = (long long)a.B * tiles_x * tiles_y;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return HCF_ERR_ARG;
  static bool attr_done = false;
  const int lds1 = 2 * (f16x3dma::A_BYTES + 9 * 4 * 32 * 16), lds2 = 2 * (f16x3dma::A_BYTES + 9 * 4 * 64 * 16);
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&f16x3dma::conv_f16x3_dma_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds1) != hipSuccess) return HCF_ERR_HIP;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&f16x3dma::conv_f16x3_dma_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds2) != hipSuccess) return HCF_ERR_HIP;
    attr_done = true;
  }
  if (nt == 1)
    hipLaunchKernelGGL((f16x3dma::conv_f16x3_dma_kernel<1>), dim3((unsigned)nblk), dim3(f16x3dma::NTHR), lds1, st, a);
  else
    hipLaunchKernelGGL((f16x3dma::conv_f16x3_dma_kernel<2>), dim3((unsigned)nblk), dim3(f16x3dma::NTHR), lds2, st, a);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace hcf
