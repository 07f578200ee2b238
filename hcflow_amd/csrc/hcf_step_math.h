// Per-pixel flow-step arithmetic shared by the stand-alone step kernels (hcf_flow.hip) and the fused conv
// epilogue (hcf_conv_f16x3.hip): a pixel's channel vector lives in registers (CMAX-sized arrays, compile-time
// indexing only), the C x C matrix is wave-uniform.
#pragma once
#include "hcf_common.h"

namespace hcf {

typedef float step_f32x4 __attribute__((ext_vector_type(4)));
#define f32x4_step step_f32x4

__device__ __forceinline__ float logscale_of(float s) {
  // 0.318 * atan(2 * scale)   (AffineCouplings.py:53,83)
  return 0.318f * atanf(2.f * s);
}

template <int CMAX>
__device__ __forceinline__ void load_pixel(const View& v, size_t pix, int C, float (&z)[CMAX]) {
  const float* p = v.p + pix * v.cs + v.c0;
  if (((v.cs | v.c0) & 3) == 0) {
#pragma unroll
    for (int c4 = 0; c4 < CMAX / 4; ++c4) {
      if (4 * c4 < C) {                      // cs = roundup4(C): the whole float4 is inside the pixel
        const step_f32x4 t = *reinterpret_cast<const step_f32x4*>(p + 4 * c4);
        z[4 * c4 + 0] = t.x;
        z[4 * c4 + 1] = (4 * c4 + 1 < C) ? t.y : 0.f;
        z[4 * c4 + 2] = (4 * c4 + 2 < C) ? t.z : 0.f;
        z[4 * c4 + 3] = (4 * c4 + 3 < C) ? t.w : 0.f;
      } else {
        z[4 * c4 + 0] = 0.f; z[4 * c4 + 1] = 0.f; z[4 * c4 + 2] = 0.f; z[4 * c4 + 3] = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < CMAX; ++c) z[c] = (c < C) ? p[c] : 0.f;
  }
}

template <int CMAX>
__device__ __forceinline__ void store_pixel(const View& v, size_t pix, int C, const float (&z)[CMAX]) {
  float* p = v.p + pix * v.cs + v.c0;
  if (((v.cs | v.c0) & 3) == 0 && (C & 3) == 0) {
#pragma unroll
    for (int c4 = 0; c4 < CMAX / 4; ++c4)
      if (4 * c4 < C) {
        step_f32x4 t = {z[4 * c4], z[4 * c4 + 1], z[4 * c4 + 2], z[4 * c4 + 3]};
        *reinterpret_cast<step_f32x4*>(p + 4 * c4) = t;
      }
  } else {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) p[c] = z[c];
  }
}

// channels [0, n) of a pixel's vector as a 16-channel record of their own, zero padded (n <= 16): the first source of the
// Winograd form of the NEXT step's FCN conv1 (hcf_engine.hip run_coupling_net)
template <int CMAX>
__device__ __forceinline__ void store_pad16(float* out16, size_t pix, int n, const float (&z)[CMAX]) {
  float* p = out16 + pix * 16;
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    step_f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (4 * c4 + e < CMAX && 4 * c4 + e < n) ? z[(4 * c4 + e < CMAX) ? 4 * c4 + e : 0] : 0.f;
    *reinterpret_cast<step_f32x4*>(p + 4 * c4) = t;
  }
}

// Wave-uniform read-only tables (the C x C matrix, ActNorm vectors) are read through the constant address space:
// scalar loads into SGPRs, which then feed the FMAs directly. Left to itself the compiler reads them with 16+
// uniform-address vector loads per pixel once the kernel also stores to global memory (it can no longer prove
// the table invariant).
#ifndef HCF_TAIL_VECTOR_TABLES
#define HCF_TAIL_VECTOR_TABLES 0   // 1 / 2: reproduce the round-1 fault (tables through per-lane vector loads; 2 adds a full wait)
#endif
#if HCF_TAIL_VECTOR_TABLES
typedef const float* step_cptr;
__device__ __forceinline__ step_cptr const_table(const float* p) { return p; }
#else
typedef const float __attribute__((address_space(4)))* step_cptr;
__device__ __forceinline__ step_cptr const_table(const float* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (step_cptr)(((uint64_t)hi << 32) | lo);
}
#endif

// y = M z with M row-major [CMAX][CMAX] (host pads rows/cols beyond C with zeros)
template <int CMAX>
__device__ __forceinline__ void matvec(step_cptr M, const float (&z)[CMAX], float (&y)[CMAX]) {
#if HCF_TAIL_VECTOR_TABLES == 2
  if (CMAX <= 12) {                          // experiment: every table load has landed before the first use
    float m[CMAX * CMAX];
#pragma unroll
    for (int i = 0; i < CMAX * CMAX; ++i) m[i] = M[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < CMAX; ++k) acc = fmaf(m[c * CMAX + k], z[k], acc);
      y[c] = acc;
    }
    return;
  }
#endif
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) acc = fmaf(M[c * CMAX + k], z[k], acc);
    y[c] = acc;
  }
}


// FlowStep.reverse_flow after the coupling network (FlowStep.py:53-64): coupling^-1 -> W^-1 -> actnorm^-1 on one
// pixel. `hp` points at the pixel's coupling-net output (global memory or LDS), stride 1.
template <int CMAX, typename HPtr>
__device__ __forceinline__ void step_tail_inverse_pixel(float (&z)[CMAX], HPtr hp, int C, int ns, int mode,
                                                        const float* __restrict__ mat,
                                                        const float* __restrict__ an_bias,
                                                        const float* __restrict__ an_mul, float (&y)[CMAX]) {
  if (mode == CPL_AFFINE) {
    // z2 = z2 * exp(-logscale) - shift, (shift, scale) = h[0::2], h[1::2]  (AffineCouplings.py:65-87)
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c >= ns && c < C) {
        const int j = c - ns;
        const float shift = hp[2 * j], scale = hp[2 * j + 1];
        z[c] = z[c] * expf(-logscale_of(scale)) - shift;
      }
    }
  } else {
    // AffineCoupling3shift, LRvsothers=False: z[:3] -= f(z[3:])  (AffineCouplings.py:150-153)
#pragma unroll
    for (int c = 0; c < 3; ++c) z[c] = z[c] - hp[c];
  }
  if (mat) {
    matvec<CMAX>(const_table(mat), z, y);
  } else {
#pragma unroll
    for (int c = 0; c < CMAX; ++c) y[c] = z[c];
  }
  // actnorm reverse: x * exp(-logs) - bias  (ActNorms.py:54,66)
  const step_cptr mul = const_table(an_mul), bias = const_table(an_bias);
#pragma unroll
  for (int c = 0; c < CMAX; ++c) y[c] = y[c] * mul[c] - bias[c];
}

}  // namespace hcf
