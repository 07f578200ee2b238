// Backward kernels of the NLL training step (the reference reaches them through autograd in
// HCFlow_SR_model.optimize_parameters, HCFlow_SR_model.py:195-202). gfx950 only. Everything here is HBM-bound
// elementwise / per-pixel work; the GEMM-shaped parts of the backward pass are the data-gradient convs (forward
// kernels on transposed weight packs) and hcf_conv_wgrad.hip.
#include "hcf_common.h"
#include <algorithm>
#include <cstdlib>
#include "hcf_step_math.h"

namespace hcf {

#define HCF_RET_T() return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP

// ---- conv epilogue backward -----------------------------------------------------------------------------------
// thread -> (pixel group, channel) so that consecutive threads touch consecutive floats of the NHWC windows;
// per-channel sums: per-thread float partials -> LDS -> one atomicAdd per channel per block.
__global__ __launch_bounds__(256) void conv_epilogue_bwd_kernel(const EpiBwdArgs a, long long npix, int ppb) {
  __shared__ float sh[2][256];
  const int n = a.gy.n;
  const int groups = 256 / n;
  const int pg = threadIdx.x / n, c = threadIdx.x - pg * n;
  float s_pre = 0.f, s_zy = 0.f, mx = 0.f;
  if (pg < groups) {
    const float sc = a.scale ? a.scale[c] : 1.f;
    const float k1 = a.has2 ? a.rs2 : 1.f;                 // gy -> gradient of (res1 + rs1 * act(..))
    const float k2 = k1 * (a.has1 ? a.rs1 : 1.f);          // gy -> gradient of act(..)
    const bool need_y = a.act != ACT_NONE || a.sum_zy != nullptr;
    const long long p0 = (long long)blockIdx.x * ppb;
    const long long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    for (long long p = p0 + pg; p < p1; p += groups) {
      const float g = a.gy.p[(size_t)p * a.gy.cs + a.gy.c0 + c];
      if (a.has2 && a.g2.p) a.g2.p[(size_t)p * a.g2.cs + a.g2.c0 + c] += g;
      if (a.has1 && a.g1.p) a.g1.p[(size_t)p * a.g1.cs + a.g1.c0 + c] += g * k1;
      const float y = need_y ? a.y.p[(size_t)p * a.y.cs + a.y.c0 + c] : 0.f;
      float dz = g * k2;
      if (a.act == ACT_RELU) dz = (y > 0.f) ? dz : 0.f;
      else if (a.act == ACT_LRELU) dz = (y >= 0.f) ? dz : dz * 0.2f;
      const float gp = dz * sc;
      a.gpre.p[(size_t)p * a.gpre.cs + a.gpre.c0 + c] = gp;
      s_pre += gp;
      s_zy += dz * y;
      mx = fmaxf(mx, fabsf(gp));
    }
  }
  if (a.absmax) {                                  // non-negative floats order like their bit patterns
    __shared__ int shm;
    if (threadIdx.x == 0) shm = 0;
    __syncthreads();
    if (mx > 0.f && mx == mx) atomicMax(&shm, __builtin_bit_cast(int, mx));
    __syncthreads();
    if (threadIdx.x == 0) {
      if (shm) atomicMax(reinterpret_cast<int*>(a.absmax), shm);
      if (a.absmax2) {
        const int c = (a.carry2 && blockIdx.x == 0) ? max(shm, *reinterpret_cast<const int*>(a.carry2)) : shm;
        if (c) atomicMax(reinterpret_cast<int*>(a.absmax2), c);
      }
    }
  }
  if (a.sum_pre || a.sum_zy) {
    sh[0][threadIdx.x] = s_pre;
    sh[1][threadIdx.x] = s_zy;
    __syncthreads();
    if (threadIdx.x < n) {
      float t0 = 0.f, t1 = 0.f;
      for (int g = 0; g < groups; ++g) { t0 += sh[0][g * n + threadIdx.x]; t1 += sh[1][g * n + threadIdx.x]; }
      if (a.part) {                                  // fixed-order reduction later (launch_sum_jobs)
        a.part[((size_t)blockIdx.x * 2 + 0) * n + threadIdx.x] = t0;
        a.part[((size_t)blockIdx.x * 2 + 1) * n + threadIdx.x] = t1;
      } else {
        if (a.sum_pre) atomicAdd(a.sum_pre + threadIdx.x, t0);
        if (a.sum_zy) atomicAdd(a.sum_zy + threadIdx.x, t1 * a.zy_mult);
      }
    }
  }
}

// The same on 16-byte lanes (every window 4-float addressable, n % 4 == 0: all 32 / 64-channel convs of the trunks and FCNs):
// thread -> (pixel group, channel quad), two pixels in flight per thread. The scalar form above moved 1.5 TB/s at 16 x 80 x 80 x 32
// (26.6 us per launch, 582 launches per training step); per-block partial sums keep their layout (launch_sum_jobs).
typedef float epi_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conv_epilogue_bwd_vec_kernel(const EpiBwdArgs a, long long npix, int ppb) {
  __shared__ float sh[2][1024];
  const int n = a.gy.n, n4 = n >> 2;
  const int groups = 256 / n4;
  const int pg = threadIdx.x / n4, c = (threadIdx.x - pg * n4) * 4;
  epi_f32x4 s_pre = {0.f, 0.f, 0.f, 0.f}, s_zy = {0.f, 0.f, 0.f, 0.f};
  float mx = 0.f;
  if (pg < groups) {
    epi_f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if (a.scale) sc = *reinterpret_cast<const epi_f32x4*>(a.scale + c);
    const float k1 = a.has2 ? a.rs2 : 1.f;
    const float k2 = k1 * (a.has1 ? a.rs1 : 1.f);
    const bool need_y = a.act != ACT_NONE || a.sum_zy != nullptr;
    const bool w2 = a.has2 && a.g2.p, w1 = a.has1 && a.g1.p;
    const float neg = a.act == ACT_RELU ? 0.f : a.act == ACT_LRELU ? 0.2f : 1.f;
    const long long p0 = (long long)blockIdx.x * ppb;
    const long long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    for (long long p = p0 + pg; p < p1; p += 2 * groups) {
      const long long pb = p + groups;
      const bool okb = pb < p1;
      const long long pbc = okb ? pb : p;
      const epi_f32x4 ga = *reinterpret_cast<const epi_f32x4*>(a.gy.p + (size_t)p * a.gy.cs + a.gy.c0 + c);
      const epi_f32x4 gb = *reinterpret_cast<const epi_f32x4*>(a.gy.p + (size_t)pbc * a.gy.cs + a.gy.c0 + c);
      epi_f32x4 ya = {0.f, 0.f, 0.f, 0.f}, yb = ya;
      if (need_y) {
        ya = *reinterpret_cast<const epi_f32x4*>(a.y.p + (size_t)p * a.y.cs + a.y.c0 + c);
        yb = *reinterpret_cast<const epi_f32x4*>(a.y.p + (size_t)pbc * a.y.cs + a.y.c0 + c);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && !okb) break;
        const long long q = h ? pb : p;
        const epi_f32x4 g = h ? gb : ga, y = h ? yb : ya;
        if (w2) {
          epi_f32x4* d = reinterpret_cast<epi_f32x4*>(a.g2.p + (size_t)q * a.g2.cs + a.g2.c0 + c);
          *d = *d + g;
        }
        if (w1) {
          epi_f32x4* d = reinterpret_cast<epi_f32x4*>(a.g1.p + (size_t)q * a.g1.cs + a.g1.c0 + c);
          *d = *d + g * k1;
        }
        epi_f32x4 gp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dz = g[e] * k2;
          if (a.act == ACT_RELU) dz = (y[e] > 0.f) ? dz : 0.f;
          else if (a.act == ACT_LRELU) dz = (y[e] >= 0.f) ? dz : dz * 0.2f;
          gp[e] = dz * sc[e];
          s_pre[e] += gp[e];
          s_zy[e] += dz * y[e];
          mx = fmaxf(mx, fabsf(gp[e]));
        }
        (void)neg;
        *reinterpret_cast<epi_f32x4*>(a.gpre.p + (size_t)q * a.gpre.cs + a.gpre.c0 + c) = gp;
      }
    }
  }
  if (a.absmax) {
    __shared__ int shm;
    if (threadIdx.x == 0) shm = 0;
    __syncthreads();
    if (mx > 0.f && mx == mx) atomicMax(&shm, __builtin_bit_cast(int, mx));
    __syncthreads();
    if (threadIdx.x == 0) {
      if (shm) atomicMax(reinterpret_cast<int*>(a.absmax), shm);
      if (a.absmax2) {
        const int c = (a.carry2 && blockIdx.x == 0) ? max(shm, *reinterpret_cast<const int*>(a.carry2)) : shm;
        if (c) atomicMax(reinterpret_cast<int*>(a.absmax2), c);
      }
    }
  }
  if (a.sum_pre || a.sum_zy) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[0][threadIdx.x * 4 + e] = s_pre[e]; sh[1][threadIdx.x * 4 + e] = s_zy[e]; }
    __syncthreads();
    if ((int)threadIdx.x < n) {                       // channel ch: quad ch / 4 of every pixel group, fixed order
      const int ch = threadIdx.x, q4 = ch >> 2, e = ch & 3;
      float t0 = 0.f, t1 = 0.f;
      for (int g = 0; g < groups; ++g) { t0 += sh[0][(g * n4 + q4) * 4 + e]; t1 += sh[1][(g * n4 + q4) * 4 + e]; }
      if (a.part) {
        a.part[((size_t)blockIdx.x * 2 + 0) * n + ch] = t0;
        a.part[((size_t)blockIdx.x * 2 + 1) * n + ch] = t1;
      } else {
        if (a.sum_pre) atomicAdd(a.sum_pre + ch, t0);
        if (a.sum_zy) atomicAdd(a.sum_zy + ch, t1 * a.zy_mult);
      }
    }
  }
}

static int epi_bwd_ppb() {                   // pixels per block; 192: ~530 blocks for a 16 x 80 x 80 tensor (96 / 192 / 384 measured 84.3 / 82.6 / 86.8 ms per training step) (HCF_EPI_PPB: experiments)
  static const int v = getenv("HCF_EPI_PPB") ? std::max(32, atoi(getenv("HCF_EPI_PPB"))) : 192;
  return v;
}
int conv_epilogue_bwd_blocks(int B, int H, int W) {
  const long long npix = (long long)B * H * W;
  return (int)((npix + epi_bwd_ppb() - 1) / epi_bwd_ppb());
}

__global__ __launch_bounds__(256) void sum_jobs_kernel(const SumJob* jobs) {
  const SumJob j = jobs[blockIdx.x];
  for (int t = threadIdx.x; t < 2 * j.n; t += 256) {
    const int which = t / j.n, c = t - which * j.n;
    float* const dst = which ? j.dst1 : j.dst0;
    if (!dst) continue;
    const float* p = j.part + (size_t)which * j.pstride + c;
    double s = 0.0;
    for (int b = 0; b < j.nblk; ++b) s += (double)p[(size_t)b * 2 * j.pstride];
    dst[c] += (float)(which ? s * (double)j.mult1 : s);
  }
}

int launch_sum_jobs(const SumJob* jobs_dev, int njobs, hipStream_t st) {
  if (njobs <= 0) return HCF_OK;
  hipLaunchKernelGGL(sum_jobs_kernel, dim3((unsigned)njobs), dim3(256), 0, st, jobs_dev);
  HCF_RET_T();
}

int launch_conv_epilogue_bwd(const EpiBwdArgs& a, hipStream_t st) {
  if (a.gy.n < 1 || a.gy.n > 256 || a.gpre.n != a.gy.n) return HCF_ERR_ARG;
  const long long npix = (long long)a.B * a.H * a.W;
  const int ppb = epi_bwd_ppb();
  auto v4 = [](const View& v) { return !v.p || ((((v.cs | v.c0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0)); };
  const bool vec = (a.gy.n & 3) == 0 && a.gy.n >= 4 && v4(a.gy) && v4(a.gpre) && v4(a.y) && v4(a.g1) && v4(a.g2) &&
                   (!a.scale || (reinterpret_cast<uintptr_t>(a.scale) & 15) == 0);
  if (vec) hipLaunchKernelGGL(conv_epilogue_bwd_vec_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, st, a, npix, ppb);
  else hipLaunchKernelGGL(conv_epilogue_bwd_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, st, a, npix, ppb);
  HCF_RET_T();
}

// ---- coupling backward (AffineCouplings.py:44-63 forward; one thread per pixel) ---------------------------------
__global__ __launch_bounds__(256) void step_couple_bwd_kernel(const StepBwdArgs a) {
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const size_t pix = (size_t)blockIdx.y * hw + i;
  const float* gp = a.gzout.p + pix * a.gzout.cs + a.gzout.c0;
  const float* zp = a.zout.p + pix * a.zout.cs + a.zout.c0;
  const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
  float* ob = a.gzb.p + pix * a.gzb.cs + a.gzb.c0;
  float* oh = a.gh.p + pix * a.gh.cs + a.gh.c0;
  if (a.mode == CPL_AFFINE) {
    for (int c = 0; c < a.ns; ++c) ob[c] = gp[c];
    for (int c = a.ns; c < a.C; ++c) {
      const int j = c - a.ns;
      const float scale = hp[2 * j + 1];
      const float e = expf(logscale_of(scale));
      const float g = gp[c];
      ob[c] = g * e;                                             // z2' = (z2 + shift) e^ls
      oh[2 * j] = g * e;                                         // d shift
      const float dls = g * zp[c] + a.gobj;                      // d z2'/d ls = z2';  logdet += ls
      oh[2 * j + 1] = dls * (0.636f / (1.f + 4.f * scale * scale));   // ls = 0.318 atan(2 scale)
    }
  } else {                                                       // z[:3] += h
    for (int c = 0; c < a.C; ++c) ob[c] = gp[c];
    for (int c = 0; c < 3; ++c) oh[c] = gp[c];
  }
}

int launch_step_couple_bwd(const StepBwdArgs& a, hipStream_t st) {
  if (a.C < 1 || a.H < 1 || a.W < 1 || a.B < 1) return HCF_ERR_ARG;
  const dim3 grid((unsigned)step_blocks_per_sample(a.H, a.W), (unsigned)a.B);
  hipLaunchKernelGGL(step_couple_bwd_kernel, grid, dim3(256), 0, st, a);
  HCF_RET_T();
}

// ---- head backward: zb = W za, za = (zin + b) e^s  ->  gza = W^T gzb, gzin = gza e^s, sums for b and s -----------
// per-channel block sums in a fixed order: wave shuffle tree, then the four wave results added in wave order
__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
template <int CMAX>
__device__ __forceinline__ void block_channel_sums(const float (&v0)[CMAX], const float (&v1)[CMAX], int C, float (*shw)[2][CMAX],
                                                   float* g0, float* g1, float* part, int blk) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    const float s0 = wave_sum_t(v0[c]), s1 = wave_sum_t(v1[c]);
    if (lane == 0) { shw[w][0][c] = s0; shw[w][1][c] = s1; }
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    const float t0 = ((shw[0][0][c] + shw[1][0][c]) + shw[2][0][c]) + shw[3][0][c];
    const float t1 = ((shw[0][1][c] + shw[1][1][c]) + shw[2][1][c]) + shw[3][1][c];
    if (part) {
      part[((size_t)blk * 2 + 0) * CMAX + c] = t0;
      part[((size_t)blk * 2 + 1) * CMAX + c] = t1;
    } else {
      atomicAdd(g0 + c, t0);
      atomicAdd(g1 + c, t1);
    }
  }
}

template <int CMAX>
__global__ __launch_bounds__(256) void step_head_bwd_kernel(const StepBwdArgs a) {
  __shared__ float shw[4][2][CMAX];
  float v0[CMAX], v1[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) { v0[c] = 0.f; v1[c] = 0.f; }
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < hw) {
    const size_t pix = (size_t)blockIdx.y * hw + i;
    float g[CMAX], gza[CMAX], za[CMAX];
    load_pixel<CMAX>(a.gzb, pix, a.C, g);
    load_pixel<CMAX>(a.za, pix, a.C, za);
    if (a.matT) {
      matvec<CMAX>(const_table(a.matT), g, gza);
    } else {
#pragma unroll
      for (int c = 0; c < CMAX; ++c) gza[c] = g[c];
    }
    const step_cptr mul = const_table(a.an_mul);
    float gin[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      gin[c] = gza[c] * mul[c];
      if (c < a.C) {
        v0[c] = gin[c];                            // d bias  = sum gza e^s
        v1[c] = gza[c] * za[c];                    // d logs  = sum gza * za
      }
    }
    store_pixel<CMAX>(a.gzin, pix, a.C, gin);
  }
  block_channel_sums<CMAX>(v0, v1, a.C, shw, a.g_bias, a.g_logs, a.part, blockIdx.y * gridDim.x + blockIdx.x);
}

int launch_step_head_bwd(const StepBwdArgs& a, hipStream_t st) {
  if (a.C < 1 || a.H < 1 || a.W < 1 || a.B < 1 || !a.g_bias || !a.g_logs) return HCF_ERR_ARG;
  const dim3 grid((unsigned)step_blocks_per_sample(a.H, a.W), (unsigned)a.B);
  if (a.C <= 8) hipLaunchKernelGGL((step_head_bwd_kernel<8>), grid, dim3(256), 0, st, a);
  else if (a.C <= 12) hipLaunchKernelGGL((step_head_bwd_kernel<12>), grid, dim3(256), 0, st, a);
  else if (a.C <= 24) hipLaunchKernelGGL((step_head_bwd_kernel<24>), grid, dim3(256), 0, st, a);
  else if (a.C <= 48) hipLaunchKernelGGL((step_head_bwd_kernel<48>), grid, dim3(256), 0, st, a);
  else return HCF_ERR_UNSUPPORTED;
  HCF_RET_T();
}

// ---- Gaussian prior logp backward (Basic.py:78-93; SR: logs = s) --------------------------------------------------
__global__ __launch_bounds__(256) void gauss_logp_bwd_kernel(const PriorBwdArgs a) {
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const size_t pix = (size_t)blockIdx.y * hw + i;
  const float* ap = a.a.p + pix * a.a.cs + a.a.c0;
  const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
  float* ga = a.ga.p + pix * a.ga.cs + a.ga.c0;
  float* gh = a.gh.p + pix * a.gh.cs + a.gh.c0;
  for (int c = 0; c < a.C; ++c) {
    const float mean = hp[2 * c], logs = hp[2 * c + 1];
    const float d = ap[c] - mean, iv = expf(-2.f * logs);
    // logp = -1/2 (2 logs + d^2 e^{-2 logs} + ln 2pi)
    ga[c] = -a.gobj * d * iv;
    gh[2 * c] = a.gobj * d * iv;
    gh[2 * c + 1] = a.gobj * (d * d * iv - 1.f);
  }
}

int launch_gauss_logp_bwd(const PriorBwdArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)step_blocks_per_sample(a.H, a.W), (unsigned)a.B);
  hipLaunchKernelGGL(gauss_logp_bwd_kernel, grid, dim3(256), 0, st, a);
  HCF_RET_T();
}

// ---- Dirac term backward with the straight-through Quant (HCFlowNet_SR_arch.py:58-63, Basic.py:186-196) ----------
__global__ __launch_bounds__(256) void quant_logp_bwd_kernel(View z, const float* lr, View gz, int hw, float gobj) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const size_t pix = (size_t)blockIdx.y * hw + i;
  const float e12 = expf(12.f);
  for (int c = 0; c < 3; ++c) {
    const float v = z.p[pix * z.cs + z.c0 + c];
    const float q = rintf(fminf(fmaxf(v, 0.f), 1.f) * 255.f) / 255.f;
    const float l = lr[((size_t)blockIdx.y * 3 + c) * hw + i];
    // logp(x = q; mean = lr, logs = -6) = -1/2 (-12 + (q - lr)^2 e^12 + ln 2pi);  d/dq = -(q - lr) e^12
    gz.p[pix * gz.cs + gz.c0 + c] += gobj * (l - q) * e12;
  }
}

int launch_quant_logp_bwd(View z, const float* lr_nchw, View gz, int B, int H, int W, float gobj, hipStream_t st) {
  const dim3 grid((unsigned)step_blocks_per_sample(H, W), (unsigned)B);
  hipLaunchKernelGGL(quant_logp_bwd_kernel, grid, dim3(256), 0, st, z, lr_nchw, gz, H * W, gobj);
  HCF_RET_T();
}

// ---- inverse flow step backward (one thread per pixel) ---------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void step_inv_bwd_kernel(const StepInvBwdArgs a) {
  __shared__ float shw[4][2][CMAX];
  float v0[CMAX], v1[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) { v0[c] = 0.f; v1[c] = 0.f; }
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < hw) {
    const size_t pix = (size_t)blockIdx.y * hw + i;
    float gx[CMAX], x[CMAX], gy[CMAX], y[CMAX], gzc[CMAX];
    load_pixel<CMAX>(a.gx, pix, a.C, gx);
    load_pixel<CMAX>(a.x, pix, a.C, x);
    const step_cptr b = const_table(a.an_bias), mf = const_table(a.mul_fwd), mi = const_table(a.mul_inv);
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const float xb = x[c] + b[c];                 // = y e^-s
      y[c] = xb * mf[c];
      gy[c] = gx[c] * mi[c];
      if (c < a.C) {
        v0[c] = -gx[c];                             // x = y e^-s - b
        v1[c] = -gx[c] * xb;
      }
    }
    if (a.matInvT) {
      matvec<CMAX>(const_table(a.matInvT), gy, gzc);
    } else {
#pragma unroll
      for (int c = 0; c < CMAX; ++c) gzc[c] = gy[c];
    }
    store_pixel<CMAX>(a.y, pix, a.C, y);
    store_pixel<CMAX>(a.gzc, pix, a.C, gzc);
    // coupling^-1 backward: zc2 = z2 e^-ls - shift
    const float* zp = a.zc.p + pix * a.zc.cs + a.zc.c0;
    const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
    float* oh = a.gh.p + pix * a.gh.cs + a.gh.c0;
    float gz[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      gz[c] = gzc[c];
      if (a.mode == CPL_AFFINE && c >= a.ns && c < a.C) {
        const int j = c - a.ns;
        const float shift = hp[2 * j], scale = hp[2 * j + 1];
        const float e = expf(-logscale_of(scale));
        gz[c] = gzc[c] * e;
        oh[2 * j] = -gzc[c];
        const float gls = -gzc[c] * (zp[c] + shift);               // d(z2 e^-ls)/d ls = -(zc2 + shift)
        oh[2 * j + 1] = gls * (0.636f / (1.f + 4.f * scale * scale));
      }
    }
    if (a.mode != CPL_AFFINE)
      for (int c = 0; c < 3; ++c) oh[c] = -gzc[c];                  // z[:3] -= h
    store_pixel<CMAX>(a.gz, pix, a.C, gz);
  }
  block_channel_sums<CMAX>(v0, v1, a.C, shw, a.g_bias, a.g_logs, a.part, blockIdx.y * gridDim.x + blockIdx.x);
}

int launch_step_inv_bwd(const StepInvBwdArgs& a, hipStream_t st) {
  if (a.C < 1 || a.H < 1 || a.W < 1 || a.B < 1 || !a.g_bias || !a.g_logs) return HCF_ERR_ARG;
  const dim3 grid((unsigned)step_blocks_per_sample(a.H, a.W), (unsigned)a.B);
  if (a.C <= 8) hipLaunchKernelGGL((step_inv_bwd_kernel<8>), grid, dim3(256), 0, st, a);
  else if (a.C <= 12) hipLaunchKernelGGL((step_inv_bwd_kernel<12>), grid, dim3(256), 0, st, a);
  else if (a.C <= 24) hipLaunchKernelGGL((step_inv_bwd_kernel<24>), grid, dim3(256), 0, st, a);
  else if (a.C <= 48) hipLaunchKernelGGL((step_inv_bwd_kernel<48>), grid, dim3(256), 0, st, a);
  else return HCF_ERR_UNSUPPORTED;
  HCF_RET_T();
}

// ---- prior sample backward (SR: logs = h[1::2]) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void gauss_sample_bwd_kernel(const PriorBwdArgs a) {
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const size_t pix = (size_t)blockIdx.y * hw + i;
  const float* ap = a.a.p + pix * a.a.cs + a.a.c0;
  const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
  const float* ga = a.ga.p + pix * a.ga.cs + a.ga.c0;
  float* gh = a.gh.p + pix * a.gh.cs + a.gh.c0;
  for (int c = 0; c < a.C; ++c) {
    gh[2 * c] = ga[c];                               // a = mean + e^logs eps
    float gl = ga[c] * (ap[c] - hp[2 * c]);          // d a / d logs = e^logs eps = a - mean
    if (a.rescale) { const float s = hp[2 * c + 1]; gl *= 0.636f / (1.f + 4.f * s * s); }   // logs = 0.318 atan(2 s)
    gh[2 * c + 1] = gl;
  }
}

// z = (a - mean) e^-logs  ->  ga = gz e^-logs, g mean = -ga, g logs = -gz z
__global__ __launch_bounds__(256) void gauss_encode_bwd_kernel(const PriorBwdArgs a) {
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const size_t pix = (size_t)blockIdx.y * hw + i;
  const float* ap = a.a.p + pix * a.a.cs + a.a.c0;
  const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
  float* ga = a.ga.p + pix * a.ga.cs + a.ga.c0;
  float* gh = a.gh.p + pix * a.gh.cs + a.gh.c0;
  for (int c = 0; c < a.C; ++c) {
    const float s = hp[2 * c + 1];
    const float logs = a.rescale ? logscale_of(s) : s;
    const float e = expf(-logs);
    const float gz = a.gz_nchw ? a.gz_nchw[((size_t)blockIdx.y * a.C + c) * hw + i] : 0.f;
    const float z = (ap[c] - hp[2 * c]) * e;
    ga[c] = gz * e;
    gh[2 * c] = -gz * e;
    float gl = -gz * z;
    if (a.rescale) gl *= 0.636f / (1.f + 4.f * s * s);
    gh[2 * c + 1] = gl;
  }
}
int launch_gauss_encode_bwd(const PriorBwdArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)step_blocks_per_sample(a.H, a.W), (unsigned)a.B);
  hipLaunchKernelGGL(gauss_encode_bwd_kernel, grid, dim3(256), 0, st, a);
  HCF_RET_T();
}

__global__ __launch_bounds__(256) void add_nchw_grad_kernel(const float* g, View z, View gz, int hw, int clamp01) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const size_t pix = (size_t)blockIdx.y * hw + i;
  for (int c = 0; c < gz.n; ++c) {
    const float v = z.p[pix * z.cs + z.c0 + c];
    if (!clamp01 || (v >= 0.f && v <= 1.f)) gz.p[pix * gz.cs + gz.c0 + c] += g[((size_t)blockIdx.y * gz.n + c) * hw + i];
  }
}
int launch_add_nchw_grad(const float* g_nchw, View z, View gz, int B, int H, int W, int clamp01, hipStream_t st) {
  const dim3 grid((unsigned)step_blocks_per_sample(H, W), (unsigned)B);
  hipLaunchKernelGGL(add_nchw_grad_kernel, grid, dim3(256), 0, st, g_nchw, z, gz, H * W, clamp01);
  HCF_RET_T();
}

__global__ __launch_bounds__(256) void mask_flat_kernel(const float* g, const float* raw, float* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = raw[i];
  out[i] = (v >= 0.f && v <= 1.f) ? g[i] : 0.f;
}
int launch_mask_flat(const float* g, const float* raw, float* out, size_t n, hipStream_t st) {
  if (n == 0) return HCF_OK;
  hipLaunchKernelGGL(mask_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, raw, out, n);
  HCF_RET_T();
}
int launch_gauss_sample_bwd(const PriorBwdArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)step_blocks_per_sample(a.H, a.W), (unsigned)a.B);
  hipLaunchKernelGGL(gauss_sample_bwd_kernel, grid, dim3(256), 0, st, a);
  HCF_RET_T();
}

__global__ __launch_bounds__(256) void mask_unit_range_kernel(View z, View g, int hw) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n = g.n;
  if (e >= (long long)hw * n) return;
  const int c = (int)(e % n);
  const size_t pix = (size_t)blockIdx.y * hw + (size_t)(e / n);
  const float v = z.p[pix * z.cs + z.c0 + c];
  if (!(v >= 0.f && v <= 1.f)) g.p[pix * g.cs + g.c0 + c] = 0.f;      // torch.clamp(x, 0, 1) backward
}
int launch_mask_unit_range(View z, View g, int B, int H, int W, hipStream_t st) {
  if (z.n != g.n) return HCF_ERR_ARG;
  const long long per = (long long)H * W * g.n;
  hipLaunchKernelGGL(mask_unit_range_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, st, z, g, H * W);
  HCF_RET_T();
}

// ---- small helpers ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_view_kernel(View in, View out, int hw, float alpha) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n = out.n;
  if (e >= (long long)hw * n) return;
  const int c = (int)(e % n);
  const size_t pix = (size_t)blockIdx.y * hw + (size_t)(e / n);
  out.p[pix * out.cs + out.c0 + c] += alpha * in.p[pix * in.cs + in.c0 + c];
}
int launch_add_view(View in, View out, int B, int H, int W, float alpha, hipStream_t st) {
  if (in.n != out.n) return HCF_ERR_ARG;
  const long long per = (long long)H * W * out.n;
  hipLaunchKernelGGL(add_view_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, st, in, out, H * W, alpha);
  HCF_RET_T();
}

__global__ __launch_bounds__(256) void add_const_kernel(float* p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] += v;
}
int launch_add_const(float* p, size_t n, float v, hipStream_t st) {
  if (n == 0) return HCF_OK;
  hipLaunchKernelGGL(add_const_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
  HCF_RET_T();
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* x, float* y, size_t n, float alpha) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] += alpha * x[i];
}
int launch_axpy(const float* x, float* y, size_t n, float alpha, hipStream_t st) {
  if (n == 0) return HCF_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n, alpha);
  HCF_RET_T();
}

__global__ __launch_bounds__(256) void axpy_jobs_kernel(const AxpyJob* jobs) {
  const AxpyJob j = jobs[blockIdx.x];
  for (int i = threadIdx.x; i < j.n; i += 256) j.y[i] += j.alpha * (j.x ? j.x[i] : 1.f);
}
int launch_axpy_jobs(const AxpyJob* jobs_dev, int njobs, hipStream_t st) {
  if (njobs <= 0) return HCF_OK;
  hipLaunchKernelGGL(axpy_jobs_kernel, dim3((unsigned)njobs), dim3(256), 0, st, jobs_dev);
  HCF_RET_T();
}

// ---- LU-decomposed invertible 1x1 conv: dL/dW -> dl, du, dlog_s (see hcf_common.h) --------------------------------------
__global__ __launch_bounds__(256) void lu_chain_kernel(const LuChainArgs a) {
  constexpr int MAXC = 48;
  __shared__ float sG[MAXC * MAXC], sA[MAXC * MAXC], sP[MAXC * MAXC], sL[MAXC * MAXC], sU[MAXC * MAXC];
  const int C = a.C, n = C * C, tid = threadIdx.x;
  for (int e = tid; e < n; e += 256) { sG[e] = a.dW[e]; sP[e] = a.P[e]; sL[e] = a.L[e]; sU[e] = a.U[e]; }
  __syncthreads();
  for (int e = tid; e < n; e += 256) {             // A = P^T G
    const int i = e / C, j = e % C;
    float acc = 0.f;
    for (int k = 0; k < C; ++k) acc = fmaf(sP[k * C + i], sG[k * C + j], acc);
    sA[e] = acc;
  }
  __syncthreads();
  for (int e = tid; e < n; e += 256) {
    const int i = e / C, j = e % C;
    if (j < i) {                                   // dL = A U'^T, strictly lower part -> l
      float acc = 0.f;
      for (int k = 0; k < C; ++k) acc = fmaf(sA[i * C + k], sU[j * C + k], acc);
      a.dl[e] += acc;
    } else {                                       // dU' = L^T A, strictly upper part -> u, diagonal -> log_s
      float acc = 0.f;
      for (int k = 0; k < C; ++k) acc = fmaf(sL[k * C + i], sA[k * C + j], acc);
      if (j > i) a.du[e] += acc;
      else a.dlog_s[i] += acc * sU[i * C + i];
    }
  }
}
int launch_lu_chain(const LuChainArgs& a, hipStream_t st) {
  if (a.C < 1 || a.C > 48) return HCF_ERR_ARG;
  hipLaunchKernelGGL(lu_chain_kernel, dim3(1), dim3(256), 0, st, a);
  HCF_RET_T();
}

}  // namespace hcf
