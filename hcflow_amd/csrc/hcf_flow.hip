// Flow-step glue and boundary kernels (HBM-bound, < 1 % of the FLOPs; SURVEY.md 8a rows a4-a8,
// a13-a18). One thread per pixel: a pixel's channel vector (<= 48 floats) lives in registers,
// ActNorm + invertible 1x1 conv + affine coupling are fused into a single pass over z, the C x C
// matrices are wave-uniform (scalar loads, SGPR operands of v_fma). Per-sample log-det sums use
// wave shuffles -> LDS -> one partial per block (deterministic; reduced later in double).
#include "hcf_common.h"
#include "hcf_step_math.h"

namespace hcf {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the 256-thread block; result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
  return r;
}

// ---- inverse flow step tail: coupling^-1 -> W^-1 -> actnorm^-1  (FlowStep.py:53-64) -----------
template <int CMAX>
__global__ __launch_bounds__(256) void step_tail_inv_kernel(const StepArgs a) {
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const size_t pix = (size_t)blockIdx.y * hw + i;
  float z[CMAX];
  load_pixel<CMAX>(a.z, pix, a.C, z);
  const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
  float y[CMAX];
  step_tail_inverse_pixel<CMAX>(z, hp, a.C, a.ns, a.mode, a.mat, a.an_bias, a.an_mul, y);
  if (a.aux.p) store_pixel<CMAX>(a.aux, pix, a.C, z);          // training tape: z after coupling^-1 (input of W^-1)
  store_pixel<CMAX>(a.out, pix, a.C, y);
  if (a.zpad16) store_pad16<CMAX>(a.zpad16, pix, a.zpad_n, y);
}

// The same for the 25..48-channel steps (the x8 net's deepest level: 20 x 20 pixels per sample, a C x C mat-vec of up to 2 304
// FMAs per pixel): one thread per pixel leaves 4/5 of the chip idle for 40 us per step at B = 32. Here a block takes 64 pixels and
// its four waves a quarter of the channels each (wave-uniform: the matrix rows still come through scalar loads): coupling^-1 on
// the wave's own channels -> LDS [channel][pixel] -> every wave reads the whole vector and computes its 12 output rows.
__global__ __launch_bounds__(256) void step_tail_inv48_kernel(const StepArgs a) {
  __shared__ float zs[48 * 64];
  const int hw = a.H * a.W;
  const int lane = threadIdx.x & 63;
  const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.x * 64 + lane;
  const bool ok = i < hw;
  const size_t pix = (size_t)blockIdx.y * hw + (ok ? i : hw - 1);
  const int C = a.C, ns = a.ns, c0 = part * 12;
  {
    const float* zp = a.z.p + pix * a.z.cs + a.z.c0;
    const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int c = c0 + k;
      float v = 0.f;
      if (c < C) {
        v = zp[c];
        if (a.mode == CPL_AFFINE) {
          if (c >= ns) {
            const int j = c - ns;
            v = v * expf(-logscale_of(hp[2 * j + 1])) - hp[2 * j];       // z2 * exp(-logscale) - shift  (AffineCouplings.py:65-87)
          }
        } else if (c < 3) {
          v = v - hp[c];                                                 // AffineCoupling3shift, LRvsothers = False (:150-153)
        }
      }
      zs[c * 64 + lane] = v;
    }
  }
  __syncthreads();
  float z[48];
#pragma unroll
  for (int c = 0; c < 48; ++c) z[c] = zs[c * 64 + lane];
  float y[12];
  if (a.mat) {
    const step_cptr M = const_table(a.mat) + c0 * 48;
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 48; ++k) acc = fmaf(M[r * 48 + k], z[k], acc);
      y[r] = acc;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 12; ++r) y[r] = zs[(c0 + r) * 64 + lane];
  }
  const step_cptr mul = const_table(a.an_mul) + c0, bias = const_table(a.an_bias) + c0;
#pragma unroll
  for (int r = 0; r < 12; ++r) y[r] = y[r] * mul[r] - bias[r];                                        // x * exp(-logs) - bias
  if (!ok) return;
  float* op = a.out.p + pix * a.out.cs + a.out.c0 + c0;
  if (((a.out.cs | a.out.c0) & 3) == 0 && (C & 3) == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (c0 + 4 * q < C) {
        step_f32x4 t = {y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
        *reinterpret_cast<step_f32x4*>(op + 4 * q) = t;
      }
  } else {
#pragma unroll
    for (int r = 0; r < 12; ++r)
      if (c0 + r < C) op[r] = y[r];
  }
}

// ---- forward flow step head: actnorm -> W  (FlowStep.py:40-47) --------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void step_head_fwd_kernel(const StepArgs a) {
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const size_t pix = (size_t)blockIdx.y * hw + i;
  float z[CMAX];
  load_pixel<CMAX>(a.z, pix, a.C, z);
#pragma unroll
  for (int c = 0; c < CMAX; ++c) z[c] = (z[c] + a.an_bias[c]) * a.an_mul[c];   // (x + b) * exp(logs)
  if (a.aux.p) store_pixel<CMAX>(a.aux, pix, a.C, z);                          // training tape: input of W
  if (a.mat) {
    float y[CMAX];
    matvec<CMAX>(const_table(a.mat), z, y);
    store_pixel<CMAX>(a.out, pix, a.C, y);
  } else {
    store_pixel<CMAX>(a.out, pix, a.C, z);
  }
}

// ---- forward coupling: z2 = (z2 + shift) * exp(logscale), partial += sum(logscale) ------------
// A pixel's channel vector in registers, 16-byte loads / stores (the first version walked the channels with scalar global accesses
// in runtime loops: 86 us at 8 x 320 x 320 x 12, four times what the bytes take).
template <int CMAX>
__global__ __launch_bounds__(256) void step_couple_fwd_kernel(const StepArgs a) {
  __shared__ float sh[4];
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float lsum = 0.f;
  if (i < hw) {
    const size_t pix = (size_t)blockIdx.y * hw + i;
    float z[CMAX];
    load_pixel<CMAX>(a.z, pix, a.C, z);
    const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
    if (a.mode == CPL_AFFINE) {
      // h = (shift, scale) interleaved for channels [ns, C). All loads first (unrolled, no store in between: the first version's
      // load -> compute -> store per channel could not be reordered, z and out may alias), then the arithmetic.
      float sh_[CMAX], sc_[CMAX];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        const bool on = c >= a.ns && c < a.C;
        const int jj = on ? c - a.ns : 0;
        sh_[c] = on ? hp[2 * jj] : 0.f;
        sc_[c] = on ? hp[2 * jj + 1] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        if (c >= a.ns && c < a.C) {
          const float ls = logscale_of(sc_[c]);
          z[c] = (z[c] + sh_[c]) * expf(ls);
          lsum += ls;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) z[c] = z[c] + hp[c];
    }
    store_pixel<CMAX>(a.out, pix, a.C, z);
  }
  if (a.partial) {
    const float s = block_sum(lsum, sh);
    if (threadIdx.x == 0) a.partial[(size_t)blockIdx.y * a.partial_stride + blockIdx.x] = s;
  }
}

int step_blocks_per_sample(int H, int W) { return (H * W + 255) / 256; }

#define HCF_DISPATCH_CMAX(KERNEL, A, ST)                                                       \
  do {                                                                                         \
    const dim3 grid((unsigned)step_blocks_per_sample((A).H, (A).W), (unsigned)(A).B);          \
    if ((A).C <= 8) hipLaunchKernelGGL((KERNEL<8>), grid, dim3(256), 0, ST, A);                \
    else if ((A).C <= 12) hipLaunchKernelGGL((KERNEL<12>), grid, dim3(256), 0, ST, A);         \
    else if ((A).C <= 24) hipLaunchKernelGGL((KERNEL<24>), grid, dim3(256), 0, ST, A);         \
    else if ((A).C <= 48) hipLaunchKernelGGL((KERNEL<48>), grid, dim3(256), 0, ST, A);         \
    else return HCF_ERR_UNSUPPORTED;                                                           \
  } while (0)

int step_cmax(int C) { return C <= 8 ? 8 : C <= 12 ? 12 : C <= 24 ? 24 : C <= 48 ? 48 : -1; }

int launch_step_tail_inv(const StepArgs& a, hipStream_t st) {
  if (a.C < 1 || a.H < 1 || a.W < 1 || a.B < 1) return HCF_ERR_ARG;
  if (a.C > 24 && a.C <= 48 && !a.aux.p && !a.zpad16) {        // (inference; the taped passes keep the one-thread-per-pixel form)
    const dim3 grid((unsigned)((a.H * a.W + 63) / 64), (unsigned)a.B);
    hipLaunchKernelGGL(step_tail_inv48_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
  }
  HCF_DISPATCH_CMAX(step_tail_inv_kernel, a, st);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

int launch_step_head_fwd(const StepArgs& a, hipStream_t st) {
  if (a.C < 1 || a.H < 1 || a.W < 1 || a.B < 1) return HCF_ERR_ARG;
  HCF_DISPATCH_CMAX(step_head_fwd_kernel, a, st);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

int launch_step_couple_fwd(const StepArgs& a, hipStream_t st) {
  if (a.C < 1 || a.H < 1 || a.W < 1 || a.B < 1) return HCF_ERR_ARG;
  HCF_DISPATCH_CMAX(step_couple_fwd_kernel, a, st);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// ---- Philox4x32-10 + Box-Muller (device-side eps for perf runs; parity runs inject eps) -------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint64_t idx) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
  const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// ---- Gaussian prior ---------------------------------------------------------------------------
// sample: a = mean + exp(logs) * eps,  (mean, s) = h[0::2], h[1::2]   (Basic.py:96-101,
// ConditionalFlow.py:61-64 SR, :88-91 rescaling with logs = 0.318 atan(2 s))
// One thread per (pixel, channel), channel fastest: consecutive lanes read consecutive (mean, s) pairs and write consecutive
// outputs of the NHWC records (the one-thread-per-pixel form walked its pixel's channels in a loop: every load of a wave touched 64
// records, and the Philox / Box-Muller chains of a pixel's 6-21 channels ran back to back in one lane: 217 us for 8.6 M elements).
// Same arithmetic per element, same Philox counter (the NCHW element index): bit-identical draws.
__global__ __launch_bounds__(256) void gauss_sample_kernel(const GaussArgs a) {
  const int hw = a.H * a.W;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)hw * a.C) return;
  const int b = blockIdx.y;
  const int i = (int)(t / a.C), c = (int)(t - (long long)i * a.C);
  const size_t pix = (size_t)b * hw + i;
  const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
  float* op = a.out.p + pix * a.out.cs + a.out.c0;
  const float mean = hp[2 * c], s = hp[2 * c + 1];
  const float logs = a.rescale ? logscale_of(s) : s;
  const size_t e = ((size_t)b * a.C + c) * hw + i;           // NCHW element index
  float eps;
  if (a.eps) eps = a.eps[e];
  else eps = (a.tau == 0.f) ? 0.f : a.tau * philox_normal(a.seed, a.offset, e + (size_t)a.b0 * a.C * hw);
  op[c] = mean + expf(logs) * eps;
}

// logp: sum -0.5 (2 logs + (x - mean)^2 / exp(2 logs) + ln 2pi)   (Basic.py:78-94)
__global__ __launch_bounds__(256) void gauss_logp_kernel(const GaussArgs a) {
  __shared__ float sh[4];
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (i < hw) {
    const size_t pix = (size_t)blockIdx.y * hw + i;
    const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
    const float* xp = a.out.p + pix * a.out.cs + a.out.c0;
    for (int c = 0; c < a.C; ++c) {
      const float mean = hp[2 * c], logs = hp[2 * c + 1];
      const float d = xp[c] - mean;
      acc += -0.5f * (logs * 2.f + (d * d) / expf(logs * 2.f) + 1.8378770664093453f);
    }
  }
  const float s = block_sum(acc, sh);
  if (threadIdx.x == 0) a.partial[(size_t)blockIdx.y * a.partial_stride + blockIdx.x] = s;
}

// rescaling forward: z = (a - mean) * exp(-logscale) -> NCHW   (ConditionalFlow.py:76-80)
__global__ __launch_bounds__(256) void gauss_encode_kernel(const GaussArgs a) {
  const int hw = a.H * a.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const int b = blockIdx.y;
  const size_t pix = (size_t)b * hw + i;
  const float* hp = a.h.p + pix * a.h.cs + a.h.c0;
  const float* xp = a.out.p + pix * a.out.cs + a.out.c0;
  for (int c = 0; c < a.C; ++c) {
    const float mean = hp[2 * c], ls = logscale_of(hp[2 * c + 1]);
    a.aux[((size_t)b * a.C + c) * hw + i] = (xp[c] - mean) * expf(-ls);
  }
}

static inline dim3 pix_grid(int B, int H, int W) { return dim3((unsigned)((H * W + 255) / 256), (unsigned)B); }
static inline dim3 elem_grid(int B, long long per_sample) { return dim3((unsigned)((per_sample + 255) / 256), (unsigned)B); }

int launch_gauss_sample(const GaussArgs& a, hipStream_t st) {
  const long long n = (long long)a.H * a.W * a.C;
  if (n <= 0 || (n + 255) / 256 > 0x7fffffffLL) return HCF_ERR_ARG;
  hipLaunchKernelGGL(gauss_sample_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)a.B), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}
int launch_gauss_logp(const GaussArgs& a, hipStream_t st) {
  if (!a.partial) return HCF_ERR_ARG;
  hipLaunchKernelGGL(gauss_logp_kernel, pix_grid(a.B, a.H, a.W), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}
int launch_gauss_encode(const GaussArgs& a, hipStream_t st) {
  if (!a.aux) return HCF_ERR_ARG;
  hipLaunchKernelGGL(gauss_encode_kernel, pix_grid(a.B, a.H, a.W), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// ---- layout / squeeze / Haar ------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* src, View dst, int C, int hw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const int b = blockIdx.y;
  float* op = dst.p + ((size_t)b * hw + i) * dst.cs + dst.c0;
  for (int c = 0; c < C; ++c) op[c] = src[((size_t)b * C + c) * hw + i];
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(View src, float* dst, int C, int hw, int clamp01) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const int b = blockIdx.y;
  const float* ip = src.p + ((size_t)b * hw + i) * src.cs + src.c0;
  for (int c = 0; c < C; ++c) {
    float v = ip[c];
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    dst[((size_t)b * C + c) * hw + i] = v;
  }
}

// Haar signs s_k(i,j), k = LL,HL,LH,HH (Basic.py:455-464)
__device__ __forceinline__ float haar_sign(int k, int i, int j) {
  const bool neg = (k == 1 && j == 1) || (k == 2 && i == 1) || (k == 3 && (i != j));
  return neg ? -1.f : 1.f;
}

// in [B,H,W,C] -> out [B,H/2,W/2,4C]; one thread per OUTPUT pixel
__global__ __launch_bounds__(256) void squeeze_kernel(View in, View out, int C, int H, int W, int haar) {
  const int H2 = H >> 1, W2 = W >> 1, hw2 = H2 * W2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw2) return;
  const int b = blockIdx.y, h = i / W2, w = i - h * W2;
  float* op = out.p + ((size_t)b * hw2 + i) * out.cs + out.c0;
  const float* ip[2][2];
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
      ip[di][dj] = in.p + ((size_t)((size_t)b * H + 2 * h + di) * W + 2 * w + dj) * in.cs + in.c0;
  for (int c = 0; c < C; ++c) {
    const float v00 = ip[0][0][c], v01 = ip[0][1][c], v10 = ip[1][0][c], v11 = ip[1][1][c];
    if (!haar) {
      op[c * 4 + 0] = v00; op[c * 4 + 1] = v01; op[c * 4 + 2] = v10; op[c * 4 + 3] = v11;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        op[k * C + c] = (((v00 * haar_sign(k, 0, 0) + v01 * haar_sign(k, 0, 1)) + v10 * haar_sign(k, 1, 0)) +
                         v11 * haar_sign(k, 1, 1)) / 4.0f;
    }
  }
}

// in [B,H,W,C4] -> out [B,2H,2W,C4/4]; one thread per INPUT pixel
// one thread per OUTPUT element (pixel, channel), channel fastest: consecutive lanes write consecutive floats of the NHWC records
// (the one-thread-per-input-pixel form looped over 4 C scattered stores per lane); same index map / same Haar sum order
__global__ __launch_bounds__(256) void unsqueeze_kernel(View in, View out, int C4, int H, int W, int haar) {
  const int C = C4 >> 2, W2 = 2 * W;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)4 * H * W * C) return;
  const int b = blockIdx.y;
  const int o = (int)(t / C), c = (int)(t - (long long)o * C);        // output pixel (of this sample), channel
  const int oy = o / W2, ox = o - oy * W2;
  const int h = oy >> 1, di = oy & 1, w = ox >> 1, dj = ox & 1;
  const float* ip = in.p + ((size_t)b * H * W + (size_t)h * W + w) * in.cs + in.c0;
  float* op = out.p + ((size_t)b * 4 * H * W + o) * out.cs + out.c0;
  if (!haar) op[c] = ip[c * 4 + di * 2 + dj];
  else
    op[c] = ((ip[c] * haar_sign(0, di, dj) + ip[C + c] * haar_sign(1, di, dj)) + ip[2 * C + c] * haar_sign(2, di, dj)) +
            ip[3 * C + c] * haar_sign(3, di, dj);
}

// NCHW (+ dequantisation noise) -> squeeze/haar -> NHWC   (HCFlowNet_SR_arch.py:52 + first layer)
__global__ __launch_bounds__(256) void nchw_squeeze_kernel(const float* src, const float* noise, float quant,
                                                          View out, int C, int H, int W, int haar) {
  const int H2 = H >> 1, W2 = W >> 1, hw2 = H2 * W2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw2) return;
  const int b = blockIdx.y, h = i / W2, w = i - h * W2;
  float* op = out.p + ((size_t)b * hw2 + i) * out.cs + out.c0;
  for (int c = 0; c < C; ++c) {
    float v[2][2];
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        const size_t e = ((size_t)((size_t)b * C + c) * H + 2 * h + di) * W + 2 * w + dj;
        float t = src[e];
        if (noise) t = t + noise[e] / quant;         // hr + rand / quant (HCFlowNet_SR_arch.py:52)
        v[di][dj] = t;
      }
    if (!haar) {
      op[c * 4 + 0] = v[0][0]; op[c * 4 + 1] = v[0][1]; op[c * 4 + 2] = v[1][0]; op[c * 4 + 3] = v[1][1];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        op[k * C + c] = (((v[0][0] * haar_sign(k, 0, 0) + v[0][1] * haar_sign(k, 0, 1)) + v[1][0] * haar_sign(k, 1, 0)) +
                         v[1][1] * haar_sign(k, 1, 1)) / 4.0f;
    }
  }
}

// NHWC [B,H,W,C4] -> unsqueeze/haar^-1 -> (clamp) -> NCHW [B,C4/4,2H,2W]
__global__ __launch_bounds__(256) void unsqueeze_nchw_kernel(View in, float* dst, int C4, int H, int W, int haar,
                                                            int clamp01) {
  const int hw = H * W, C = C4 >> 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const int b = blockIdx.y, h = i / W, w = i - h * W;
  const float* ip = in.p + ((size_t)b * hw + i) * in.cs + in.c0;
  for (int c = 0; c < C; ++c) {
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        float v;
        if (!haar) v = ip[c * 4 + di * 2 + dj];
        else
          v = ((ip[c] * haar_sign(0, di, dj) + ip[C + c] * haar_sign(1, di, dj)) + ip[2 * C + c] * haar_sign(2, di, dj)) +
              ip[3 * C + c] * haar_sign(3, di, dj);
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        dst[((size_t)((size_t)b * C + c) * (2 * H) + 2 * h + di) * (2 * W) + 2 * w + dj] = v;
      }
  }
}

// one thread per (pixel, channel), channel fastest (coalesced within the NHWC records)
__global__ __launch_bounds__(256) void copy_view_kernel(View in, View out, int hw) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)hw * in.n) return;
  const int i = (int)(t / in.n), c = (int)(t - (long long)i * in.n);
  const size_t pix = (size_t)blockIdx.y * hw + i;
  out.p[pix * out.cs + out.c0 + c] = in.p[pix * in.cs + in.c0 + c];
}

// z1 of a coupling net as a tensor of its own, zero padded to whole 16-channel chunks (out.n = 16 / 32 / 48 at stride out.cs): the
// first source of the Winograd form of the net's first convs, whose sources are whole chunks (hcf_engine.hip run_coupling_net)
__global__ __launch_bounds__(256) void copy_pad_kernel(View in, View out, int hw) {
  // one thread per (pixel, channel quad): consecutive threads write consecutive 16-byte pieces
  const int nq = out.n >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)hw * nq) return;
  const int k = (int)(idx % nq);
  const size_t pix = (size_t)blockIdx.y * hw + (size_t)(idx / nq);
  const float* ip = in.p + pix * in.cs + in.c0;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (4 * k + e < in.n) ? ip[4 * k + e] : 0.f;
  reinterpret_cast<float4*>(out.p + pix * out.cs + out.c0)[k] = make_float4(v[0], v[1], v[2], v[3]);
}

// Quant (Basic.py:187-191) + logp(lr, logs=-6, zq) (HCFlowNet_SR_arch.py:58-63); z has 3 channels
__global__ __launch_bounds__(256) void quant_logp_kernel(View z, const float* lr, float* lr_hat, int C, int hw,
                                                        float* partial, int partial_stride) {
  __shared__ float sh[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  float acc = 0.f;
  if (i < hw) {
    const float* zp = z.p + ((size_t)b * hw + i) * z.cs + z.c0;
    for (int c = 0; c < C; ++c) {
      const float zc = fminf(fmaxf(zp[c], 0.f), 1.f);
      const float zq = rintf(zc * 255.f) / 255.f;
      const size_t e = ((size_t)b * C + c) * hw + i;
      if (lr_hat) lr_hat[e] = fminf(fmaxf(zq, 0.f), 1.f);
      if (lr) {
        const float logs = -6.f;
        const float d = zq - lr[e];
        acc += -0.5f * (logs * 2.f + (d * d) / expf(logs * 2.f) + 1.8378770664093453f);
      }
    }
  }
  if (partial) {
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)b * partial_stride + blockIdx.x] = s;
  }
}

// out[b] = cst + sum_j partial[b][j] in double; nll = mean_b(-out[b]) / (ln2 * pixels)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* partial, int stride, int n, int B, double cst,
                                                             double pixels, float* out_logdet, float* out_nll) {
  __shared__ double shd[4];
  double nll = 0.0;
  for (int b = 0; b < B; ++b) {
    double acc = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) acc += (double)partial[(size_t)b * stride + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double tot = cst + shd[0] + shd[1] + shd[2] + shd[3];
      if (out_logdet) out_logdet[b] = (float)tot;
      nll += -tot / (0.6931471805599453 * pixels);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && out_nll) out_nll[0] = (float)(nll / B);
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

#define HCF_RET() return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP

// ---- per-channel statistics (ActNorm data-dependent init, ActNorms.py:37-43) ------------------------------------
// out[c] += sum over (b, y, x) of v[c];  out[n + c] += sum of v[c]^2, in double (atomics): the host turns them into
// bias = -mean, logs = log(scale / (sqrt(var) + 1e-6)). Consecutive threads read consecutive floats of the NHWC
// window (thread -> (pixel group, channel)); one block covers `ppb` pixels.
__global__ __launch_bounds__(256) void channel_stats_kernel(View v, long long npix, int ppb, double* out) {
  __shared__ double sh[2][256];
  const int n = v.n;
  const int groups = 256 / n;                  // pixels handled concurrently by one block (n <= 256)
  const int pg = threadIdx.x / n, c = threadIdx.x - pg * n;
  double s = 0.0, q = 0.0;
  if (pg < groups) {
    const long long p0 = (long long)blockIdx.x * ppb;
    const long long p1 = p0 + ppb < npix ? p0 + ppb : npix;
    for (long long p = p0 + pg; p < p1; p += groups) {
      const float x = v.p[(size_t)p * v.cs + v.c0 + c];
      s += (double)x;
      q += (double)x * (double)x;
    }
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < n) {
    double ts = 0.0, tq = 0.0;
    for (int g = 0; g < groups; ++g) { ts += sh[0][g * n + threadIdx.x]; tq += sh[1][g * n + threadIdx.x]; }
    atomicAdd(out + threadIdx.x, ts);
    atomicAdd(out + n + threadIdx.x, tq);
  }
}

// ---- backward of the nearest upsample folded into conv addressing: out[y, x] (+)= sum of the 2^up x 2^up block -----
__global__ __launch_bounds__(256) void sumpool_kernel(View in, View out, int Ho, int Wo, int up, int accumulate) {
  const int n = out.n;
  const long long total = (long long)gridDim.y * Ho * Wo * n;      // gridDim.y = B
  (void)total;
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)Ho * Wo * n) return;
  const int c = (int)(e % n);
  const long long pix = e / n;
  const int x = (int)(pix % Wo), y = (int)(pix / Wo);
  const int f = 1 << up, Wi = Wo << up, Hi = Ho << up;
  float s = 0.f;
  for (int i = 0; i < f; ++i)
    for (int j = 0; j < f; ++j)
      s += in.p[((size_t)((size_t)b * Hi + (y << up) + i) * Wi + (x << up) + j) * in.cs + in.c0 + c];
  float* o = out.p + ((size_t)((size_t)b * Ho + y) * Wo + x) * out.cs + out.c0 + c;
  *o = accumulate ? *o + s : s;
}

int launch_nchw_to_nhwc(const float* src, View dst, int B, int C, int H, int W, hipStream_t st) {
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, pix_grid(B, H, W), dim3(256), 0, st, src, dst, C, H * W);
  HCF_RET();
}
int launch_nhwc_to_nchw(View src, float* dst, int B, int C, int H, int W, int clamp01, hipStream_t st) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, pix_grid(B, H, W), dim3(256), 0, st, src, dst, C, H * W, clamp01);
  HCF_RET();
}
int launch_squeeze(View in, View out, int B, int C, int H, int W, hipStream_t st) {
  if ((H | W) & 1) return HCF_ERR_SHAPE;
  hipLaunchKernelGGL(squeeze_kernel, pix_grid(B, H / 2, W / 2), dim3(256), 0, st, in, out, C, H, W, 0);
  HCF_RET();
}
int launch_unsqueeze(View in, View out, int B, int C4, int H, int W, hipStream_t st) {
  if (C4 & 3) return HCF_ERR_SHAPE;
  hipLaunchKernelGGL(unsqueeze_kernel, elem_grid(B, (long long)H * W * C4), dim3(256), 0, st, in, out, C4, H, W, 0);
  HCF_RET();
}
int launch_haar_fwd(View in, View out, int B, int C, int H, int W, hipStream_t st) {
  if ((H | W) & 1) return HCF_ERR_SHAPE;
  hipLaunchKernelGGL(squeeze_kernel, pix_grid(B, H / 2, W / 2), dim3(256), 0, st, in, out, C, H, W, 1);
  HCF_RET();
}
int launch_haar_inv(View in, View out, int B, int C4, int H, int W, hipStream_t st) {
  if (C4 & 3) return HCF_ERR_SHAPE;
  hipLaunchKernelGGL(unsqueeze_kernel, elem_grid(B, (long long)H * W * C4), dim3(256), 0, st, in, out, C4, H, W, 1);
  HCF_RET();
}
int launch_nchw_squeeze(const float* src, const float* noise, float quant, View out, int B, int C, int H, int W,
                        int haar, hipStream_t st) {
  if ((H | W) & 1) return HCF_ERR_SHAPE;
  hipLaunchKernelGGL(nchw_squeeze_kernel, pix_grid(B, H / 2, W / 2), dim3(256), 0, st, src, noise, quant, out, C, H,
                     W, haar);
  HCF_RET();
}
int launch_unsqueeze_nchw(View in, float* dst, int B, int C4, int H, int W, int haar, int clamp01, hipStream_t st) {
  if (C4 & 3) return HCF_ERR_SHAPE;
  hipLaunchKernelGGL(unsqueeze_nchw_kernel, pix_grid(B, H, W), dim3(256), 0, st, in, dst, C4, H, W, haar, clamp01);
  HCF_RET();
}
int launch_copy_view(View in, View out, int B, int H, int W, hipStream_t st) {
  if (in.n < 1) return HCF_ERR_ARG;
  hipLaunchKernelGGL(copy_view_kernel, elem_grid(B, (long long)H * W * in.n), dim3(256), 0, st, in, out, H * W);
  HCF_RET();
}
int launch_copy_pad(View in, View out, int B, int H, int W, hipStream_t st) {
  if (in.n < 1 || in.n > out.n || (out.n & 15) || ((out.cs | out.c0) & 3) || !out.p) return HCF_ERR_ARG;
  const long long per = (long long)H * W * (out.n >> 2);
  hipLaunchKernelGGL(copy_pad_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, st, in, out, H * W);
  HCF_RET();
}
int launch_copy_pad16(View in, float* out16, int B, int H, int W, hipStream_t st) {
  return launch_copy_pad(in, mkview(out16, 16, 0, 16, 0), B, H, W, st);
}
int launch_quant_logp(View z, const float* lr_nchw, float* lr_hat_nchw, int B, int H, int W, float* partial,
                      int partial_stride, hipStream_t st) {
  hipLaunchKernelGGL(quant_logp_kernel, pix_grid(B, H, W), dim3(256), 0, st, z, lr_nchw, lr_hat_nchw, z.n, H * W,
                     partial, partial_stride);
  HCF_RET();
}
int launch_reduce_partials(const float* partial, int stride, int n, int B, double cst, double pixels, float* out_logdet,
                           float* out_nll, hipStream_t st) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, partial, stride, n, B, cst, pixels, out_logdet,
                     out_nll);
  HCF_RET();
}
int launch_fill(float* p, size_t n, float v, hipStream_t st) {
  if (n == 0) return HCF_OK;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
  HCF_RET();
}

int launch_channel_stats(View v, int B, int H, int W, double* out, hipStream_t st) {
  if (v.n < 1 || v.n > 256) return HCF_ERR_ARG;
  if (hipMemsetAsync(out, 0, sizeof(double) * 2 * v.n, st) != hipSuccess) return HCF_ERR_HIP;
  const long long npix = (long long)B * H * W;
  const int ppb = 4096;
  hipLaunchKernelGGL(channel_stats_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, st, v, npix, ppb, out);
  HCF_RET();
}

int launch_sumpool(View in, View out, int B, int Ho, int Wo, int up, int accumulate, hipStream_t st) {
  if (in.n != out.n || up < 0 || up > 4) return HCF_ERR_ARG;
  const long long per = (long long)Ho * Wo * out.n;
  hipLaunchKernelGGL(sumpool_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, st, in, out, Ho, Wo, up,
                     accumulate);
  HCF_RET();
}

}  // namespace hcf
