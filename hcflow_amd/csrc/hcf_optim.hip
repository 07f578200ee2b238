// Adam update of a net's parameters in ONE launch (the optimiser step of the training caller: reference
// HCFlow_SR_model.py:118-120 builds torch.optim.Adam over netG's ~1500 parameter tensors, :202 / HCFlow_Rescaling_model.py
// steps it once per iteration -- on the host side that is ~1500-entry multi-tensor lists per step).
//
// Layout (hcflow_amd/optim.py): every parameter of a group lives in one flat fp32 buffer P (64-float aligned slots), the moment
// buffers M (exp_avg) and V (exp_avg_sq) mirror it; gradients stay wherever autograd left them (the engine's flat gradient buffer,
// or anything else): a CHUNK table lists {gradient pointer, slot offset, length <= 4096} -- one block per chunk, one launch per step.
// HBM-bound: 28 bytes per parameter (read p, g, m, v; write p, m, v).
//
// Arithmetic = torch.optim.Adam's (amsgrad = False, maximize = False, L2 weight decay), torch/optim/adam.py _single_tensor_adam:
//   g += wd * p;  m += (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g;  p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps)
// with step_size = lr / bc1, bc_i = 1 - b_i^t computed on the host in double.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "hcf_common.h"
#include "../../include/hcflow.h"

namespace {

constexpr int kThreads = 256;

struct AdamK { float step_size, w1, b2, w2, bc2s, eps, wd; };

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamK& k) {
  g = (k.wd != 0.f) ? g + k.wd * p : g;
  m = m + (g - m) * k.w1;
  v = v * k.b2 + k.w2 * g * g;
  p = p - k.step_size * (m / (sqrtf(v) / k.bc2s + k.eps));
}

__global__ __launch_bounds__(kThreads) void adam_chunks_kernel(float* __restrict__ P, float* __restrict__ M, float* __restrict__ V,
                                                                const hcf_adam_chunk* __restrict__ chunks, AdamK k) {
  const hcf_adam_chunk c = chunks[blockIdx.x];
  const float* __restrict__ g = c.grad;
  float* const p = P + c.offset;
  float* const m = M + c.offset;
  float* const v = V + c.offset;
  const int n = (int)c.n, n4 = n >> 2, tid = threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {            // block-uniform: 16-byte gradient lanes
    for (int i = tid; i < n4; i += kThreads) {
      float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
      const float4 gv = reinterpret_cast<const float4*>(g)[i];
      adam1(pv.x, gv.x, mv.x, vv.x, k); adam1(pv.y, gv.y, mv.y, vv.y, k);
      adam1(pv.z, gv.z, mv.z, vv.z, k); adam1(pv.w, gv.w, mv.w, vv.w, k);
      reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
    }
  } else {                                                     // a gradient slice that starts off a 16-byte boundary
    for (int i = tid; i < n4; i += kThreads) {
      float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
      const float g0 = g[4 * i], g1 = g[4 * i + 1], g2 = g[4 * i + 2], g3 = g[4 * i + 3];
      adam1(pv.x, g0, mv.x, vv.x, k); adam1(pv.y, g1, mv.y, vv.y, k);
      adam1(pv.z, g2, mv.z, vv.z, k); adam1(pv.w, g3, mv.w, vv.w, k);
      reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
    }
  }
  for (int i = 4 * n4 + tid; i < n; i += kThreads) {           // the last 1..3 elements of a tensor
    float pv = p[i], mv = m[i], vv = v[i];
    adam1(pv, g[i], mv, vv, k);
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}

}  // namespace

extern "C" int hcf_adam_step(float* param, float* exp_avg, float* exp_avg_sq, const hcf_adam_chunk* chunks_dev, int32_t n_chunks,
                             double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                             hcf_stream_t stream) {
  if (n_chunks == 0) return HCF_OK;
  if (!param || !exp_avg || !exp_avg_sq || !chunks_dev || n_chunks < 0 || step < 1) return HCF_ERR_ARG;
  if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && lr >= 0.0 && weight_decay >= 0.0)) return HCF_ERR_ARG;
  if (((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) != 0)
    return HCF_ERR_ARG;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamK k;
  k.step_size = (float)(lr / bc1);
  k.w1 = (float)(1.0 - beta1);
  k.b2 = (float)beta2;
  k.w2 = (float)(1.0 - beta2);
  k.bc2s = (float)sqrt(bc2);
  k.eps = (float)eps;
  k.wd = (float)weight_decay;
  hipLaunchKernelGGL(adam_chunks_kernel, dim3((unsigned)n_chunks), dim3(kThreads), 0, static_cast<hipStream_t>(stream), param, exp_avg,
                     exp_avg_sq, chunks_dev, k);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}
