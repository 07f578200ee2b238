// Device-side weight repack (training: the optimiser updates the parameters in place on the GPU every step; SURVEY.md
// 8f rank 2 "repack cache ... device-side repack"). Rewrites the conv packs of hcf_conv.hip / hcf_conv_f16x3.hip and
// the epilogue tables straight from the PyTorch-layout parameter tensors in device memory.
#include "hcf_common.h"

namespace hcf {

// logical weight L[n][ci][t]:
//   forward packs     L = w[n][ci][t]                          (w: [cout][cin][taps])
//   transposed packs  L = w[ci][off + n][taps - 1 - t]         (data gradient of channel block [off, off + nb))
__device__ __forceinline__ float logical_weight(const RepackArgs& a, int n, int ci, int t) {
  if (!a.transposed) return a.w[((size_t)n * a.cin_w + ci) * a.taps + t];
  return a.w[((size_t)ci * a.cin_w + a.off + n) * a.taps + (a.taps - 1 - t)];
}

// one thread per (chunk, tap, n, e in 0..15) of one pack
__device__ __forceinline__ void repack_one(const RepackArgs& a, long long i) {
  const long long total = (long long)a.nchunk * a.taps * a.cout * 16;
  if (i >= total) return;
  const int e = (int)(i & 15);
  const int n = (int)((i >> 4) % a.cout);
  const int t = (int)((i >> 4) / a.cout % a.taps);
  const int ch = (int)((i >> 4) / a.cout / a.taps);
  // virtual channel -> real input channel (sources padded to multiples of 4)
  int v = ch * 16 + e, ci = -1, base = 0;
  for (int s = 0; s < a.nsrc; ++s) {
    const int w4 = (a.srcs[s] + 3) & ~3;
    if (v < w4) { if (v < a.srcs[s]) ci = base + v; break; }
    v -= w4;
    base += a.srcs[s];
  }
  const float x = (ci >= 0) ? logical_weight(a, n, ci, t) : 0.f;
  if (a.pk) a.pk[(((size_t)ch * a.taps + t) * 2 + (e >> 3)) * ((size_t)a.npad * 8) + (size_t)n * 8 + (e & 7)] = x;
  if (a.pk16) {
    const size_t khalf = (size_t)a.npad * 8, plane = 2 * khalf, tapsz = 2 * plane, chunksz = (size_t)a.taps * tapsz;
    const size_t o = (size_t)ch * chunksz + (size_t)t * tapsz + (size_t)(e >> 3) * khalf + (size_t)n * 8 + (e & 7);
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)((x - (float)hi) * 2048.f);
    a.pk16[o] = (_Float16)((float)hi * 2048.f);
    a.pk16[o + plane] = lo;
  }
}

__global__ __launch_bounds__(256) void repack_conv_kernel(const RepackArgs a) {
  repack_one(a, (long long)blockIdx.x * 256 + threadIdx.x);
}

// every pack of the net in one launch: the block looks its job up in the block-prefix table (binary search, <= 12 steps)
__global__ __launch_bounds__(256) void repack_conv_batch_kernel(const RepackArgs* jobs, const long long* prefix, int njobs) {
  const long long blk = blockIdx.x;
  int lo = 0, hi = njobs;                       // invariant: prefix[lo] <= blk < prefix[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= blk) lo = mid; else hi = mid;
  }
  const RepackArgs a = jobs[lo];
  repack_one(a, (blk - prefix[lo]) * 256 + threadIdx.x);
}

int launch_repack_conv(const RepackArgs& a, hipStream_t st) {
  if (!a.w || a.cout < 1 || a.nchunk < 1) return HCF_ERR_ARG;
  const long long total = (long long)a.nchunk * a.taps * a.cout * 16;
  hipLaunchKernelGGL(repack_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// epilogue tables: kind 0 plain conv (bias), 1 conv + ActNorm (bias, exp(logs)), 2 Conv2dZeros (bias, exp(3 logs))
__global__ void repack_epilogue_kernel(int kind, const float* b, const float* l, int cout, float* bias, float* scale) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= cout) return;
  bias[c] = b ? b[c] : 0.f;
  scale[c] = (kind == 1) ? expf(l[c]) : (kind == 2) ? expf(3.f * l[c]) : 1.f;
}

int launch_repack_epilogue(int kind, const float* b, const float* l, int cout, float* bias, float* scale, hipStream_t st) {
  if (kind != 0 && !l) return HCF_ERR_ARG;
  hipLaunchKernelGGL(repack_epilogue_kernel, dim3((unsigned)((cout + 63) / 64)), dim3(64), 0, st, kind, b, l, cout, bias, scale);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

int launch_repack_conv_batch(const RepackArgs* jobs, const long long* prefix, int njobs, long long nblocks, hipStream_t st) {
  if (!jobs || !prefix || njobs < 1 || nblocks < 1 || nblocks > 0x7fffffffLL) return HCF_ERR_ARG;
  hipLaunchKernelGGL(repack_conv_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, jobs, prefix, njobs);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

__global__ void repack_epilogue_batch_kernel(const RepackEpiJob* jobs) {
  const RepackEpiJob j = jobs[blockIdx.x];
  for (int c = threadIdx.x; c < j.cout; c += blockDim.x) {
    j.bias[c] = j.b ? j.b[c] : 0.f;
    j.scale[c] = (j.kind == 1) ? expf(j.l[c]) : (j.kind == 2) ? expf(3.f * j.l[c]) : 1.f;
  }
}

int launch_repack_epilogue_batch(const RepackEpiJob* jobs, int njobs, hipStream_t st) {
  if (!jobs || njobs < 1) return HCF_ERR_ARG;
  hipLaunchKernelGGL(repack_epilogue_batch_kernel, dim3((unsigned)njobs), dim3(64), 0, st, jobs);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

__global__ void copy_jobs_kernel(const CopyJob* jobs) {
  const CopyJob j = jobs[blockIdx.x];
  for (int i = threadIdx.x; i < j.n; i += blockDim.x) j.dst[i] = j.src[i];
}

int launch_copy_jobs(const CopyJob* jobs, int njobs, hipStream_t st) {
  if (!jobs || njobs < 1) return HCF_ERR_ARG;
  hipLaunchKernelGGL(copy_jobs_kernel, dim3((unsigned)njobs), dim3(128), 0, st, jobs);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace hcf
