// Validation metrics on the device: what test_HCFlow.py computes per image on the CPU after the
// path -- tensor2img (utils/util.py:790-816), calculate_psnr_ssim (:898-982) with the Y channel of data/util.py:209-230,
// and the same on MATLAB-style bicubic down-scaled copies (utils/imresize.py). float64 arithmetic like the reference.
// HBM-bound elementwise / small-stencil work; images are kept as double planes [B][3 (B, G, R)][H][W] in 0..255 units.
#include <math.h>
#include <vector>
#include "../../include/hcflow.h"
#include "hcf_common.h"

namespace hcf {
namespace metrics {

struct Gauss11 { double g[11]; };

// tensor2img: clamp to [0, 1], * 255, round half to even (numpy .round()), BGR plane order
__global__ __launch_bounds__(256) void img255_kernel(const float* x, double* img, long long hw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const int b = blockIdx.y;
  for (int c = 0; c < 3; ++c) {
    const float v = fminf(fmaxf(x[((size_t)b * 3 + c) * hw + i], 0.f), 1.f);
    img[((size_t)b * 3 + (2 - c)) * hw + i] = rint((double)(v * 255.0f));
  }
}

// bgr2ycbcr(only_y) in 255 units: (24.966 B + 128.553 G + 65.481 R) / 255 + 16
__global__ __launch_bounds__(256) void y_kernel(const double* img, double* y, long long hw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const double* p = img + (size_t)blockIdx.y * 3 * hw;
  y[(size_t)blockIdx.y * hw + i] = (24.966 * p[i] + 128.553 * p[hw + i] + 65.481 * p[2 * hw + i]) / 255.0 + 16.0;
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  return (threadIdx.x == 0) ? sh[0] + sh[1] + sh[2] + sh[3] : 0.0;
}

// sum over planes and the cropped window of (a - b)^2 ; out[b] +=
__global__ __launch_bounds__(256) void sqdiff_kernel(const double* a, const double* b, int C, int H, int W, int crop, double* out) {
  __shared__ double sh[4];
  const int Hc = H - 2 * crop, Wc = W - 2 * crop;
  const long long n = (long long)C * Hc * Wc;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    const int c = (int)(i / ((long long)Hc * Wc));
    const long long r = i - (long long)c * Hc * Wc;
    const int y = (int)(r / Wc) + crop, x = (int)(r % Wc) + crop;
    const size_t e = (((size_t)blockIdx.y * C + c) * H + y) * W + x;
    const double d = a[e] - b[e];
    v = d * d;
  }
  const double s = block_sum_d(v, sh);
  if (threadIdx.x == 0) atomicAdd(out + blockIdx.y, s);
}

// SSIM map (utils/util.py:914-934) summed over the valid positions of one plane (blockIdx.y = b * C + c)
__global__ __launch_bounds__(256) void ssim_kernel(const double* a, const double* b, int H, int W, int crop, Gauss11 gk, double* out) {
  __shared__ double sh[4];
  const int Hv = H - 2 * crop - 10, Wv = W - 2 * crop - 10;
  const long long n = (long long)Hv * Wv;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    const int y0 = (int)(i / Wv) + crop, x0 = (int)(i % Wv) + crop;
    const double* pa = a + (size_t)blockIdx.y * H * W;
    const double* pb = b + (size_t)blockIdx.y * H * W;
    double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
    for (int dy = 0; dy < 11; ++dy) {
      const double* ra = pa + (size_t)(y0 + dy) * W + x0;
      const double* rb = pb + (size_t)(y0 + dy) * W + x0;
      for (int dx = 0; dx < 11; ++dx) {
        const double w = gk.g[dy] * gk.g[dx], u = ra[dx], t = rb[dx];
        m1 += w * u; m2 += w * t; s11 += w * u * u; s22 += w * t * t; s12 += w * u * t;
      }
    }
    const double C1 = 6.5025, C2 = 58.5225;                       // (0.01 * 255)^2, (0.03 * 255)^2
    const double mu11 = m1 * m1, mu22 = m2 * m2, mu12 = m1 * m2;
    v = ((2 * mu12 + C1) * (2 * (s12 - mu12) + C2)) / ((mu11 + mu22 + C1) * ((s11 - mu11) + (s22 - mu22) + C2));
  }
  const double s = block_sum_d(v, sh);
  if (threadIdx.x == 0) atomicAdd(out + blockIdx.y, s);
}

// imresize along one dimension: out[o] = sum_k w[o][k] * in[idx[o][k]]   (planes = B * C)
__global__ __launch_bounds__(256) void resize_dim_kernel(const double* in, double* out, int Hin, int Win, int Hout, int Wout,
                                                        int dim, const double* w, const int* idx, int P) {
  const long long n = (long long)Hout * Wout;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int y = (int)(i / Wout), x = (int)(i % Wout);
  const double* p = in + (size_t)blockIdx.y * Hin * Win;
  const int o = dim == 0 ? y : x;
  double s = 0.0;
  for (int k = 0; k < P; ++k) {
    const int j = idx[o * P + k];
    s += w[o * P + k] * (dim == 0 ? p[(size_t)j * Win + x] : p[(size_t)y * Win + j]);
  }
  out[(size_t)blockIdx.y * n + i] = s;
}

static double cubic(double x) {                                   // utils/imresize.py:50-57
  const double ax = fabs(x), ax2 = ax * ax, ax3 = ax2 * ax;
  if (ax <= 1) return 1.5 * ax3 - 2.5 * ax2 + 1;
  if (ax <= 2) return -0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2;
  return 0.0;
}

// utils/imresize.py:60-82 (all columns kept: the dropped ones have zero weight)
static void contributions(int in_len, int out_len, double scale, std::vector<double>& w, std::vector<int>& idx, int& P) {
  const double kw = (scale < 1) ? 4.0 / scale : 4.0;
  P = (int)ceil(kw) + 2;
  w.assign((size_t)out_len * P, 0.0);
  idx.assign((size_t)out_len * P, 0);
  for (int o = 0; o < out_len; ++o) {
    const double u = (o + 1) / scale + 0.5 * (1 - 1 / scale);
    const double left = floor(u - kw / 2);
    double sum = 0;
    for (int k = 0; k < P; ++k) {
      const int ind = (int)(left + k - 1);
      const double t = u - ind - 1;
      const double v = (scale < 1) ? scale * cubic(scale * t) : cubic(t);
      w[(size_t)o * P + k] = v;
      sum += v;
      const int m = 2 * in_len;
      int r = ((ind % m) + m) % m;                                // mirror padding through aux = [0..n-1, n-1..0]
      idx[(size_t)o * P + k] = r < in_len ? r : 2 * in_len - 1 - r;
    }
    for (int k = 0; k < P; ++k) w[(size_t)o * P + k] /= sum;
  }
}

struct DevBuf {
  std::vector<void*> ptrs;
  bool ok = true;
  template <class T> T* get(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
    ptrs.push_back(p);
    return (T*)p;
  }
  template <class T> T* up(const std::vector<T>& v) {
    T* p = get<T>(v.size());
    if (p && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) ok = false;
    return p;
  }
  ~DevBuf() { for (void* p : ptrs) hipFree(p); }
};

static inline dim3 grid1(long long n, int gy) { return dim3((unsigned)((n + 255) / 256), (unsigned)gy); }

// psnr / ssim of 3 planes + the Y plane for images already in 255 units; res[4] per image
static int psnr_ssim_planes(const double* a, const double* b, int B, int H, int W, int crop, DevBuf& d, double* res, hipStream_t st) {
  if (H - 2 * crop < 1 || W - 2 * crop < 1) return HCF_ERR_SHAPE;
  const bool has_ssim = H - 2 * crop >= 11 && W - 2 * crop >= 11;    // else the reference's valid region is empty: nan
  const long long hw = (long long)H * W;
  double* ya = d.get<double>((size_t)B * hw);
  double* yb = d.get<double>((size_t)B * hw);
  double* acc = d.get<double>((size_t)B * 10);      // per image: sq3, sqy, ssim c0..c2, ssim y
  if (!d.ok) return HCF_ERR_NOMEM;
  if (hipMemsetAsync(acc, 0, sizeof(double) * B * 10, st) != hipSuccess) return HCF_ERR_HIP;
  hipLaunchKernelGGL(y_kernel, grid1(hw, B), dim3(256), 0, st, a, ya, hw);
  hipLaunchKernelGGL(y_kernel, grid1(hw, B), dim3(256), 0, st, b, yb, hw);
  const int Hc = H - 2 * crop, Wc = W - 2 * crop;
  double* sq3 = acc;            // [B]
  double* sqy = acc + B;        // [B]
  double* ss3 = acc + 2 * B;    // [B][3]
  double* ssy = acc + 5 * B;    // [B]
  hipLaunchKernelGGL(sqdiff_kernel, grid1(3LL * Hc * Wc, B), dim3(256), 0, st, a, b, 3, H, W, crop, sq3);
  hipLaunchKernelGGL(sqdiff_kernel, grid1(1LL * Hc * Wc, B), dim3(256), 0, st, ya, yb, 1, H, W, crop, sqy);
  Gauss11 gk;
  double sum = 0;
  for (int i = 0; i < 11; ++i) { gk.g[i] = exp(-((i - 5.0) * (i - 5.0)) / (2.0 * 1.5 * 1.5)); sum += gk.g[i]; }
  for (int i = 0; i < 11; ++i) gk.g[i] /= sum;
  const long long nv = has_ssim ? (long long)(Hc - 10) * (Wc - 10) : 0;
  if (has_ssim) {
    hipLaunchKernelGGL(ssim_kernel, grid1(nv, B * 3), dim3(256), 0, st, a, b, H, W, crop, gk, ss3);
    hipLaunchKernelGGL(ssim_kernel, grid1(nv, B), dim3(256), 0, st, ya, yb, H, W, crop, gk, ssy);
  }
  if (hipGetLastError() != hipSuccess) return HCF_ERR_HIP;
  std::vector<double> h((size_t)B * 10);
  if (hipMemcpyAsync(h.data(), acc, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return HCF_ERR_HIP;
  for (int b_ = 0; b_ < B; ++b_) {
    const double mse3 = h[b_] / (3.0 * Hc * Wc), msey = h[B + b_] / ((double)Hc * Wc);
    res[b_ * 4 + 0] = mse3 == 0 ? INFINITY : 20 * log10(255.0 / sqrt(mse3));
    res[b_ * 4 + 1] = has_ssim ? (h[2 * B + 3 * b_] + h[2 * B + 3 * b_ + 1] + h[2 * B + 3 * b_ + 2]) / (3.0 * nv) : NAN;
    res[b_ * 4 + 2] = msey == 0 ? INFINITY : 20 * log10(255.0 / sqrt(msey));
    res[b_ * 4 + 3] = has_ssim ? h[5 * B + b_] / (double)nv : NAN;
  }
  return HCF_OK;
}

static int resize_down(const double* in, int planes, int H, int W, int scale, DevBuf& d, double** out, int& Ho, int& Wo, hipStream_t st) {
  const double s = 1.0 / scale;
  Ho = (int)ceil(s * H); Wo = (int)ceil(s * W);
  std::vector<double> w0, w1;
  std::vector<int> i0, i1;
  int P0 = 0, P1 = 0;
  contributions(H, Ho, s, w0, i0, P0);
  contributions(W, Wo, s, w1, i1, P1);
  double* dw0 = d.up(w0); int* di0 = d.up(i0);
  double* dw1 = d.up(w1); int* di1 = d.up(i1);
  double* t = d.get<double>((size_t)planes * Ho * W);
  double* o = d.get<double>((size_t)planes * Ho * Wo);
  if (!d.ok) return HCF_ERR_NOMEM;
  // equal scales: rows first (np.argsort of [s, s] = [0, 1], imresize.py:157,169)
  hipLaunchKernelGGL(resize_dim_kernel, grid1((long long)Ho * W, planes), dim3(256), 0, st, in, t, H, W, Ho, W, 0, dw0, di0, P0);
  hipLaunchKernelGGL(resize_dim_kernel, grid1((long long)Ho * Wo, planes), dim3(256), 0, st, t, o, Ho, W, Ho, Wo, 1, dw1, di1, P1);
  *out = o;
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace metrics
}  // namespace hcf

using namespace hcf;
using namespace hcf::metrics;

extern "C" {

int hcf_metric_psnr_ssim(const float* gt, const float* sr, int32_t B, int32_t H, int32_t W, int32_t crop_border,
                         int32_t scale, double* out, hcf_stream_t stream) {
  if (!gt || !sr || !out || B < 1 || H < 1 || W < 1 || crop_border < 0 || scale < 0) return HCF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  DevBuf d;
  const long long hw = (long long)H * W;
  double* a = d.get<double>((size_t)B * 3 * hw);
  double* b = d.get<double>((size_t)B * 3 * hw);
  if (!d.ok) return HCF_ERR_NOMEM;
  hipLaunchKernelGGL(img255_kernel, grid1(hw, B), dim3(256), 0, st, gt, a, hw);
  hipLaunchKernelGGL(img255_kernel, grid1(hw, B), dim3(256), 0, st, sr, b, hw);
  std::vector<double> r((size_t)B * 4);
  int rc = psnr_ssim_planes(a, b, B, H, W, crop_border, d, r.data(), st);
  if (rc != HCF_OK) return rc;
  for (int i = 0; i < B; ++i)
    for (int k = 0; k < 8; ++k) out[i * 8 + k] = k < 4 ? r[i * 4 + k] : 0.0;
  if (scale > 1) {
    double *da = nullptr, *db = nullptr;
    int Ho = 0, Wo = 0;
    rc = resize_down(a, B * 3, H, W, scale, d, &da, Ho, Wo, st);
    if (rc == HCF_OK) rc = resize_down(b, B * 3, H, W, scale, d, &db, Ho, Wo, st);
    if (rc == HCF_OK) rc = psnr_ssim_planes(da, db, B, Ho, Wo, 0, d, r.data(), st);
    if (rc != HCF_OK) return rc;
    for (int i = 0; i < B; ++i)
      for (int k = 0; k < 4; ++k) out[i * 8 + 4 + k] = r[i * 4 + k];
  }
  return HCF_OK;
}

int hcf_metric_imresize_down(const float* x, int32_t B, int32_t H, int32_t W, int32_t scale, double* out, hcf_stream_t stream) {
  if (!x || !out || B < 1 || H < 1 || W < 1 || scale < 2) return HCF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  DevBuf d;
  const long long hw = (long long)H * W;
  double* a = d.get<double>((size_t)B * 3 * hw);
  if (!d.ok) return HCF_ERR_NOMEM;
  hipLaunchKernelGGL(img255_kernel, grid1(hw, B), dim3(256), 0, st, x, a, hw);
  double* o = nullptr;
  int Ho = 0, Wo = 0;
  int rc = resize_down(a, B * 3, H, W, scale, d, &o, Ho, Wo, st);
  if (rc != HCF_OK) return rc;
  if (hipMemcpyAsync(out, o, sizeof(double) * (size_t)B * 3 * Ho * Wo, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return HCF_ERR_HIP;
  return HCF_OK;
}

}  // extern "C"
