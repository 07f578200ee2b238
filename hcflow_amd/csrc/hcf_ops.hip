// Per-op C-ABI entry points used by the unit parity tests (tests/test_gpu_ops.py). They wrap the
// same kernels the engine launches, on caller-provided NCHW device tensors; temporaries are
// hipMalloc'ed per call (test path, not the hot path).
#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/hcflow.h"
#include "hcf_common.h"

namespace hcf {
void pack_conv_weights(const float* w, int cin, int cout, int taps, const int* srcs, int nsrc, std::vector<float>& pk,
                       int& nchunk, int& npad);
bool pack_conv_weights_f16x3(const float* w, int cin, int cout, int taps, const int* srcs, int nsrc,
                             std::vector<float>& pk_as_float, int& nchunk, int& npad);
static int g_op_precision = PREC_EXACT;
static double g_last_clock_mhz = 0;
extern int g_f16x3_ablation;   // hcf_conv_f16x3.hip

static inline int ru4(int c) { return (c + 3) & ~3; }

struct Tmp {
  std::vector<void*> ptrs;
  bool ok = true;
  float* dev(size_t nfloat) {
    void* p = nullptr;
    if (hipMalloc(&p, nfloat * sizeof(float) + 256) != hipSuccess) { ok = false; return nullptr; }
    ptrs.push_back(p);
    return (float*)p;
  }
  float* up(const std::vector<float>& v) {
    float* d = dev(v.size());
    if (d && hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) ok = false;
    return d;
  }
  ~Tmp() {
    for (void* p : ptrs) hipFree(p);
  }
};

// pack + launch with the process-wide op precision; returns HCF_ERR_* (f16x3: also checks the range flag)
static int pack_and_launch(Tmp& t, ConvArgs& a, const float* w, int cin, int cout, int k, const int* srcs, int n_src,
                           hipStream_t st, int iters = 1) {
  std::vector<float> pk;
  int nchunk = 0, npad = 0;
  const bool f16 = (g_op_precision == PREC_F16X3) && cout <= 64 && k == 3;
  if (f16) {
    if (!pack_conv_weights_f16x3(w, cin, cout, k * k, srcs, n_src, pk, nchunk, npad)) return HCF_ERR_UNSUPPORTED;
    a.ovf = (int*)t.dev(64);
    if (a.ovf && hipMemsetAsync(a.ovf, 0, 256, st) != hipSuccess) return HCF_ERR_HIP;
    a.zeros = reinterpret_cast<const float*>(a.ovf) + 16;
  } else {
    pack_conv_weights(w, cin, cout, k * k, srcs, n_src, pk, nchunk, npad);
  }
  a.wpack = t.up(pk);
  a.nchunk = nchunk;
  if (!t.ok) return HCF_ERR_NOMEM;
  const float* wino = nullptr;             // eligible layers take the Winograd form, as in the engine (--ablate 256: off)
  if (f16 && !(g_f16x3_ablation & 256)) {
    std::vector<float> pkw;
    if (pack_conv_weights_wino(w, cin, cout, srcs, n_src, pkw)) wino = t.up(pkw);
    if (!t.ok) return HCF_ERR_NOMEM;
  }
  int rc = HCF_OK;
  for (int i = 0; i < iters && rc == HCF_OK; ++i) {
    rc = HCF_ERR_UNSUPPORTED;
    if (wino) rc = launch_conv_wino(a, wino, st);
    if (rc == HCF_ERR_UNSUPPORTED) rc = f16 ? launch_conv_f16x3(a, k * k, st) : launch_conv(a, k * k, st);
  }
  return rc;
}


static View nhwc_from_nchw(Tmp& t, const float* x, int B, int C, int H, int W, hipStream_t st, int& rc) {
  float* p = t.dev((size_t)B * H * W * ru4(C));
  View v = mkview(p, ru4(C), 0, C);
  if (!p) { rc = HCF_ERR_NOMEM; return v; }
  if (hipMemsetAsync(p, 0, (size_t)B * H * W * ru4(C) * sizeof(float), st) != hipSuccess) rc = HCF_ERR_HIP;
  const int r = launch_nchw_to_nhwc(x, v, B, C, H, W, st);
  if (r != HCF_OK) rc = r;
  return v;
}

static bool invert64(const float* Wm, int n, std::vector<float>& out_padded, int M) {
  std::vector<double> a((size_t)n * n), inv((size_t)n * n, 0.0);
  for (int i = 0; i < n * n; ++i) a[i] = Wm[i];
  for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
  for (int col = 0; col < n; ++col) {
    int piv = col;
    double best = fabs(a[(size_t)col * n + col]);
    for (int r = col + 1; r < n; ++r)
      if (fabs(a[(size_t)r * n + col]) > best) { best = fabs(a[(size_t)r * n + col]); piv = r; }
    if (best == 0.0) return false;
    if (piv != col)
      for (int c = 0; c < n; ++c) {
        std::swap(a[(size_t)piv * n + c], a[(size_t)col * n + c]);
        std::swap(inv[(size_t)piv * n + c], inv[(size_t)col * n + c]);
      }
    const double d = a[(size_t)col * n + col];
    for (int c = 0; c < n; ++c) { a[(size_t)col * n + c] /= d; inv[(size_t)col * n + c] /= d; }
    for (int r = 0; r < n; ++r) {
      if (r == col) continue;
      const double f = a[(size_t)r * n + col];
      if (f == 0.0) continue;
      for (int c = 0; c < n; ++c) {
        a[(size_t)r * n + c] -= f * a[(size_t)col * n + c];
        inv[(size_t)r * n + c] -= f * inv[(size_t)col * n + c];
      }
    }
  }
  out_padded.assign((size_t)M * M, 0.f);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) out_padded[(size_t)r * M + c] = (float)inv[(size_t)r * n + c];
  return true;
}

}  // namespace hcf

using namespace hcf;

extern "C" {

int hcf_op_conv2d(const float* const* src, const int32_t* src_c, const int32_t* src_up, int32_t n_src, int32_t B,
                  int32_t H, int32_t W, const float* w, const float* bias, const float* scale, int32_t cout, int32_t k,
                  int32_t act, const float* res1, float rs1, const float* res2, float rs2, float* out,
                  hcf_stream_t stream) {
  if (!src || !src_c || !w || !out || n_src < 1 || n_src > kMaxSrc || (k != 1 && k != 3) || cout < 1 || cout > 96)
    return HCF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  int cin = 0;
  int srcs[kMaxSrc];
  for (int i = 0; i < n_src; ++i) {
    const int up = src_up ? src_up[i] : 0;
    a.src[i] = nhwc_from_nchw(t, src[i], B, src_c[i], H >> up, W >> up, st, rc);
    a.src[i].up = up;
    srcs[i] = src_c[i];
    cin += src_c[i];
  }
  for (int i = n_src; i < kMaxSrc; ++i) a.src[i] = a.src[0];
  a.nsrc = n_src;
  a.B = B; a.H = H; a.W = W;
  const int npad = ((cout + 31) / 32) * 32;
  std::vector<float> hb(npad, 0.f), hs(npad, 1.f);
  for (int n = 0; n < cout; ++n) {
    if (bias) hb[n] = bias[n];
    if (scale) hs[n] = scale[n];
  }
  a.bias = t.up(hb);
  a.scale = t.up(hs);
  a.act = act;
  float* o = t.dev((size_t)B * H * W * ru4(cout));
  a.out = mkview(o, ru4(cout), 0, cout);
  a.res1 = mkview(nullptr, 0, 0, 0);
  a.res2 = mkview(nullptr, 0, 0, 0);
  if (res1) { a.res1 = nhwc_from_nchw(t, res1, B, cout, H, W, st, rc); a.rs1 = rs1; }
  if (res2) { a.res2 = nhwc_from_nchw(t, res2, B, cout, H, W, st, rc); a.rs2 = rs2; }
  if (!t.ok) return HCF_ERR_NOMEM;
  if (rc != HCF_OK) return rc;
  rc = pack_and_launch(t, a, w, cin, cout, k, srcs, n_src, st);
  if (rc != HCF_OK) return rc;
  rc = launch_nhwc_to_nchw(a.out, out, B, cout, H, W, 0, st);
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  if (a.ovf) {
    int h = 0;
    if (hipMemcpy(&h, a.ovf, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return HCF_ERR_HIP;
    if (h) return HCF_ERR_UNSUPPORTED;      // input beyond the f16 range: caller must use the exact kernel
  }
  return rc;
}

// Transposed weights for the data gradient of source window [off, off + n) of a conv with weight w [cout][cin][k][k]:
// wt[ic][oc][t] = w[oc][off + ic][taps - 1 - t]  (a 3x3 "same" conv of the output gradient with the flipped kernel)
static void transpose_weights(const float* w, int cin, int cout, int taps, int off, int n, std::vector<float>& wt) {
  wt.assign((size_t)n * cout * taps, 0.f);
  for (int ic = 0; ic < n; ++ic)
    for (int oc = 0; oc < cout; ++oc)
      for (int t = 0; t < taps; ++t)
        wt[((size_t)ic * cout + oc) * taps + t] = w[((size_t)oc * cin + off + ic) * taps + (taps - 1 - t)];
}

// Gradients of y = conv2d(cat(up(src_i)), w) + bias w.r.t. the sources, the weight and the bias, given
// g = dL/dy (torch.autograd of F.conv2d; the training path of SURVEY.md 8f rank 1). dsrc[i] may be NULL.
int hcf_op_conv2d_backward(const float* const* src, const int32_t* src_c, const int32_t* src_up, int32_t n_src,
                           int32_t B, int32_t H, int32_t W, const float* w, int32_t cout, int32_t k, const float* g,
                           float* const* dsrc, float* dw, float* dbias, hcf_stream_t stream) {
  if (!src || !src_c || !w || !g || n_src < 1 || n_src > kMaxSrc || (k != 1 && k != 3) || cout < 1 || cout > 96)
    return HCF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  const int taps = k * k;
  int cin = 0;
  for (int i = 0; i < n_src; ++i) cin += src_c[i];
  View gv = nhwc_from_nchw(t, g, B, cout, H, W, st, rc);
  if (!t.ok) return HCF_ERR_NOMEM;
  // ---- data gradients: one conv of g per <= 64-channel block of each source window
  int off = 0;
  for (int i = 0; i < n_src && rc == HCF_OK; ++i) {
    const int up = src_up ? src_up[i] : 0, n = src_c[i];
    if (dsrc && dsrc[i]) {
      float* full = t.dev((size_t)B * H * W * ru4(n));
      if (!t.ok) return HCF_ERR_NOMEM;
      for (int c0 = 0; c0 < n && rc == HCF_OK; c0 += 64) {
        const int nb = std::min(64, n - c0);
        std::vector<float> wt;
        transpose_weights(w, cin, cout, taps, off + c0, nb, wt);
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.src[0] = gv; a.src[1] = gv; a.src[2] = gv;
        a.nsrc = 1;
        a.B = B; a.H = H; a.W = W;
        std::vector<float> hb(64, 0.f), hs(64, 1.f);
        a.bias = t.up(hb);
        a.scale = t.up(hs);
        a.act = ACT_NONE;
        a.out = mkview(full, ru4(n), c0, nb);
        a.res1 = mkview(nullptr, 0, 0, 0);
        a.res2 = mkview(nullptr, 0, 0, 0);
        int one = cout;
        rc = pack_and_launch(t, a, wt.data(), cout, nb, k, &one, 1, st);
      }
      if (rc != HCF_OK) break;
      View fv = mkview(full, ru4(n), 0, n);
      if (up > 0) {
        float* low = t.dev((size_t)B * (H >> up) * (W >> up) * ru4(n));
        if (!t.ok) return HCF_ERR_NOMEM;
        View lv = mkview(low, ru4(n), 0, n);
        rc = launch_sumpool(fv, lv, B, H >> up, W >> up, up, 0, st);
        if (rc == HCF_OK) rc = launch_nhwc_to_nchw(lv, dsrc[i], B, n, H >> up, W >> up, 0, st);
      } else {
        rc = launch_nhwc_to_nchw(fv, dsrc[i], B, n, H, W, 0, st);
      }
    }
    off += n;
  }
  // ---- weight gradient
  if (rc == HCF_OK && dw) {
    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    for (int i = 0; i < n_src; ++i) {
      const int up = src_up ? src_up[i] : 0;
      wa.src[i] = nhwc_from_nchw(t, src[i], B, src_c[i], H >> up, W >> up, st, rc);
      wa.src[i].up = up;
    }
    wa.nsrc = n_src;
    wa.g = gv;
    wa.B = B; wa.H = H; wa.W = W; wa.taps = taps;
    const size_t nw = (size_t)cout * cin * taps;
    wa.dw = t.dev(nw);
    if (!t.ok) return HCF_ERR_NOMEM;
    if (hipMemsetAsync(wa.dw, 0, nw * sizeof(float), st) != hipSuccess) return HCF_ERR_HIP;
    wa.part_cap = conv_wgrad_scratch_floats(wa);
    wa.part = t.dev(wa.part_cap);
    if (!t.ok) return HCF_ERR_NOMEM;
    if (rc == HCF_OK && g_op_precision == PREC_F16X3) {        // f16 matrix cores: g is scaled by a power of two from max |g|
      float* gm = t.dev(2);
      if (!t.ok) return HCF_ERR_NOMEM;
      if (hipMemsetAsync(gm, 0, 2 * sizeof(float), st) != hipSuccess) return HCF_ERR_HIP;
      rc = launch_absmax(gv, B, H, W, gm, st);
      // X is split without a scale: an |x| beyond the f16 range (nothing range-checked this caller's tensor) would turn the
      // hi part into inf -> the fp32 weight-gradient kernel takes such inputs
      for (int i = 0; i < n_src && rc == HCF_OK; ++i)
        rc = launch_absmax(wa.src[i], B, H >> wa.src[i].up, W >> wa.src[i].up, gm + 1, st);
      float xmax = 0.f;
      if (rc == HCF_OK && (hipMemcpyAsync(&xmax, gm + 1, sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
                           hipStreamSynchronize(st) != hipSuccess))
        rc = HCF_ERR_HIP;
      if (xmax < 65504.f) wa.g_max = gm;           // NaN / inf compare false -> fp32 kernel
    }
    if (rc == HCF_OK) rc = launch_conv_wgrad(wa, st);
    if (rc == HCF_OK && hipMemcpyAsync(dw, wa.dw, nw * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) rc = HCF_ERR_HIP;
  }
  // ---- bias gradient: per-channel sums of g
  std::vector<double> hs;
  if (rc == HCF_OK && dbias) {
    double* sd = (double*)t.dev(4 * (size_t)cout);
    if (!t.ok) return HCF_ERR_NOMEM;
    hs.resize(2 * (size_t)cout);
    rc = launch_channel_stats(gv, B, H, W, sd, st);
    if (rc == HCF_OK && hipMemcpyAsync(hs.data(), sd, sizeof(double) * hs.size(), hipMemcpyDeviceToHost, st) != hipSuccess)
      rc = HCF_ERR_HIP;
  }
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  if (rc == HCF_OK && dbias)
    for (int c = 0; c < cout; ++c) dbias[c] = (float)hs[c];
  return rc;
}

// shader clock (MHz) observed inside the last hcf_bench_conv kernels (s_memtime / s_memrealtime)
double hcf_debug_last_clock_mhz(void) {
  const double probe = hcf::wino_clock_probe_mhz();       // (hcf_debug_clock_probe: the 64-channel Winograd kernel's own clock since the enable)
  return probe > 0.0 ? probe : g_last_clock_mhz;
}
int hcf_debug_clock_probe(int32_t enable) { return hcf::wino_clock_probe(enable); }
// timing ablations of the f16x3 kernel (tools/conv_bench.py --ablate); 0 = off
int hcf_debug_set_ablation(int32_t bits) { hcf::g_f16x3_ablation = bits; return HCF_OK; }

int hcf_op_set_precision(int32_t mode) {
  if (mode != PREC_EXACT && mode != PREC_F16X3) return HCF_ERR_ARG;
  g_op_precision = mode;
  return HCF_OK;
}

int hcf_op_squeeze2d(const float* x, float* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t haar,
                     hcf_stream_t stream) {
  if (!x || !out || (H & 1) || (W & 1)) return HCF_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  View in = nhwc_from_nchw(t, x, B, C, H, W, st, rc);
  float* o = t.dev((size_t)B * (H / 2) * (W / 2) * ru4(4 * C));
  if (!t.ok) return HCF_ERR_NOMEM;
  View ov = mkview(o, ru4(4 * C), 0, 4 * C);
  if (rc == HCF_OK) rc = haar ? launch_haar_fwd(in, ov, B, C, H, W, st) : launch_squeeze(in, ov, B, C, H, W, st);
  if (rc == HCF_OK) rc = launch_nhwc_to_nchw(ov, out, B, 4 * C, H / 2, W / 2, 0, st);
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  return rc;
}

int hcf_op_unsqueeze2d(const float* x, float* out, int32_t B, int32_t C4, int32_t H, int32_t W, int32_t haar,
                       hcf_stream_t stream) {
  if (!x || !out || (C4 & 3)) return HCF_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  View in = nhwc_from_nchw(t, x, B, C4, H, W, st, rc);
  if (!t.ok) return HCF_ERR_NOMEM;
  // exercises the fused boundary form (unsqueeze -> NCHW) used for the final HR image
  if (rc == HCF_OK) rc = launch_unsqueeze_nchw(in, out, B, C4, H, W, haar, 0, st);
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  return rc;
}

int hcf_op_step_inverse(const float* z, const float* h, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                        int32_t hC, int32_t mode, int32_t ns, const float* mat, const float* an_bias,
                        const float* an_logs, hcf_stream_t stream) {
  if (!z || !h || !out || !an_bias || !an_logs) return HCF_ERR_ARG;
  const int M = step_cmax(C);
  if (M < 0) return HCF_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  StepArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.H = H; a.W = W; a.C = C; a.ns = ns; a.mode = mode;
  a.z = nhwc_from_nchw(t, z, B, C, H, W, st, rc);
  a.h = nhwc_from_nchw(t, h, B, hC, H, W, st, rc);
  a.out = a.z;
  std::vector<float> hb(M, 0.f), hm(M, 0.f);
  for (int c = 0; c < C; ++c) { hb[c] = an_bias[c]; hm[c] = expf(-an_logs[c]); }
  a.an_bias = t.up(hb);
  a.an_mul = t.up(hm);
  if (mat) {
    std::vector<float> wi;
    if (!invert64(mat, C, wi, M)) return HCF_ERR_ARG;
    a.mat = t.up(wi);
  }
  if (!t.ok) return HCF_ERR_NOMEM;
  if (rc == HCF_OK) rc = launch_step_tail_inv(a, st);
  if (rc == HCF_OK) rc = launch_nhwc_to_nchw(a.out, out, B, C, H, W, 0, st);
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  return rc;
}

int hcf_op_step_forward_head(const float* z, float* out, int32_t B, int32_t C, int32_t H, int32_t W, const float* mat,
                             const float* an_bias, const float* an_logs, hcf_stream_t stream) {
  if (!z || !out || !an_bias || !an_logs) return HCF_ERR_ARG;
  const int M = step_cmax(C);
  if (M < 0) return HCF_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  StepArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.H = H; a.W = W; a.C = C;
  a.z = nhwc_from_nchw(t, z, B, C, H, W, st, rc);
  a.out = a.z;
  std::vector<float> hb(M, 0.f), hm(M, 0.f);
  for (int c = 0; c < C; ++c) { hb[c] = an_bias[c]; hm[c] = expf(an_logs[c]); }
  a.an_bias = t.up(hb);
  a.an_mul = t.up(hm);
  if (mat) {
    std::vector<float> wf((size_t)M * M, 0.f);
    for (int r = 0; r < C; ++r)
      for (int c = 0; c < C; ++c) wf[(size_t)r * M + c] = mat[(size_t)r * C + c];
    a.mat = t.up(wf);
  }
  if (!t.ok) return HCF_ERR_NOMEM;
  if (rc == HCF_OK) rc = launch_step_head_fwd(a, st);
  if (rc == HCF_OK) rc = launch_nhwc_to_nchw(a.out, out, B, C, H, W, 0, st);
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  return rc;
}

int hcf_op_step_forward_couple(const float* z, const float* h, float* out, float* logdet, int32_t B, int32_t C,
                               int32_t H, int32_t W, int32_t hC, int32_t mode, int32_t ns, hcf_stream_t stream) {
  if (!z || !h || !out) return HCF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  StepArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.H = H; a.W = W; a.C = C; a.ns = ns; a.mode = mode;
  a.z = nhwc_from_nchw(t, z, B, C, H, W, st, rc);
  a.h = nhwc_from_nchw(t, h, B, hC, H, W, st, rc);
  a.out = a.z;
  const int nb = step_blocks_per_sample(H, W);
  float* part = nullptr;
  if (logdet) {
    part = t.dev((size_t)B * nb);
    a.partial = part;
    a.partial_stride = nb;
  }
  if (!t.ok) return HCF_ERR_NOMEM;
  if (rc == HCF_OK) rc = launch_step_couple_fwd(a, st);
  if (rc == HCF_OK && logdet) {
    // engine-style final reduction; nll output unused here
    rc = launch_reduce_partials(part, nb, (mode == CPL_AFFINE) ? nb : 0, B, 0.0, 1.0, logdet, nullptr, st);
  }
  if (rc == HCF_OK) rc = launch_nhwc_to_nchw(a.out, out, B, C, H, W, 0, st);
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  return rc;
}

int hcf_op_gauss_logp(const float* h, const float* x, float* out_logp, int32_t B, int32_t C, int32_t H, int32_t W,
                      hcf_stream_t stream) {
  if (!h || !x || !out_logp) return HCF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  GaussArgs g;
  memset(&g, 0, sizeof(g));
  g.B = B; g.H = H; g.W = W; g.C = C;
  g.h = nhwc_from_nchw(t, h, B, 2 * C, H, W, st, rc);
  g.out = nhwc_from_nchw(t, x, B, C, H, W, st, rc);
  const int nb = step_blocks_per_sample(H, W);
  g.partial = t.dev((size_t)B * nb);
  g.partial_stride = nb;
  if (!t.ok) return HCF_ERR_NOMEM;
  if (rc == HCF_OK) rc = launch_gauss_logp(g, st);
  if (rc == HCF_OK) rc = launch_reduce_partials(g.partial, nb, nb, B, 0.0, 1.0, out_logp, nullptr, st);
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  return rc;
}

int hcf_op_gauss_sample(const float* h, const float* eps, float tau, uint64_t seed, float* out, int32_t B, int32_t C,
                        int32_t H, int32_t W, int32_t rescale, hcf_stream_t stream) {
  if (!h || !out) return HCF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  int rc = HCF_OK;
  GaussArgs g;
  memset(&g, 0, sizeof(g));
  g.B = B; g.H = H; g.W = W; g.C = C;
  g.h = nhwc_from_nchw(t, h, B, 2 * C, H, W, st, rc);
  g.rescale = rescale;
  g.eps = eps; g.tau = tau; g.seed = seed; g.offset = 0;
  float* o = t.dev((size_t)B * H * W * ru4(C));
  g.out = mkview(o, ru4(C), 0, C);
  if (!t.ok) return HCF_ERR_NOMEM;
  if (rc == HCF_OK) rc = launch_gauss_sample(g, st);
  if (rc == HCF_OK) rc = launch_nhwc_to_nchw(g.out, out, B, C, H, W, 0, st);
  if (hipStreamSynchronize(st) != hipSuccess) return HCF_ERR_HIP;
  return rc;
}

}  // extern "C"

// ---- micro-benchmark of the conv kernel (tools/conv_bench.py) ---------------------------------
// Allocates NHWC slabs once, launches the conv `iters` times back to back and times them with HIP
// events on the given stream. Inputs are uniform random in [-1, 1) (never zeros: DVFS, cdna guide
// rule 25); weights random. ms = average per launch.
extern "C" int hcf_bench_conv(int32_t B, int32_t H, int32_t W, const int32_t* src_c, int32_t n_src, int32_t cout,
                              int32_t k, int32_t iters, double* ms_per_launch, double* flops_per_launch,
                              hcf_stream_t stream) {
  if (!src_c || n_src < 1 || n_src > kMaxSrc || iters < 1) return HCF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Tmp t;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  int cin = 0, srcs[kMaxSrc];
  uint32_t rng = 12345u;
  auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
  for (int i = 0; i < n_src; ++i) {
    const int cs = ru4(src_c[i]);
    const size_t n = (size_t)B * H * W * cs;
    std::vector<float> h(n);
    for (size_t j = 0; j < n; ++j) h[j] = rnd();
    a.src[i] = mkview(t.up(h), cs, 0, src_c[i]);
    srcs[i] = src_c[i];
    cin += src_c[i];
  }
  for (int i = n_src; i < kMaxSrc; ++i) a.src[i] = a.src[0];
  a.nsrc = n_src;
  a.B = B; a.H = H; a.W = W;
  std::vector<float> w((size_t)cout * cin * k * k);
  for (auto& v : w) v = rnd() * 0.05f;
  const int npad = ((cout + 31) / 32) * 32;
  std::vector<float> hb(npad, 0.1f), hs(npad, 1.f);
  a.bias = t.up(hb); a.scale = t.up(hs);
  a.act = ACT_LRELU;
  a.out = mkview(t.dev((size_t)B * H * W * ru4(cout)), ru4(cout), 0, cout);
  a.res1 = mkview(nullptr, 0, 0, 0);
  a.res2 = mkview(nullptr, 0, 0, 0);
  if (!t.ok) return HCF_ERR_NOMEM;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  a.dbg = (unsigned long long*)t.dev(32);
  if (a.dbg) hipMemsetAsync(a.dbg, 0, 128, st);
  int rc = pack_and_launch(t, a, w.data(), cin, cout, k, srcs, n_src, st);     // pack + warm-up
  if (a.dbg) hipMemsetAsync(a.dbg, 0, 128, st);
  const bool f16 = a.ovf != nullptr;
  hipEventRecord(e0, st);
  for (int i = 0; i < iters && rc == HCF_OK; ++i) rc = f16 ? launch_conv_f16x3(a, k * k, st) : launch_conv(a, k * k, st);
  hipEventRecord(e1, st);
  if (hipEventSynchronize(e1) != hipSuccess) rc = HCF_ERR_HIP;
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (a.dbg) {
    unsigned long long hd[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpy(hd, a.dbg, 80, hipMemcpyDeviceToHost);
    if (hd[5] && hd[6])
      fprintf(stderr, "prologue parts: setup %.2f us, load issue %.2f us, wait + split %.2f us, LDS write + barrier %.2f us\n",
              hd[6] / 100.0 / hd[5], hd[7] / 100.0 / hd[5], hd[8] / 100.0 / hd[5], hd[9] / 100.0 / hd[5]);
    if (hd[1]) g_last_clock_mhz = 100.0 * (double)hd[0] / (double)hd[1];
    if (hd[5])
      fprintf(stderr, "block phases (avg of %llu blocks): prologue %.2f us, chunk loop %.2f us, epilogue %.2f us\n", hd[5],
              hd[2] / 100.0 / hd[5], hd[3] / 100.0 / hd[5], hd[4] / 100.0 / hd[5]);
  }
  if (ms_per_launch) *ms_per_launch = ms / iters;
  if (flops_per_launch) *flops_per_launch = 2.0 * k * k * cin * (double)cout * B * H * W;
  return rc;
}
