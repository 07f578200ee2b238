// Stride-1 "same" convolution (3x3 / 1x1) with its gradients on DEVICE-RESIDENT tensors: the building block of the auxiliary
// nets of the HCFlow+ / ++ recipes (Discriminator_VGG_160 and VGGFeatureExtractor,
// codes/models/modules/discriminator_vgg_arch.py:68-157) in hcflow_amd/gan.py. Same kernels as the flow's own convs
// (hcf_conv.hip fp32 MFMA / hcf_conv_f16x3.hip + hcf_conv_wino.hip f16x3, hcf_conv_wgrad.hip), driven straight from the caller's
// PyTorch-layout weight in device memory: the packs are rebuilt by the device-side repack kernels (hcf_repack.hip) into a
// caller-provided workspace on every call, output channels are walked in blocks of <= 64.
//   x, y, g, dx: dense NHWC fp32 [B][H][W][cs], cs % 4 == 0, 16-byte aligned;  w: [cout][cin][k][k];  bias: [cout] or NULL
// The 4x4 stride-2 convs of the discriminator are expressed by the caller as squeeze2d + a 3x3 conv with a re-indexed
// weight (gan.py), so they run here too.
#include <algorithm>
#include <cstring>

#include "../../include/hcflow.h"
#include "hcf_common.h"

namespace hcf {
static inline int aux_ru4(int c) { return (c + 3) & ~3; }
static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// workspace layout (bytes): [0,256) range flag + zero page | bias (64 f) | scale (64 f) | fp32 pack | f16x3 pack | Winograd pack |
// weight-gradient partial tiles
struct AuxPlan {
  int cin, cout, k, taps, nchunk_f, nchunk_t;
  size_t o_bias, o_scale, o_pk, o_pk16, o_wino, o_part, total;
};
static AuxPlan aux_plan(int cin, int cout, int k, int B, int H, int W) {
  AuxPlan p;
  memset(&p, 0, sizeof(p));
  p.cin = cin; p.cout = cout; p.k = k; p.taps = k * k;
  p.nchunk_f = (aux_ru4(cin) + 15) / 16;                  // forward: K = cin
  p.nchunk_t = (aux_ru4(cout) + 15) / 16;                 // data gradient: K = cout
  const int nch = std::max(p.nchunk_f, p.nchunk_t);
  const size_t pk = ((size_t)nch * p.taps * 2 + 1) * 64 * 8 * sizeof(float);
  const size_t pk16 = ((size_t)nch + 1) * p.taps * 2 * 2 * 64 * 8 * sizeof(_Float16);
  const size_t wino = (k == 3) ? ((size_t)(cin / 16) + 1) * 65536 : 0;
  p.o_bias = 256; p.o_scale = p.o_bias + 256; p.o_pk = p.o_scale + 256;
  p.o_pk16 = p.o_pk + al256(pk); p.o_wino = p.o_pk16 + al256(pk16); p.o_part = p.o_wino + al256(wino);
  // weight gradient scratch: one resident round of partial tiles (conv_wgrad_scratch_floats), bounded generously
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  w.nsrc = 1; w.src[0] = mkview(nullptr, aux_ru4(cin), 0, cin); w.g = mkview(nullptr, aux_ru4(cout), 0, cout);
  w.B = B; w.H = H; w.W = W; w.taps = p.taps;
  size_t part = conv_wgrad_scratch_floats(w) * sizeof(float);
  float one = 1.f;
  w.g_max = &one;                                          // the f16x3 form sizes its grid differently
  part = std::max(part, conv_wgrad_scratch_floats(w) * sizeof(float));
  p.total = p.o_part + al256(part);
  return p;
}
}  // namespace hcf

using namespace hcf;

extern "C" {

size_t hcf_aux_conv2d_workspace(int32_t cin, int32_t cout, int32_t k, int32_t B, int32_t H, int32_t W) {
  if (cin < 1 || cout < 1 || (k != 1 && k != 3) || B < 1 || H < 1 || W < 1) return 0;
  return aux_plan(cin, cout, k, B, H, W).total;
}

// y[:, :, :, 0:cout] = act(conv(x[..., 0:cin], w) + bias). precision: HCF_PRECISION_EXACT / HCF_PRECISION_F16X3 (3x3 only; an
// input beyond the f16 range raises the int at work[0], which the caller reads once per network pass: hcflow_amd/gan.py).
int hcf_aux_conv2d(const float* x, int32_t cs_in, int32_t cin, int32_t B, int32_t H, int32_t W, const float* w,
                   const float* bias, int32_t cout, int32_t k, int32_t act, float* y, int32_t cs_out, void* work,
                   size_t work_bytes, int32_t precision, hcf_stream_t stream) {
  if (!x || !w || !y || !work || cin < 1 || cout < 1 || (k != 1 && k != 3) || B < 1 || H < 1 || W < 1 || (cs_in & 3) ||
      (cs_out & 3) || cs_in < cin || cs_out < cout || act < 0 || act > 2)
    return HCF_ERR_ARG;
  const AuxPlan p = aux_plan(cin, cout, k, B, H, W);
  if (work_bytes < p.total) return HCF_ERR_NOMEM;
  hipStream_t st = (hipStream_t)stream;
  char* wk = (char*)work;
  const bool f16 = (precision == PREC_F16X3) && k == 3;
  int rc = HCF_OK;
  for (int oc0 = 0; oc0 < cout && rc == HCF_OK; oc0 += 64) {
    const int nb = std::min(64, cout - oc0);
    const int npad = ((nb + 31) / 32) * 32;
    float* bvec = (float*)(wk + p.o_bias);
    float* svec = (float*)(wk + p.o_scale);
    rc = launch_repack_epilogue(0, bias ? bias + oc0 : nullptr, nullptr, nb, bvec, svec, st);
    if (rc != HCF_OK) break;
    RepackArgs r;
    memset(&r, 0, sizeof(r));
    r.w = w + (size_t)oc0 * cin * p.taps; r.cin_w = cin; r.taps = p.taps; r.cout = nb;
    r.srcs[0] = cin; r.nsrc = 1; r.nchunk = p.nchunk_f; r.npad = npad;
    r.pk = f16 ? nullptr : (float*)(wk + p.o_pk);
    r.pk16 = f16 ? (_Float16*)(wk + p.o_pk16) : nullptr;
    // (the packs have zero K-padding lanes beyond cin and zero n-padding beyond nb: clear, then fill the live entries)
    const size_t pkb = f16 ? ((size_t)p.nchunk_f + 1) * p.taps * 2 * 2 * npad * 8 * sizeof(_Float16)
                           : ((size_t)p.nchunk_f * p.taps * 2 + 1) * npad * 8 * sizeof(float);
    if (hipMemsetAsync(f16 ? (void*)r.pk16 : (void*)r.pk, 0, pkb, st) != hipSuccess) return HCF_ERR_HIP;
    rc = launch_repack_conv(r, st);
    if (rc != HCF_OK) break;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src[0] = mkview(const_cast<float*>(x), cs_in, 0, cin);
    a.src[1] = a.src[2] = a.src[0];
    a.nsrc = 1; a.B = B; a.H = H; a.W = W;
    a.nchunk = p.nchunk_f; a.bias = bvec; a.scale = svec; a.act = act;
    a.out = mkview(y, cs_out, oc0, nb);
    a.res1 = mkview(nullptr, 0, 0, 0);
    a.res2 = mkview(nullptr, 0, 0, 0);
    if (f16) {
      a.ovf = (int*)wk;
      a.zeros = reinterpret_cast<const float*>(wk) + 16;
      int r2 = HCF_ERR_UNSUPPORTED;
      if ((cin & 15) == 0 && cin >= 64 && (nb == 32 || nb == 64)) {      // the dense 3x3 layers take the Winograd form
        // the kernels over-read ONE chunk behind the pack, which must be zero: the region is shared by the 64- and the 32-wide
        // layouts of successive output-channel blocks (cout = 96, 160 ...), whose chunk sizes and hence "behind the pack" differ
        const size_t chunk_b = (nb == 64) ? 65536 : 32768;
        if (hipMemsetAsync(wk + p.o_wino + (size_t)(cin / 16) * chunk_b, 0, chunk_b, st) != hipSuccess) return HCF_ERR_HIP;
        if (launch_repack_wino(r.w, cin, nb, nb, wk + p.o_wino, st) == HCF_OK) r2 = launch_conv_wino(a, wk + p.o_wino, st);
      }
      if (r2 == HCF_ERR_UNSUPPORTED) {
        a.wpack = (const float*)(wk + p.o_pk16);
        r2 = launch_conv_f16x3(a, p.taps, st);
      }
      rc = r2;
    } else {
      a.wpack = (const float*)(wk + p.o_pk);
      rc = launch_conv(a, p.taps, st);
    }
  }
  return rc;
}

// Gradients of y = conv(x, w) (+ bias) given g = dL/dy (the caller applies the activation's derivative first):
//   dx (nullable): NHWC [B,H,W,cs_dx], channels [0, cin) overwritten;  dw: device [cout][cin][k][k], overwritten.
// (dL/dbias = per-channel sum of g: one reduction on the caller's side.)
int hcf_aux_conv2d_backward(const float* x, int32_t cs_in, int32_t cin, int32_t B, int32_t H, int32_t W, const float* w,
                            int32_t cout, int32_t k, const float* g, int32_t cs_g, float* dx, int32_t cs_dx, float* dw,
                            void* work, size_t work_bytes, int32_t precision, hcf_stream_t stream) {
  if (!x || !w || !g || !work || cin < 1 || cout < 1 || (k != 1 && k != 3) || B < 1 || H < 1 || W < 1 || (cs_in & 3) ||
      (cs_g & 3) || cs_in < cin || cs_g < cout || (dx && ((cs_dx & 3) || cs_dx < cin)))
    return HCF_ERR_ARG;
  const AuxPlan p = aux_plan(cin, cout, k, B, H, W);
  if (work_bytes < p.total) return HCF_ERR_NOMEM;
  hipStream_t st = (hipStream_t)stream;
  char* wk = (char*)work;
  int rc = HCF_OK;
  const View gv = mkview(const_cast<float*>(g), cs_g, 0, cout);
  // ---- data gradient: the same conv kernel on transposed, tap-flipped packs, <= 64 input channels per launch (exact fp32)
  for (int ic0 = 0; dx && ic0 < cin && rc == HCF_OK; ic0 += 64) {
    const int nb = std::min(64, cin - ic0);
    const int npad = ((nb + 31) / 32) * 32;
    float* bvec = (float*)(wk + p.o_bias);
    float* svec = (float*)(wk + p.o_scale);
    rc = launch_repack_epilogue(0, nullptr, nullptr, nb, bvec, svec, st);
    if (rc != HCF_OK) break;
    RepackArgs r;
    memset(&r, 0, sizeof(r));
    r.w = w; r.cin_w = cin; r.taps = p.taps; r.transposed = 1; r.off = ic0; r.cout = nb;
    r.srcs[0] = cout; r.nsrc = 1; r.nchunk = p.nchunk_t; r.npad = npad; r.pk = (float*)(wk + p.o_pk);
    if (hipMemsetAsync(r.pk, 0, ((size_t)p.nchunk_t * p.taps * 2 + 1) * npad * 8 * sizeof(float), st) != hipSuccess) return HCF_ERR_HIP;
    rc = launch_repack_conv(r, st);
    if (rc != HCF_OK) break;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src[0] = a.src[1] = a.src[2] = gv;
    a.nsrc = 1; a.B = B; a.H = H; a.W = W;
    a.wpack = r.pk; a.nchunk = p.nchunk_t; a.bias = bvec; a.scale = svec; a.act = ACT_NONE;
    a.out = mkview(dx, cs_dx, ic0, nb);
    a.res1 = mkview(nullptr, 0, 0, 0);
    a.res2 = mkview(nullptr, 0, 0, 0);
    rc = launch_conv(a, p.taps, st);
  }
  // ---- weight gradient (fp32 MFMA, fixed-order split-K reduce: bit-reproducible)
  if (rc == HCF_OK && dw) {
    if (hipMemsetAsync(dw, 0, (size_t)cout * cin * p.taps * sizeof(float), st) != hipSuccess) return HCF_ERR_HIP;
    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.src[0] = mkview(const_cast<float*>(x), cs_in, 0, cin);
    wa.nsrc = 1; wa.g = gv; wa.B = B; wa.H = H; wa.W = W; wa.taps = p.taps; wa.dw = dw;
    wa.part = (float*)(wk + p.o_part);
    wa.part_cap = (p.total - p.o_part) / sizeof(float);
    if (conv_wgrad_scratch_floats(wa) > wa.part_cap) return HCF_ERR_NOMEM;
    rc = launch_conv_wgrad(wa, st);
  }
  (void)precision;
  return rc;
}

}  // extern "C"
