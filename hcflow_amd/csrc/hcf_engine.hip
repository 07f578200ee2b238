// Engine: builds the HCFlow layer graph from hcf_config, packs parameters for the kernels,
// owns the activation arena and enqueues the forward / inverse pass on the caller's stream.
//
// Reference structure mirrored here (state_dict names must match for strict checkpoint loads):
//   FlowNet.__init__            FlowNet_SR_x4.py:11-72, FlowNet_SR_x8.py:11-79, FlowNet_Rescaling_x4.py:11-80
//   FlowStep                    FlowStep.py:8-64
//   ConditionalFlow             ConditionalFlow.py:15-110
//   FCN / DenseBlock / RRDB     Basic.py:329-447
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <vector>

#include "../../include/hcflow.h"
#include "hcf_common.h"

namespace hcf {
int step_cmax(int C);

// The process' side streams: at most TWO per device, low priority, non-blocking, created on first use and never destroyed. Every
// engine of the process takes its extra streams from here (training: slot 0 = weight gradients, slot 1 = the conditional features'
// data gradients; inference: the two half batches of a split call, through hcf_aux_stream), so a process holds the caller's stream
// + two -- HIP spreads streams over four hardware queues, and a process that trained beside streams of its own for the split
// inference calls had five (backward pass 37 -> 66 ms: profiles/r05_notes.md section 4).
hipStream_t aux_stream(int slot) {
  static hipStream_t pool[64][2] = {};
  int dev = 0;
  if (slot < 0 || slot > 1 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!pool[dev][slot]) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) return nullptr;
    pool[dev][slot] = st;
  }
  return pool[dev][slot];
}

static inline int ru4(int c) { return (c + 3) & ~3; }

// ------------------------------------------------------------------------------------------------
struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool set = false;
};

struct Spec {
  std::string key;
  std::vector<int64_t> shape;
};

// One fused conv layer, packed for conv_mfma_kernel
extern int g_f16x3_ablation;   // hcf_conv_f16x3.hip (hcf_debug_set_ablation; bit 256: no Winograd kernels, bit 512: no Winograd form of FCN conv1 + conv2, bit 1024: no Winograd form of the DenseBlock coupling convs)

struct Conv {
  int taps = 9, cout = 0, nsrc = 0, src_n[kMaxSrc] = {0, 0, 0}, nchunk = 0, npad = 0, act = ACT_NONE;
  float *wpack = nullptr, *bias = nullptr, *scale = nullptr;   // device
  float* wpack16 = nullptr;                                     // device, f16x3 split pack (or null: exact only)
  float* wpack_wino = nullptr;                                  // device, Winograd f16x3 pack (eligible dense-block convs only)
  int wino_ntile = 0;                                           // > 0: the pack's channel tiles when cout is not 32 / 64 (zero-padded rows)
  double flops_per_pixel = 0;                                   // 2 * taps * cin * cout (algorithmic)
  std::string an_key;                                           // Basic.Conv2d: prefix of its ActNorm ("....conv1.actnorm")
  // training path: state_dict keys of the parameters behind this layer and what the epilogue sums mean for them
  std::string wkey, bkey, lkey;                                 // weight; bias (sum of d pre-activation); logs (or "")
  float l_mult = 0.f;                                           // d logs = l_mult * sum(dz * y): ActNorm 1, Conv2dZeros 3
  struct TPack { float *wpack = nullptr, *wpack16 = nullptr; int nchunk = 0, npad = 0, src = 0, c0 = 0, n = 0; };
  std::vector<TPack> tpacks;                                    // data-gradient packs: (source window, <= 64-ch block)
  bool gathered = false;                                        // a dense-block conv whose data gradients run in gather form (Rdb::gt)
};

struct Step {
  int C = 0, ns = 0, mode = CPL_AFFINE, cond = 0, cmax = 0;
  bool lr_vs_others = true, has_mat = false, fcn = true;
  int f_in = 0, f_out = 0, hid = 0;
  Conv c[5];                        // FCN: c[0..2]; DenseBlock: c[0..4]
  float *mat_inv = nullptr, *mat_fwd = nullptr, *bias = nullptr, *mul_inv = nullptr, *mul_fwd = nullptr;
  double ld_const = 0;              // per pixel: sum(actnorm logs) + slogdet(W)
  double lad = 0;                   // slogdet(W) alone
  std::string an_key;               // prefix of the step's ActNorm ("....actnorm")
  std::string wkey;                 // "....permute.weight" (or "")
  // conditional FCN coupling nets, f16x3 inference: conv1 as a 64-channel Winograd launch over [z1 padded to 16 | features] with
  // conv2 (1x1) in its epilogue (profiles/r03_notes.md section 8). c1w = c[0] with that source list and pack; built by finalize.
  Conv c1w;
  float* w4f_frag = nullptr;
  // DenseBlock coupling nets (the rescaling nets' steps, Basic.py:329-356), f16x3 inference: conv i >= 1 in Winograd form over
  // [z1 padded to whole 16-channel chunks | growth]; cw[i] = c[i] with that source list, pack and (last conv) a zero-padded
  // output tile. dw_pad = the padded width of z1 (0: none of this).
  Conv cw[5];
  int dw_pad = 0;
  float *mat_fwdT = nullptr;        // training: W^T padded [cmax][cmax] (gza = W^T gzb)
  float *winvT = nullptr;           // training: W^-T, [C][C] unpadded (d slogdet / dW)
  float *mat_invT = nullptr;        // training: (W^-1)^T padded [cmax][cmax] (reverse path: gzc = W^-T gy)
  // LU-decomposed invertible conv (Permutations.py:41-57,78-92): the engine composes W = P L U' on the host (fp64, rounded to
  // fp32) and runs the same kernels; lad = sum(log_s). Training: dL/dW is accumulated in lu_dw and chained into l / u / log_s
  // by lu_chain_kernel (hcf_train.hip) from the device copies of P, L = l o mask + I, U' = u o mask^T + diag(sign_s e^log_s).
  bool lu = false;
  std::string lu_pre;               // "....permute"
  float *lu_P = nullptr, *lu_L = nullptr, *lu_U = nullptr, *lu_dw = nullptr;   // device [C][C] each
};

// fat = the "fat launch" form of conv3 / conv4 (profiles/r03_notes.md): c34 = conv3 + the [x, x1, x2] part of conv4 as ONE
// 64-output-channel Winograd launch (second tile stored raw), c4b = conv4's completion over x3 (adds the stored partial)
// Pair j = 0: (conv1, conv2), j = 1: (conv3, conv4): ca[j] = conv 2j+1 + the old-input part of conv 2j+2, cb[j] = the completion.
// gt[m] (training): GATHER-form data-gradient pack of the block's tensor x_m (m = 0: the block input, m >= 1: growth tensor m):
// dL/dx_m = conv3x3 over cat(dL/dpre of conv m+1 .. conv 4 (gc channels each), dL/dpre of conv 5 (nf)) with the transposed,
// tap-flipped slices of those convs' weights -- one launch with K = (4 - m) gc + nf instead of one short-K launch per (conv, window).
struct Rdb { Conv c[5]; Conv ca[2], cb[2]; bool fat[2] = {false, false}; Conv::TPack gt[5]; bool gather = false; };
struct Rrdb { Rdb r[3]; };

struct CondFlow {
  int level = 0, C = 0, ns = 0, Ca = 0, nlc = 0;
  Conv conv_first, trunk_conv1, head;
  std::vector<Rrdb> trunk0, trunk1;
  std::vector<Step> steps;
};

struct Level {
  int C = 0, ns = 0;
  std::vector<Step> steps;
  CondFlow cf;
};

struct Arena {
  char* base = nullptr;
  size_t cap = 0, top = 0, peak = 0;
  bool dry = false;
  float* alloc(size_t nfloat) {
    size_t bytes = (nfloat * sizeof(float) + 64 + 255) & ~(size_t)255;   // +64: 4-float over-read slack
    float* p = dry ? reinterpret_cast<float*>((uintptr_t)0x1000 + top) : reinterpret_cast<float*>(base + top);
    top += bytes;
    if (top > peak) peak = top;
    return p;
  }
};

// w: PyTorch [cout][cin][k][k]. Virtual K order = sources concatenated, each padded to a multiple
// of 4 channels; packed as [chunk][tap][kg(2)][npad][8] plus one zero K-step so the kernel's
// one-step-ahead weight prefetch never reads past the end.
void pack_conv_weights(const float* w, int cin, int cout, int taps, const int* srcs, int nsrc, std::vector<float>& pk,
                       int& nchunk, int& npad) {
  int kv = 0;
  for (int i = 0; i < nsrc; ++i) kv += ru4(srcs[i]);
  nchunk = (kv + 15) / 16;
  npad = ((cout + 31) / 32) * 32;
  std::vector<int> vmap((size_t)nchunk * 16, -1);     // virtual channel -> real input channel (or -1)
  int v = 0, real = 0;
  for (int i = 0; i < nsrc; ++i) {
    for (int c = 0; c < srcs[i]; ++c) vmap[v + c] = real + c;
    v += ru4(srcs[i]);
    real += srcs[i];
  }
  const size_t step = (size_t)npad * 8;
  pk.assign(((size_t)nchunk * taps * 2 + 1) * step, 0.f);
  for (int ch = 0; ch < nchunk; ++ch)
    for (int t = 0; t < taps; ++t)
      for (int kg = 0; kg < 2; ++kg)
        for (int n = 0; n < cout; ++n)
          for (int e = 0; e < 8; ++e) {
            const int ci = vmap[ch * 16 + kg * 8 + e];
            if (ci < 0) continue;
            pk[(((size_t)ch * taps + t) * 2 + kg) * step + (size_t)n * 8 + e] = w[((size_t)n * cin + ci) * taps + t];
          }
}

// f16x3 pack for hcf_conv_f16x3.hip: halves [chunk][tap][plane(2)][k-half(2)][npad][8] with
// plane 0 = f16(w) * 2^11, plane 1 = f16((w - f16(w)) * 2^11); a chunk is one contiguous block that the
// kernel copies linearly into LDS. Returns false when a weight is too large for the scaled hi plane.
bool pack_conv_weights_f16x3(const float* w, int cin, int cout, int taps, const int* srcs, int nsrc,
                             std::vector<float>& pk_as_float, int& nchunk, int& npad) {
  int kv = 0;
  for (int i = 0; i < nsrc; ++i) kv += ru4(srcs[i]);
  nchunk = (kv + 15) / 16;
  npad = ((cout + 31) / 32) * 32;
  std::vector<int> vmap((size_t)nchunk * 16, -1);
  int v = 0, real = 0;
  for (int i = 0; i < nsrc; ++i) {
    for (int c = 0; c < srcs[i]; ++c) vmap[v + c] = real + c;
    v += ru4(srcs[i]);
    real += srcs[i];
  }
  const size_t khalf = (size_t)npad * 8, plane = 2 * khalf, tapsz = 2 * plane, chunksz = (size_t)taps * tapsz;
  std::vector<_Float16> pk(((size_t)nchunk + 1) * chunksz, (_Float16)0.f);   // +1 zero chunk: over-read slack
  for (int ch = 0; ch < nchunk; ++ch)
    for (int t = 0; t < taps; ++t)
      for (int n = 0; n < cout; ++n)
        for (int e = 0; e < 16; ++e) {
          const int ci = vmap[ch * 16 + e];
          if (ci < 0) continue;
          const float x = w[((size_t)n * cin + ci) * taps + t];
          if (!(fabsf(x) * 2048.f < 60000.f)) return false;
          const _Float16 hi = (_Float16)x;
          const _Float16 lo = (_Float16)((x - (float)hi) * 2048.f);
          const size_t o = (size_t)ch * chunksz + (size_t)t * tapsz + (size_t)(e >> 3) * khalf + (size_t)n * 8 + (e & 7);
          pk[o] = (_Float16)((float)hi * 2048.f);
          pk[o + plane] = lo;
        }
  pk_as_float.assign((pk.size() + 1) / 2, 0.f);
  memcpy(pk_as_float.data(), pk.data(), pk.size() * sizeof(_Float16));
  return true;
}

}  // namespace hcf

using namespace hcf;

struct hcf_engine {
  hcf_config cfg;
  std::vector<Spec> specs;
  std::map<std::string, HostTensor> params;
  std::vector<Level> levels;
  std::vector<float*> dev_allocs;      // packed weights
  size_t weight_bytes = 0;
  bool finalized = false;
  int device = -1;
  Arena arena;
  std::string err;
  // conv profiling
  bool prof = false;
  // (e0 is only recorded when something else was enqueued since the previous conv's e1; back-to-back convs -- the whole RRDB
  //  trunk -- share one event: the records cost ~2.5 us of stream time each, 2.4 % of a step with two per conv)
  struct ProfRec { hipEvent_t e0, e1; int taps, nt, kind, chained; double flops, bytes; };
  std::vector<ProfRec> prof_events;
  size_t prof_used = 0;
  unsigned long long launch_seq = 0, prof_last_seq = ~0ull;    // enqueue counter; value right after the last recorded e1
  // build state
  bool spec_mode = true;
  int rc = HCF_OK;
  // numerics: PREC_EXACT = fp32 MFMA everywhere; PREC_F16X3 = fp32-equivalent split on f16 MFMA
  int precision = PREC_EXACT;
  bool use_f16 = false;        // precision of the pass being enqueued
  int* ovf_flag = nullptr;     // device: [0] = range flag, bytes 64..191 = zero page for the f16x3 kernel
  int64_t n_fallbacks = 0;
  bool taping = false;         // a training forward is being recorded: no fused epilogues
  // Winograd packs of the eligible convs (built by hcf_finalize, rebuilt on the device by hcf_refresh_from_device);
  // HCF_NO_WINO=1 keeps them from being built at all.
  bool wino_enabled = getenv("HCF_NO_WINO") == nullptr;
  bool wino_stale = false;
  // ActNorm data-dependent initialisation (ActNorms.py:29-43), armed for ONE forward pass by hcf_actnorm_init_request
  std::set<std::string> an_pending, an_fitted;
  bool an_active = false;
  double* stats_dev = nullptr;   // 2 * 256 doubles
  float* unit_dev = nullptr;     // [0,256) zeros, [256,512) ones: identity epilogue of the statistics pass

  int fail(int code, const std::string& msg) {
    err = msg;
    if (rc == HCF_OK) rc = code;
    return code;
  }

  // ---------------------------------------------------------------- config helpers
  int level_channels(int level) const {
    int c = cfg.in_nc;
    for (int l = 0; l <= level; ++l) {
      c *= 4;
      if (l < level) c = (l < cfg.L - 1) ? c / 2 : 3;
    }
    return c;
  }
  int split_channels(int level) const { return level < cfg.L - 1 ? level_channels(level) / 2 : 3; }
  int cond_ch() const { return cfg.rrdb_nf * (cfg.kind == HCF_KIND_SR ? 2 : 1); }
  bool sr() const { return cfg.kind == HCF_KIND_SR; }

  // ---------------------------------------------------------------- parameter access
  const float* P(const std::string& key, std::vector<int64_t> shape) {
    if (spec_mode) {
      specs.push_back({key, shape});
      return nullptr;
    }
    auto it = params.find(key);
    if (it == params.end() || !it->second.set) {
      fail(HCF_ERR_KEY, "missing parameter: " + key);
      return nullptr;
    }
    if (it->second.shape != shape) {
      fail(HCF_ERR_SHAPE, "shape mismatch for parameter: " + key);
      return nullptr;
    }
    return it->second.data.data();
  }

  float* upload(const std::vector<float>& v) {
    float* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(float)) != hipSuccess) {
      fail(HCF_ERR_NOMEM, "hipMalloc failed for packed weights");
      return nullptr;
    }
    if (hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      fail(HCF_ERR_HIP, "hipMemcpy H2D failed");
      hipFree(d);
      return nullptr;
    }
    dev_allocs.push_back(d);
    weight_bytes += v.size() * sizeof(float);
    return d;
  }

  // ---------------------------------------------------------------- conv packing
  void pack_conv(Conv& cv, const float* w, const float* bias, const float* scale, int cin, int cout, int k,
                 std::vector<int> srcs, int act) {
    cv.taps = k * k;
    cv.cout = cout;
    cv.act = act;
    cv.nsrc = (int)srcs.size();
    int kv = 0, csum = 0;
    for (int i = 0; i < cv.nsrc; ++i) {
      cv.src_n[i] = srcs[i];
      kv += ru4(srcs[i]);
      csum += srcs[i];
    }
    cv.nchunk = (kv + 15) / 16;
    cv.npad = ((cout + 31) / 32) * 32;
    cv.flops_per_pixel = 2.0 * cv.taps * cin * cout;
    if (spec_mode) return;
    if (csum != cin) {
      fail(HCF_ERR_SHAPE, "internal: conv source channels do not add up");
      return;
    }
    if (!w) return;
    std::vector<float> pk;
    pack_conv_weights(w, cin, cout, cv.taps, srcs.data(), cv.nsrc, pk, cv.nchunk, cv.npad);
    std::vector<float> b(cv.npad, 0.f), s(cv.npad, 1.f);
    for (int n = 0; n < cout; ++n) {
      if (bias) b[n] = bias[n];
      if (scale) s[n] = scale[n];
    }
    cv.wpack = upload(pk);
    cv.bias = upload(b);
    cv.scale = upload(s);
    cv.wpack16 = nullptr;
    if (cv.npad <= 64) {      // 1x1 packs are only used fused into a preceding 3x3 (FCN conv1 + conv2)
      std::vector<float> pk16;
      int nc = 0, np = 0;
      if (pack_conv_weights_f16x3(w, cin, cout, cv.taps, srcs.data(), cv.nsrc, pk16, nc, np)) cv.wpack16 = upload(pk16);
    }
    cv.wpack_wino = nullptr;
    cv.wino_ntile = 0;
    if (cv.wpack16 && cv.taps == 9 && wino_enabled) {
      std::vector<float> pkw;
      if (pack_conv_weights_wino(w, cin, cout, srcs.data(), cv.nsrc, pkw)) cv.wpack_wino = upload(pkw);
      else if (wino_pad_ok && cout >= 8 && cout < 64 && cout != 32 && (cout & 3) == 0 && !getenv("HCF_NO_WINO_PAD")) {       // (A/B knob)
        // other widths of the dense-block growth convs (the rescaling trunk's 16 channels): a zero-padded 32 / 64-channel tile
        // (the prior heads, 12 / 24 output channels at K = 128, measured even-to-slower in that form and keep the direct kernel)
        const int cout_t = cout < 32 ? 32 : 64;
        std::vector<float> wp((size_t)cout_t * cin * 9, 0.f);
        memcpy(wp.data(), w, (size_t)cout * cin * 9 * sizeof(float));
        if (pack_conv_weights_wino(wp.data(), cin, cout_t, srcs.data(), cv.nsrc, pkw)) {
          cv.wpack_wino = upload(pkw);
          cv.wino_ntile = cout_t / 32;
        }
      }
    }
  }

  // nn.Conv2d(cin, cout, 3, 1, 1, bias=True)
  void build_conv(Conv& cv, const std::string& p, int cin, int cout, std::vector<int> srcs, int act) {
    const float* w = P(p + ".weight", {cout, cin, 3, 3});
    const float* b = P(p + ".bias", {cout});
    pack_conv(cv, w, b, nullptr, cin, cout, 3, srcs, act);
    cv.wkey = p + ".weight"; cv.bkey = p + ".bias"; cv.lkey.clear(); cv.l_mult = 0.f;
  }
  // Basic.Conv2d with ActNorm (Basic.py:14-53) + ReLU
  void build_conv_an(Conv& cv, const std::string& p, int cin, int cout, int k, std::vector<int> srcs) {
    const float* w = P(p + ".weight", {cout, cin, k, k});
    const float* ab = P(p + ".actnorm.bias", {1, cout, 1, 1});
    const float* al = P(p + ".actnorm.logs", {1, cout, 1, 1});
    std::vector<float> sc(cout, 1.f);
    if (al)
      for (int i = 0; i < cout; ++i) sc[i] = expf(al[i]);
    pack_conv(cv, w, ab, al ? sc.data() : nullptr, cin, cout, k, srcs, ACT_RELU);
    cv.an_key = p + ".actnorm";
    cv.wkey = p + ".weight"; cv.bkey = p + ".actnorm.bias"; cv.lkey = p + ".actnorm.logs"; cv.l_mult = 1.f;
  }
  // Basic.Conv2dZeros (Basic.py:57-72): (conv + bias) * exp(logs * 3)
  void build_conv_zeros(Conv& cv, const std::string& p, int cin, int cout, std::vector<int> srcs) {
    const float* w = P(p + ".weight", {cout, cin, 3, 3});
    const float* b = P(p + ".bias", {cout});
    const float* lg = P(p + ".logs", {cout, 1, 1});
    std::vector<float> sc(cout, 1.f);
    if (lg)
      for (int i = 0; i < cout; ++i) sc[i] = expf(lg[i] * 3.f);
    pack_conv(cv, w, b, lg ? sc.data() : nullptr, cin, cout, 3, srcs, ACT_NONE);
    cv.wkey = p + ".weight"; cv.bkey = p + ".bias"; cv.lkey = p + ".logs"; cv.l_mult = 3.f;
  }

  static std::vector<int> srcs2(int a, int b) {
    std::vector<int> v;
    v.push_back(a);
    if (b > 0) v.push_back(b);
    return v;
  }

  // ---------------------------------------------------------------- linear algebra (host, fp64)
  static bool invert(const std::vector<double>& A, int n, std::vector<double>& inv, double& logabsdet) {
    std::vector<double> a(A);
    inv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
    logabsdet = 0.0;
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
      logabsdet += log(fabs(d));
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
    return true;
  }

  // ---------------------------------------------------------------- FlowStep
  void build_step(Step& s, const std::string& p, int C, int cond, int perm, int coupling, int nn_module, int hid,
                  bool lr_vs_others) {
    s.C = C;
    s.cond = cond;
    s.cmax = step_cmax(C);
    s.lr_vs_others = lr_vs_others;
    s.hid = hid;
    s.fcn = (nn_module == HCF_NN_FCN);
    s.an_key = p + ".actnorm";
    s.lu = (perm == HCF_PERM_INVCONV) && cfg.lu_decomposed != 0;
    s.lu_pre = p + ".permute";
    s.wkey = (perm == HCF_PERM_INVCONV && !s.lu) ? p + ".permute.weight" : std::string();
    if (s.cmax < 0) { fail(HCF_ERR_UNSUPPORTED, "flow step with more than 48 channels"); return; }
    const float* ab = P(p + ".actnorm.bias", {1, C, 1, 1});
    const float* al = P(p + ".actnorm.logs", {1, C, 1, 1});
    const float* W = nullptr;
    s.has_mat = (perm == HCF_PERM_INVCONV);
    std::vector<float> lu_w, lu_l, lu_u;
    double lu_sumlogs = 0;
    if (s.lu) {
      // state_dict order of the module: parameters l, log_s, u, then the buffers p, sign_s (Permutations.py:51-55)
      const float* pl = P(s.lu_pre + ".l", {C, C});
      const float* ps = P(s.lu_pre + ".log_s", {C});
      const float* pu = P(s.lu_pre + ".u", {C, C});
      const float* pp = P(s.lu_pre + ".p", {C, C});
      const float* pg = P(s.lu_pre + ".sign_s", {C});
      if (!spec_mode && rc == HCF_OK) {
        compose_lu(pl, ps, pu, pp, pg, C, lu_w, lu_l, lu_u, lu_sumlogs);
        W = lu_w.data();
      }
    } else if (s.has_mat) W = P(p + ".permute.weight", {C, C});
    // coupling geometry (AffineCouplings.py:18-19, 101-106)
    int z1_n;
    if (coupling == HCF_COUPLING_AFFINE) {
      s.mode = CPL_AFFINE; s.ns = C / 2; z1_n = C / 2; s.f_out = (C - C / 2) * 2;
    } else if (lr_vs_others) {
      s.mode = CPL_AFFINE; s.ns = 3; z1_n = 3; s.f_out = (C - 3) * 2;
    } else {
      s.mode = CPL_SHIFT3; s.ns = 3; z1_n = C - 3; s.f_out = 3;
    }
    s.f_in = z1_n + cond;
    const std::string f = p + ".affine.f";
    if (s.fcn) {
      build_conv_an(s.c[0], f + ".conv1", s.f_in, hid, 3, srcs2(z1_n, cond));
      build_conv_an(s.c[1], f + ".conv2", hid, hid, 1, srcs2(hid, 0));
      build_conv_zeros(s.c[2], f + ".conv3", hid, s.f_out, srcs2(hid, 0));
      s.c1w = Conv();
      s.w4f_frag = nullptr;
      static const bool no_w4f = getenv("HCF_NO_W4F") != nullptr;      // A/B knob, read once
      if (!spec_mode && rc == HCF_OK && wino_enabled && !no_w4f && cond >= 16 && (cond & 15) == 0 && z1_n <= 16 && hid == 64 &&
          s.c[0].wpack16 && s.c[1].wpack16) {
        const std::vector<float>& w1 = params[f + ".conv1.weight"].data;
        const std::vector<float>& w2 = params[f + ".conv2.weight"].data;
        const int cin_p = 16 + cond;
        std::vector<float> wp((size_t)hid * cin_p * 9, 0.f), pk, fr;
        for (int oc = 0; oc < hid; ++oc)
          for (int ic = 0; ic < s.f_in; ++ic)
            memcpy(&wp[((size_t)oc * cin_p + (ic < z1_n ? ic : ic - z1_n + 16)) * 9], &w1[((size_t)oc * s.f_in + ic) * 9], 9 * sizeof(float));
        const int sp[2] = {16, cond};
        if (w1.size() == (size_t)hid * s.f_in * 9 && w2.size() == (size_t)hid * hid &&
            pack_conv_weights_wino(wp.data(), cin_p, hid, sp, 2, pk, 16) && pack_conv_weights_1x1_frag(w2.data(), fr)) {
          s.c1w = s.c[0];
          s.c1w.src_n[0] = 16;
          s.c1w.wpack_wino = upload(pk);
          s.c1w.tpacks.clear();
          s.w4f_frag = upload(fr);
        }
      }
    } else {
      // DenseBlock(in, out, gc=hid) (Basic.py:329-356); dense concat order is (x, x1, x2, ...) and x itself
      // is cat(z1, u) when conditional -> sources: z1 [, u], growth
      for (int i = 0; i < 5; ++i) {
        std::vector<int> srcs;
        srcs.push_back(z1_n);
        if (cond > 0) srcs.push_back(cond);
        if (i > 0) srcs.push_back(i * hid);
        if ((int)srcs.size() > kMaxSrc) { fail(HCF_ERR_UNSUPPORTED, "too many conv sources"); return; }
        build_conv(s.c[i], f + ".conv" + std::to_string(i + 1), s.f_in + i * hid, i < 4 ? hid : s.f_out, srcs,
                   i < 4 ? ACT_LRELU : ACT_NONE);
      }
      s.dw_pad = 0;
      for (int i = 0; i < 5; ++i) s.cw[i] = Conv();
      static const bool no_dw = getenv("HCF_NO_DENSE_WINO") != nullptr;      // A/B knob, read once
      if (!spec_mode && rc == HCF_OK && wino_enabled && !no_dw && cond == 0 && hid >= 16 && (hid & 15) == 0 && z1_n <= 48 && s.f_out <= 64) {
        const int npad = (z1_n + 15) & ~15;
        bool any = false;
        for (int i = 1; i < 5; ++i) {
          const int cin = s.f_in + i * hid, cin_p = npad + i * hid, cout = i < 4 ? hid : s.f_out, cout_t = cout <= 32 ? 32 : 64;
          const std::vector<float>& w = params[f + ".conv" + std::to_string(i + 1) + ".weight"].data;
          if (cin_p < 48 || !s.c[i].wpack16 || w.size() != (size_t)cout * cin * 9) continue;
          std::vector<float> wp((size_t)cout_t * cin_p * 9, 0.f), pk;
          for (int oc = 0; oc < cout; ++oc)
            for (int ic = 0; ic < cin; ++ic)
              memcpy(&wp[((size_t)oc * cin_p + (ic < z1_n ? ic : ic - z1_n + npad)) * 9], &w[((size_t)oc * cin + ic) * 9], 9 * sizeof(float));
          const int sp[2] = {npad, i * hid};
          if (!pack_conv_weights_wino(wp.data(), cin_p, cout_t, sp, 2, pk, 48)) continue;
          s.cw[i] = s.c[i];
          s.cw[i].src_n[0] = npad;
          s.cw[i].wpack_wino = upload(pk);
          s.cw[i].wino_ntile = (cout == 32 || cout == 64) ? 0 : cout_t / 32;
          s.cw[i].tpacks.clear();
          any = true;
        }
        if (any) s.dw_pad = npad;
      }
    }
    if (spec_mode || rc != HCF_OK) return;
    const int M = s.cmax;
    std::vector<float> bias(M, 0.f), mi(M, 0.f), mf(M, 0.f);
    double sumlogs = 0;
    for (int c = 0; c < C; ++c) {
      bias[c] = ab[c];
      mi[c] = expf(-al[c]);
      mf[c] = expf(al[c]);
      sumlogs += (double)al[c];
    }
    s.bias = upload(bias);
    s.mul_inv = upload(mi);
    s.mul_fwd = upload(mf);
    s.ld_const = sumlogs;
    if (s.has_mat) {
      std::vector<double> A((size_t)C * C), inv;
      for (int i = 0; i < C * C; ++i) A[i] = (double)W[i];
      double lad = 0;
      if (!invert(A, C, inv, lad)) { fail(HCF_ERR_ARG, "singular invertible-conv weight: " + p); return; }
      std::vector<float> wi((size_t)M * M, 0.f), wf((size_t)M * M, 0.f);
      for (int r = 0; r < C; ++r)
        for (int c = 0; c < C; ++c) {
          wi[(size_t)r * M + c] = (float)inv[(size_t)r * C + c];    // inverse(W.double()).float(), Permutations.py:74
          wf[(size_t)r * M + c] = W[(size_t)r * C + c];
        }
      s.mat_inv = upload(wi);
      s.mat_fwd = upload(wf);
      {
        std::vector<float> wt((size_t)M * M, 0.f), it((size_t)C * C, 0.f), itp((size_t)M * M, 0.f);
        for (int r = 0; r < C; ++r)
          for (int c = 0; c < C; ++c) {
            wt[(size_t)c * M + r] = W[(size_t)r * C + c];
            it[(size_t)c * C + r] = (float)inv[(size_t)r * C + c];
            itp[(size_t)c * M + r] = (float)inv[(size_t)r * C + c];
          }
        s.mat_fwdT = upload(wt);
        s.winvT = upload(it);
        s.mat_invT = upload(itp);
      }
      if (s.lu) {
        lad = lu_sumlogs;                              // dlogdet = sum(log_s) * pixels (Permutations.py:84)
        std::vector<float> pm(s_lu_p(s), s_lu_p(s) + (size_t)C * C);
        s.lu_P = upload(pm);
        s.lu_L = upload(lu_l);
        s.lu_U = upload(lu_u);
        s.lu_dw = upload(std::vector<float>((size_t)C * C, 0.f));
      }
      s.lad = lad;
      s.ld_const += lad;
    }
  }
  const float* s_lu_p(const Step& s) { return params[s.lu_pre + ".p"].data.data(); }

  // W = P (L o mask + I) (U o mask^T + diag(sign_s exp(log_s))) (Permutations.py:78-86), composed in fp64 and rounded once;
  // Lc / Uc = the two triangular factors as the chain rule of the training path needs them; sumlogs = sum(log_s)
  static void compose_lu(const float* l, const float* log_s, const float* u, const float* p, const float* sign_s, int C,
                         std::vector<float>& W, std::vector<float>& Lc, std::vector<float>& Uc, double& sumlogs) {
    std::vector<double> L((size_t)C * C, 0.0), U((size_t)C * C, 0.0), LU((size_t)C * C, 0.0);
    sumlogs = 0;
    for (int i = 0; i < C; ++i) {
      for (int j = 0; j < C; ++j) {
        L[(size_t)i * C + j] = j < i ? (double)l[(size_t)i * C + j] : (i == j ? 1.0 : 0.0);
        U[(size_t)i * C + j] = j > i ? (double)u[(size_t)i * C + j] : 0.0;
      }
      U[(size_t)i * C + i] = (double)(sign_s[i] * expf(log_s[i]));       // fp32 exp, as torch.exp on the fp32 parameter
      sumlogs += (double)log_s[i];
    }
    for (int i = 0; i < C; ++i)
      for (int k = 0; k <= i; ++k) {
        const double a = L[(size_t)i * C + k];
        if (a == 0.0) continue;
        for (int j = k; j < C; ++j) LU[(size_t)i * C + j] += a * U[(size_t)k * C + j];
      }
    W.assign((size_t)C * C, 0.f);
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        double acc = 0;
        for (int k = 0; k < C; ++k) acc += (double)p[(size_t)i * C + k] * LU[(size_t)k * C + j];
        W[(size_t)i * C + j] = (float)acc;
      }
    Lc.resize((size_t)C * C);
    Uc.resize((size_t)C * C);
    for (size_t i = 0; i < (size_t)C * C; ++i) { Lc[i] = (float)L[i]; Uc[i] = (float)U[i]; }
  }

  bool wino_pad_ok = false;      // pack_conv may build a zero-padded Winograd tile for output widths other than 32 / 64
  void build_rdb(Rdb& r, const std::string& p, int nf, int gc) {
    wino_pad_ok = true;
    for (int i = 0; i < 4; ++i)
      build_conv(r.c[i], p + ".conv" + std::to_string(i + 1), nf + i * gc, gc, srcs2(nf, i * gc), ACT_LRELU);
    wino_pad_ok = false;
    build_conv(r.c[4], p + ".conv5", nf + 4 * gc, nf, srcs2(nf, 4 * gc), ACT_NONE);
    r.fat[0] = r.fat[1] = false;
    static const bool no_fat = getenv("HCF_NO_FAT") != nullptr;          // A/B knob, read once
    // gc = 32: pairs as one 64-channel launch (tile 1 = the partial) + a 32 -> 32 completion. gc = 16 (the rescaling nets' trunks,
    // round 4): pairs as one 32-channel launch whose upper half-tile is the partial + a 16 -> 16 completion in a zero-padded tile
    // -- instead of four convs that each fill half of a 32-wide MFMA tile with padding.
    if (spec_mode || rc != HCF_OK || !wino_enabled || no_fat || (gc != 32 && gc != 16) || (nf & 15) || nf < 32) return;
    for (int j = 0; j < 2 && rc == HCF_OK; ++j) {
      const std::string pa = p + ".conv" + std::to_string(2 * j + 1), pb = p + ".conv" + std::to_string(2 * j + 2);
      auto wa = params.find(pa + ".weight"), wb = params.find(pb + ".weight");
      auto ba = params.find(pa + ".bias"), bb = params.find(pb + ".bias");
      if (wa == params.end() || wb == params.end() || ba == params.end() || bb == params.end()) return;
      const int ka = nf + 2 * j * gc, kb = ka + gc;            // input channels of the two convs
      std::vector<float> wab((size_t)2 * gc * ka * 9), biasab(2 * gc, 0.f), wc((size_t)gc * gc * 9);
      for (int oc = 0; oc < gc; ++oc) {
        memcpy(&wab[(size_t)oc * ka * 9], &wa->second.data[(size_t)oc * ka * 9], sizeof(float) * ka * 9);
        memcpy(&wab[(size_t)(gc + oc) * ka * 9], &wb->second.data[(size_t)oc * kb * 9], sizeof(float) * ka * 9);
        memcpy(&wc[(size_t)oc * gc * 9], &wb->second.data[((size_t)oc * kb + ka) * 9], sizeof(float) * gc * 9);
        biasab[oc] = ba->second.data[oc];
      }
      pack_conv(r.ca[j], wab.data(), biasab.data(), nullptr, ka, 2 * gc, 3, srcs2(nf, 2 * j * gc), ACT_LRELU);
      std::vector<int> s1(1, gc);
      pack_conv(r.cb[j], wc.data(), bb->second.data.data(), nullptr, gc, gc, 3, s1, ACT_LRELU);
      if (rc == HCF_OK && !r.cb[j].wpack_wino) {         // 32 / 16 input channels: below the general Winograd threshold, wanted here
        std::vector<float> pkw;
        int one = gc;
        if (gc == 32) {
          if (pack_conv_weights_wino(wc.data(), gc, gc, &one, 1, pkw, 16)) r.cb[j].wpack_wino = upload(pkw);
        } else {                                          // 16 -> 16 in a zero-padded 32-channel tile
          std::vector<float> wp((size_t)32 * gc * 9, 0.f);
          memcpy(wp.data(), wc.data(), wc.size() * sizeof(float));
          if (pack_conv_weights_wino(wp.data(), gc, 32, &one, 1, pkw, 16)) { r.cb[j].wpack_wino = upload(pkw); r.cb[j].wino_ntile = 1; }
        }
      }
      r.ca[j].wkey = pa + ".weight+" + pb + ".weight[:, :" + std::to_string(ka) + "]";
      r.cb[j].wkey = pb + ".weight[:, " + std::to_string(ka) + ":]";
      r.fat[j] = rc == HCF_OK && r.ca[j].wpack_wino && r.cb[j].wpack_wino;
    }
  }

  void build_condflow(CondFlow& cf, const std::string& p, int level) {
    cf.level = level;
    cf.C = level_channels(level);
    cf.ns = split_channels(level);
    cf.Ca = cf.C - cf.ns;
    cf.nlc = cfg.L - 1 - level;
    const int nf = cfg.rrdb_nf, gc = cfg.rrdb_gc, cc = cond_ch();
    std::vector<int> srcs;
    srcs.push_back(cf.ns);
    for (int i = 0; i < cf.nlc; ++i) srcs.push_back(cc);
    build_conv(cf.conv_first, p + ".conv_first", cf.ns + cc * cf.nlc, nf, srcs, ACT_NONE);
    cf.trunk0.resize(cfg.rrdb_nb[0]);
    for (int n = 0; n < cfg.rrdb_nb[0]; ++n)
      for (int r = 0; r < 3; ++r)
        build_rdb(cf.trunk0[n].r[r], p + ".RRDB_trunk0." + std::to_string(n) + ".RDB" + std::to_string(r + 1), nf, gc);
    cf.trunk1.resize(cfg.rrdb_nb[1]);
    for (int n = 0; n < cfg.rrdb_nb[1]; ++n)
      for (int r = 0; r < 3; ++r)
        build_rdb(cf.trunk1[n].r[r], p + ".RRDB_trunk1." + std::to_string(n) + ".RDB" + std::to_string(r + 1), nf, gc);
    build_conv(cf.trunk_conv1, p + ".trunk_conv1", nf, nf, srcs2(nf, 0), ACT_NONE);
    cf.steps.resize(cfg.after[level]);
    for (int k = 0; k < cfg.after[level]; ++k)
      build_step(cf.steps[k], p + ".additional_flow_steps." + std::to_string(k), cf.Ca, cc, cfg.c_perm, cfg.c_coupling,
                 cfg.c_nn_module, cfg.c_hidden, true);
    build_conv_zeros(cf.head, p + ".f", cc, cf.Ca * 2, srcs2(cc, 0));
  }

  // Walk the module tree in the reference's registration order. spec_mode: record keys only.
  int build() {
    levels.clear();
    levels.resize(cfg.L);
    int idx = 0;
    int C = cfg.in_nc;
    for (int level = 0; level < cfg.L; ++level) {
      Level& lv = levels[level];
      if (cfg.squeeze == HCF_SQUEEZE_HAAR) {
        const float* hw = P("flow.layers." + std::to_string(idx) + ".haar_weights", {4 * C, 1, 2, 2});
        if (hw) {
          // frozen +-1 pattern (Basic.py:455-468); anything else is not a Haar transform
          for (int c = 0; c < 4 * C; ++c)
            for (int i = 0; i < 2; ++i)
              for (int j = 0; j < 2; ++j) {
                const int k = c % 4;
                const bool neg = (k == 1 && j == 1) || (k == 2 && i == 1) || (k == 3 && (i != j));
                if (hw[(c * 2 + i) * 2 + j] != (neg ? -1.f : 1.f))
                  return fail(HCF_ERR_UNSUPPORTED, "haar_weights differ from the fixed Haar pattern");
              }
        }
      }
      idx++;
      C *= 4;
      lv.C = C;
      const int nmain = cfg.K[level] - cfg.after[level];
      lv.steps.resize(nmain);
      for (int k = 0; k < nmain; ++k) {
        const bool lrv = sr() ? true : (k % 2 == 0);
        build_step(lv.steps[k], "flow.layers." + std::to_string(idx), C, 0, cfg.perm, cfg.coupling, cfg.nn_module,
                   cfg.hidden, lrv);
        idx++;
      }
      lv.ns = split_channels(level);
      idx++;   // Split
      C = lv.ns;
    }
    for (int level = 0; level < cfg.L; ++level)
      build_condflow(levels[level].cf, "flow.level" + std::to_string(level) + "_condFlow", level);
    return rc;
  }

  void free_weights() {
    for (float* p : dev_allocs) hipFree(p);
    dev_allocs.clear();
    ++refresh_gen;                           // the cached refresh job tables point into these allocations
    unit_dev = nullptr;
    weight_bytes = 0;
  }

  // ---------------------------------------------------------------- execution helpers
  hipStream_t st = nullptr;
  bool dry() const { return arena.dry; }

  struct Buf {
    float* p; int C; int cs;
    View v(int c0, int n, int up = 0) const { return mkview(p, cs, c0, n, up); }
    View all() const { return mkview(p, cs, 0, C, 0); }
  };
  Buf alloc(int B, int H, int W, int C) {
    Buf b;
    b.C = C;
    b.cs = ru4(C);
    b.p = arena.alloc((size_t)B * H * W * b.cs);
    return b;
  }

  int B_ = 0;   // batch of the running pass

  // f16x3 mode can run FCN conv1 (3x3 -> 64) and conv2 (1x1 64 -> 64) as ONE launch
  bool can_fuse_fcn(const Conv& c1, const Conv& c2) const {
    static const bool off = getenv("HCF_NO_FUSE_FCN") != nullptr;      // debugging aid, read once
    if (off) return false;
    if (taping) return false;                         // the backward pass needs the intermediate tensor
    return use_f16 && c1.wpack16 && c2.wpack16 && c1.taps == 9 && c2.taps == 1 && c1.cout == 64 && c2.cout == 64 &&
           c2.nsrc == 1 && c2.src_n[0] == 64;
  }

  struct FatExtra { View out2; int act2; bool pre; const float* w4f_frag; };   // Winograd-only routing: fat launches (see Rdb), fused 1x1 layer (Step::c1w)
  void run_conv(const Conv& cv, std::vector<View> srcs, int H, int W, View out, View res1 = mkview(nullptr, 0, 0, 0),
                float rs1 = 0.f, View res2 = mkview(nullptr, 0, 0, 0), float rs2 = 0.f, const Conv* fuse2 = nullptr,
                const StepArgs* tail = nullptr, const FatExtra* fat = nullptr) {
    if (rc != HCF_OK) return;
    if ((int)srcs.size() != cv.nsrc) { fail(HCF_ERR_STATE, "internal: conv source count"); return; }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < cv.nsrc; ++i) {
      a.src[i] = srcs[i];
      if (srcs[i].n != cv.src_n[i]) { fail(HCF_ERR_STATE, "internal: conv source width"); return; }
    }
    for (int i = cv.nsrc; i < kMaxSrc; ++i) a.src[i] = srcs[0];
    a.nsrc = cv.nsrc;
    a.B = B_; a.H = H; a.W = W;
    a.wpack = cv.wpack; a.nchunk = cv.nchunk; a.bias = cv.bias; a.scale = cv.scale; a.act = cv.act;
    a.out = out; a.out.n = cv.cout;
    a.res1 = res1; a.rs1 = rs1; a.res2 = res2; a.rs2 = rs2;
    if (fat) { a.out2 = fat->out2; a.act_t2 = fat->act2; a.res1_pre = fat->pre ? 1 : 0; }
    a.wino_ntile = cv.wino_ntile;
    if (dry()) return;
    if (probe_on) { probe_conv(cv, srcs, H, W); ++launch_seq; }      // probe launches sit before this conv: it is never chained across them
    if (prof) {
      if (prof_used == prof_events.size()) {
        ProfRec r;
        hipEventCreate(&r.e0);
        hipEventCreate(&r.e1);
        prof_events.push_back(r);
      }
      prof_events[prof_used].taps = cv.taps;
      prof_events[prof_used].nt = cv.npad / 32;
      {   // kernel variant: 0 plain, 1 + fused 1x1 second layer, 2 + fused flow-step tail, 3 a source is read upsampled
        bool up = false;
        for (int i = 0; i < cv.nsrc; ++i) up = up || srcs[i].up > 0;
        prof_events[prof_used].kind = tail ? 2 : fuse2 ? 1 : up ? 3 : 0;
      }
      prof_events[prof_used].flops = (cv.flops_per_pixel + (fuse2 ? fuse2->flops_per_pixel : 0.0)) * (double)B_ * H * W;
      {   // algorithmic HBM bytes: every source window read once, output written once, residuals read once, weights once
        double px_bytes = 4.0 * cv.cout * (1 + (res1.p ? 1 : 0) + (res2.p ? 1 : 0));
        for (int i = 0; i < cv.nsrc; ++i) px_bytes += 4.0 * cv.src_n[i] / (double)(1 << (2 * srcs[i].up));
        prof_events[prof_used].bytes = px_bytes * (double)B_ * H * W + cv.flops_per_pixel * 2.0;   // + weights (4 B each)
      }
      prof_events[prof_used].chained = (prof_used > 0 && prof_last_seq == launch_seq) ? 1 : 0;
      if (!prof_events[prof_used].chained) hipEventRecord(prof_events[prof_used].e0, st);
    }
    ++launch_seq;
    int r = HCF_ERR_UNSUPPORTED;
    const bool w4f = fat && fat->w4f_frag && fuse2;    // Winograd conv1 + the 1x1 layer in its epilogue
    if (use_f16 && cv.wpack_wino && (!fuse2 || w4f) && !tail && !wino_stale && !(g_f16x3_ablation & 256)) {
      a.ovf = ovf_flag;
      a.zeros = reinterpret_cast<const float*>(ovf_flag) + 16;
      if (w4f) { a.wf1x1 = fat->w4f_frag; a.bias2 = fuse2->bias; a.scale2 = fuse2->scale; a.act2 = fuse2->act; }
      r = launch_conv_wino(a, cv.wpack_wino, st);      // HCF_ERR_UNSUPPORTED: this call's views do not qualify
      if (r == HCF_OK && prof) prof_events[prof_used].kind = w4f ? 6 : (fat && fat->pre) ? 7 : 4;
    }
    if (fat && r != HCF_OK) { fail(HCF_ERR_STATE, "internal: a fat dense-block launch did not take the Winograd kernel"); return; }
    if (r != HCF_ERR_UNSUPPORTED) {
    } else if (use_f16 && cv.wpack16 && (cv.taps == 9 || (cv.taps == 1 && !fuse2 && !tail))) {      // (a stand-alone 1x1: the training passes)
      a.wpack = cv.wpack16;
      a.ovf = ovf_flag;
      a.zeros = reinterpret_cast<const float*>(ovf_flag) + 16;
      if (fuse2) {
        a.w2 = fuse2->wpack16; a.bias2 = fuse2->bias; a.scale2 = fuse2->scale; a.act2 = fuse2->act;
      }
      if (tail) {
        a.tz = tail->z; a.tzo = tail->out; a.tmat = tail->mat; a.tbias = tail->an_bias; a.tmul = tail->an_mul;
        a.tC = tail->C; a.tns = tail->ns; a.tmode = tail->mode;
        a.tzpad = tail->zpad16; a.tzpad_n = tail->zpad_n;
      }
      r = HCF_ERR_UNSUPPORTED;
      if (fuse2 && !tail) {                 // one K chunk: the persistent small-K form (res1 = pre-activation term, if any)
        r = launch_fcn12(a, st);
        if (r == HCF_OK && prof) prof_events[prof_used].kind = 5;
        if (r == HCF_ERR_UNSUPPORTED && res1.p) { fail(HCF_ERR_STATE, "internal: pre-activation term without the fcn12 kernel"); return; }
      }
      if (r == HCF_ERR_UNSUPPORTED) r = launch_conv_f16x3(a, cv.taps, st);
      if (r == HCF_ERR_UNSUPPORTED && cv.taps == 1) { a.wpack = cv.wpack; r = launch_conv(a, cv.taps, st); }      // views the one-tap form does not take
    } else {
      r = launch_conv(a, cv.taps, st);
    }
    if (prof) {
      hipEventRecord(prof_events[prof_used].e1, st);
      prof_used++;
      prof_last_seq = launch_seq;
    }
    if (r != HCF_OK) fail(r, "conv launch failed");
  }

  // ---- range probe (hcf_debug_range_probe): max |x| of every conv's inputs and, for the layers that have a Winograd pack,
  // max |B^T d B| of the transformed input patches -- the values the f16x3 kernels must split (limit 65504). Debug only.
  bool probe_on = false;
  struct ProbeRec { std::string key; int cin, cout, H, W, f16, wino; };
  std::vector<ProbeRec> probe_recs;
  float* probe_dev = nullptr;
  static constexpr int kProbeCap = 16384;
  void probe_conv(const Conv& cv, const std::vector<View>& srcs, int H, int W) {
    if ((int)probe_recs.size() >= kProbeCap || !probe_dev) return;
    float* slot = probe_dev + 2 * probe_recs.size();
    int cin = 0;
    for (int i = 0; i < cv.nsrc; ++i) {
      View v = srcs[i];
      const int up = v.up;
      v.up = 0;
      cin += v.n;
      launch_absmax(v, B_, H >> up, W >> up, slot, st);
      if (cv.wpack_wino && up == 0) launch_wino_vmax(v, B_, H, W, slot + 1, st);
    }
    probe_recs.push_back({cv.wkey, cin, cv.cout, H, W, (use_f16 && cv.wpack16 && cv.taps == 9) ? 1 : 0, cv.wpack_wino ? 1 : 0});
  }

#define HCF_LAUNCH(expr)                                    \
  do {                                                      \
    if (rc == HCF_OK && !dry()) {                           \
      ++launch_seq;                                         \
      const int r_ = (expr);                                \
      if (r_ != HCF_OK) fail(r_, "launch failed: " #expr);  \
    }                                                       \
  } while (0)

  // ---- ActNorm data-dependent init ----------------------------------------------------------------------------
  bool an_wants(const std::string& key) const { return an_active && !arena.dry && an_pending.count(key) > 0; }

  // `initialize_parameters` on the tensor that reaches the layer: false when the stored bias is non-zero ("already
  // trained", ActNorms.py:33-35); otherwise bias = -mean, logs = log(scale / (sqrt(var) + 1e-6)) over (B, H, W)
  // (:37-43; scale = 1 for every ActNorm of these nets) are written to the host parameter table.
  bool an_fit(const std::string& key, View v, int H, int W) {
    an_pending.erase(key);
    auto ib = params.find(key + ".bias"), il = params.find(key + ".logs");
    if (ib == params.end() || il == params.end() || (int)ib->second.data.size() != v.n) {
      fail(HCF_ERR_KEY, "ActNorm init: unknown layer " + key);
      return false;
    }
    for (float x : ib->second.data)
      if (x != 0.f) return false;
    if (!stats_dev && hipMalloc((void**)&stats_dev, sizeof(double) * 512) != hipSuccess) {
      fail(HCF_ERR_NOMEM, "hipMalloc failed for the ActNorm statistics");
      return false;
    }
    std::vector<double> hs(2 * (size_t)v.n);
    if (launch_channel_stats(v, B_, H, W, stats_dev, st) != HCF_OK ||
        hipMemcpyAsync(hs.data(), stats_dev, sizeof(double) * hs.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      fail(HCF_ERR_HIP, "ActNorm statistics failed for " + key);
      return false;
    }
    const double n = (double)B_ * H * W;
    for (int c = 0; c < v.n; ++c) {
      const double mean = hs[c] / n;
      const double var = std::max(0.0, hs[v.n + c] / n - mean * mean);
      ib->second.data[c] = (float)(-mean);
      il->second.data[c] = (float)log(1.0 / (sqrt(var) + 1e-6));
    }
    an_fitted.insert(key);
    return true;
  }

  bool an_upload(float* dst, const std::vector<float>& v) {
    if (hipMemcpy(dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      fail(HCF_ERR_HIP, "hipMemcpy H2D failed (ActNorm init)");
      return false;
    }
    return true;
  }

  void an_refit_step(Step& s, View z, int H, int W) {
    if (!an_fit(s.an_key, z, H, W)) return;
    const std::vector<float>& b = params[s.an_key + ".bias"].data;
    const std::vector<float>& l = params[s.an_key + ".logs"].data;
    std::vector<float> bias(s.cmax, 0.f), mi(s.cmax, 0.f), mf(s.cmax, 0.f);
    double sumlogs = 0;
    for (int c = 0; c < s.C; ++c) {
      bias[c] = b[c]; mi[c] = expf(-l[c]); mf[c] = expf(l[c]);
      sumlogs += (double)l[c];
    }
    if (an_upload(s.bias, bias) && an_upload(s.mul_inv, mi) && an_upload(s.mul_fwd, mf)) s.ld_const = sumlogs + s.lad;
  }

  void an_refit_conv(Conv& cv, View y, int H, int W) {
    if (!an_fit(cv.an_key, y, H, W)) return;
    const std::vector<float>& b = params[cv.an_key + ".bias"].data;
    const std::vector<float>& l = params[cv.an_key + ".logs"].data;
    std::vector<float> bias(cv.npad, 0.f), sc(cv.npad, 1.f);
    for (int c = 0; c < cv.cout; ++c) { bias[c] = b[c]; sc[c] = expf(l[c]); }
    an_upload(cv.bias, bias) && an_upload(cv.scale, sc);
  }

  // Basic.Conv2d (conv -> ActNorm -> ReLU). While an init pass is armed and this layer is pending, the raw conv
  // output is produced first (identity epilogue), its statistics fix bias / logs, then the layer runs normally.
  void run_conv_an(const Conv& cv, std::vector<View> in, int H, int W, View out) {
    if (an_wants(cv.an_key)) {
      if (!unit_dev) {
        std::vector<float> u(512, 0.f);
        for (int i = 256; i < 512; ++i) u[i] = 1.f;
        unit_dev = upload(u);
      }
      if (unit_dev) {
        Conv raw = cv;
        raw.bias = unit_dev; raw.scale = unit_dev + 256; raw.act = ACT_NONE;
        run_conv(raw, in, H, W, out);
        View y = out; y.n = cv.cout;
        an_refit_conv(const_cast<Conv&>(cv), y, H, W);
      }
    }
    run_conv(cv, in, H, W, out);
  }

  struct Scratch {      // per-level temporaries
    Buf h1, h2, hout, grow, t1, t2, x, f0, rgrow, fatp, zpad, zpadd;
  };

  // coupling network f(z1 [, u]) -> sc.hout   (FCN: Basic.py:441-447, DenseBlock: :349-356)
  // f16x3 mode: the last conv of an FCN coupling net can finish the inverse flow step in its epilogue
  bool can_fuse_tail(const Step& s) const {
    static const bool off = getenv("HCF_NO_FUSE_TAIL") != nullptr;     // debugging aid, read once
    if (off) return false;
    if (taping) return false;
    const Conv& c = s.c[2];
    return use_f16 && s.fcn && c.wpack16 && c.taps == 9 && s.f_out <= 32 && s.cmax <= 24;   // the 48-channel variant spills
  }

  // conditional FCN coupling net as Winograd conv1 + 1x1 conv2 in its epilogue (Step::c1w)?
  bool zpad_valid = false;       // sc.zpad already holds the coming step's z1 (written by the tail of the step before)
  // the Winograd kernels address their sources with 31-bit byte offsets (an out-of-range offset IS the conv padding): schedules
  // that cannot fall back per launch (fat dense-block pairs, the FCN form with the 1x1 epilogue) must know beforehand
  bool wino_offsets_ok(int H, int W, int cs) const { return (long long)B_ * H * W * cs * 4 < 0x7fffe000LL; }
  bool w4f_ok(const Step& s, const View* u, int H, int W, const Scratch& sc) const {
    return s.fcn && s.w4f_frag && u && wino_offsets_ok(H, W, std::max(u->cs, 16)) && s.w4f_frag && s.cond > 0 && u && s.mode == CPL_AFFINE && can_fuse_fcn(s.c[0], s.c[1]) && !fat_stale && !wino_stale &&
           !(g_f16x3_ablation & (256 | 512)) && u->up == 0 && sc.zpad.p && conv_wino_rounds_ok(B_, H, W, 2);
  }

  void run_coupling_net(const Step& s, View z1, const View* u, int H, int W, Scratch& sc, const StepArgs* tail = nullptr) {
    std::vector<View> in;
    in.push_back(z1);
    if (s.cond > 0) {
      if (!u) { fail(HCF_ERR_UNSUPPORTED, "conditional coupling without a condition tensor"); return; }
      in.push_back(*u);
    }
    if (s.fcn) {
      if (w4f_ok(s, u, H, W, sc)) {
        // conv1 on the Winograd kernel (sources: z1 padded to one 16-channel chunk, the features), conv2 in its epilogue
        const View none = mkview(nullptr, 0, 0, 0);
        if (!zpad_valid) HCF_LAUNCH(launch_copy_pad16(z1, sc.zpad.p, B_, H, W, st));    // (else: the previous step's tail wrote it)
        zpad_valid = false;
        const FatExtra fx = {none, ACT_NONE, false, s.w4f_frag};
        run_conv(s.c1w, {sc.zpad.v(0, 16), *u}, H, W, sc.h2.v(0, s.hid), none, 0.f, none, 0.f, &s.c[1], nullptr, &fx);
      } else if (can_fuse_fcn(s.c[0], s.c[1])) {
        const View none = mkview(nullptr, 0, 0, 0);
        run_conv(s.c[0], in, H, W, sc.h2.v(0, s.hid), none, 0.f, none, 0.f, &s.c[1]);
      } else {
        run_conv_an(s.c[0], in, H, W, sc.h1.v(0, s.hid));
        run_conv_an(s.c[1], {sc.h1.v(0, s.hid)}, H, W, sc.h2.v(0, s.hid));
      }
      {
        const View none = mkview(nullptr, 0, 0, 0);
        run_conv(s.c[2], {sc.h2.v(0, s.hid)}, H, W, sc.hout.v(0, s.f_out), none, 0.f, none, 0.f, nullptr, tail);
      }
    } else {
      // DenseBlock: conv i >= 1 in Winograd form over [z1 padded | growth] where that form exists and this level qualifies
      bool need64 = false;
      for (int i = 1; i < 5; ++i) need64 = need64 || (s.cw[i].wpack_wino && (s.cw[i].wino_ntile == 2 || s.cw[i].cout == 64));
      const bool dw = s.dw_pad > 0 && s.cond == 0 && use_f16 && !taping && !fat_stale && !wino_stale && !(g_f16x3_ablation & (256 | 1024)) &&
                      sc.zpadd.p && sc.zpadd.C >= s.dw_pad && conv_wino_rounds_ok(B_, H, W, 1) && (!need64 || conv_wino_rounds_ok(B_, H, W, 2)) &&
                      wino_offsets_ok(H, W, std::max(sc.grow.cs, sc.zpadd.cs));
      if (dw) HCF_LAUNCH(launch_copy_pad(z1, sc.zpadd.v(0, s.dw_pad), B_, H, W, st));
      for (int i = 0; i < 5; ++i) {
        const View out_i = i < 4 ? sc.grow.v(i * s.hid, s.hid) : sc.hout.v(0, s.f_out);
        if (dw && s.cw[i].wpack_wino) {
          const View none = mkview(nullptr, 0, 0, 0);
          const FatExtra fx = {none, ACT_NONE, false, nullptr};       // (no per-launch fallback: the direct packs have another source list)
          run_conv(s.cw[i], {sc.zpadd.v(0, s.dw_pad), sc.grow.v(0, i * s.hid)}, H, W, out_i, none, 0.f, none, 0.f, nullptr, nullptr, &fx);
          continue;
        }
        std::vector<View> srcs = in;
        if (i > 0) srcs.push_back(sc.grow.v(0, i * s.hid));
        run_conv(s.c[i], srcs, H, W, out_i);
      }
    }
  }

  View step_z1(const Step& s, const Buf& z) const {
    if (s.mode == CPL_AFFINE) return z.v(0, s.ns);
    return z.v(3, s.C - 3);
  }

  // FlowStep.reverse_flow (FlowStep.py:53-64), in place on z
  // `next` = the step that runs after this one on the same z (or null): when it takes the Winograd form of its FCN, this
  // step's tail also writes next's z1 as the padded 16-channel tensor that form reads (saves a copy launch per step)
  void run_step_inverse(const Step& s, const Buf& z, const View* u, int H, int W, Scratch& sc, const Step* next = nullptr) {
    StepArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B_; a.H = H; a.W = W; a.C = s.C; a.ns = s.ns; a.mode = s.mode;
    a.z = z.all(); a.h = sc.hout.v(0, s.f_out); a.out = z.all();
    a.mat = s.has_mat ? s.mat_inv : nullptr; a.an_bias = s.bias; a.an_mul = s.mul_inv;
    const bool pad_next = next && next->C == s.C && w4f_ok(*next, u, H, W, sc);
    if (pad_next) { a.zpad16 = sc.zpad.p; a.zpad_n = next->ns; }
    if (can_fuse_tail(s)) {
      run_coupling_net(s, step_z1(s, z), u, H, W, sc, &a);      // conv3's epilogue finishes the step
    } else {
      run_coupling_net(s, step_z1(s, z), u, H, W, sc);
      HCF_LAUNCH(launch_step_tail_inv(a, st));
    }
    zpad_valid = pad_next && !dry() && rc == HCF_OK;
  }

  // FlowStep.normal_flow (FlowStep.py:40-51), in place on z; partial slot advanced when `partial`
  void run_step_forward(const Step& s, const Buf& z, const View* u, int H, int W, Scratch& sc, float* partial,
                        int pstride, int& pslot) {
    StepArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B_; a.H = H; a.W = W; a.C = s.C; a.ns = s.ns; a.mode = s.mode;
    a.z = z.all(); a.out = z.all();
    a.mat = s.has_mat ? s.mat_fwd : nullptr; a.an_bias = s.bias; a.an_mul = s.mul_fwd;
    if (an_wants(s.an_key)) an_refit_step(const_cast<Step&>(s), z.all(), H, W);     // same device arrays, new contents
    HCF_LAUNCH(launch_step_head_fwd(a, st));
    run_coupling_net(s, step_z1(s, z), u, H, W, sc);
    a.h = sc.hout.v(0, s.f_out);
    a.mat = nullptr;
    if (partial && s.mode == CPL_AFFINE) {
      a.partial = partial + pslot;
      a.partial_stride = pstride;
      pslot += step_blocks_per_sample(H, W);
    }
    HCF_LAUNCH(launch_step_couple_fwd(a, st));
  }

  // ResidualDenseBlock (Basic.py:379-385) with optional second residual (RRDB tail, :394-398)
  bool fat_stale = false;        // the fat packs are built from the host weights only (hcf_finalize); a device-side refresh disables them
  void run_rdb(const Rdb& r, View xin, const Buf& grow, int H, int W, View out, View res2, float rs2, const Buf* fatp = nullptr) {
    const int gc = cfg.rrdb_gc;
    // Fat pairs (profiles/r03_notes.md): measured per-launch costs say pair (3, 4) pays at every size, pair (1, 2) only where the
    // launches are short (the 64-channel kernel's fixed cost: 350 + 146 us against 199 + 267 at 16 x 320^2, 82 + 43 against
    // 61 + 79 at 16 x 160^2) -> up to 200 x 200 pixels per sample. The rule looks at the SAMPLE size, not at the batch: a sample's
    // bits must not depend on how many others share its launch (tests/test_gpu_nets.py: batch independence).
    const bool fat_ok = fatp && fatp->p && use_f16 && !taping && !fat_stale && !wino_stale && !(g_f16x3_ablation & 256) &&
                        (gc == 16 || conv_wino_rounds_ok(B_, H, W, 2)) && conv_wino_rounds_ok(B_, H, W, 1) &&
                        wino_offsets_ok(H, W, std::max(std::max(xin.cs, grow.cs), fatp->cs));
    static const long long fat12_pixels = getenv("HCF_FAT12_PIXELS") ? atoll(getenv("HCF_FAT12_PIXELS")) : 40000;   // experiment knob
    // (16-channel blocks: both halves of a pair run on the 32-channel kernel, whose fixed cost the per-conv schedule pays too:
    //  pair (1, 2) at every size)
    const bool use_fat[2] = {fat_ok && r.fat[0] && (gc == 16 || (long long)H * W <= fat12_pixels), fat_ok && r.fat[1]};
    if (use_fat[0] || use_fat[1]) {
      const View none = mkview(nullptr, 0, 0, 0);
      for (int j = 0; j < 2; ++j) {
        std::vector<View> srcs;
        srcs.push_back(xin);
        if (j > 0) srcs.push_back(grow.v(0, 2 * j * gc));
        if (use_fat[j]) {
          FatExtra fa = {fatp->v(0, gc), ACT_NONE, false};        // tile 0 -> x_{2j+1} (bias, LeakyReLU), tile 1 -> raw partial of conv 2j+2
          run_conv(r.ca[j], srcs, H, W, grow.v(2 * j * gc, gc), none, 0.f, none, 0.f, nullptr, nullptr, &fa);
          FatExtra fb = {none, ACT_NONE, true};                   // x_{2j+2} = lrelu(W[x_{2j+1}] * x_{2j+1} + partial + bias)
          run_conv(r.cb[j], {grow.v(2 * j * gc, gc)}, H, W, grow.v((2 * j + 1) * gc, gc), fatp->v(0, gc), 0.f, none, 0.f, nullptr,
                   nullptr, &fb);
        } else {
          run_conv(r.c[2 * j], srcs, H, W, grow.v(2 * j * gc, gc));
          std::vector<View> s2;
          s2.push_back(xin);
          s2.push_back(grow.v(0, (2 * j + 1) * gc));
          run_conv(r.c[2 * j + 1], s2, H, W, grow.v((2 * j + 1) * gc, gc));
        }
      }
      run_conv(r.c[4], {xin, grow.v(0, 4 * gc)}, H, W, out, xin, 0.2f, res2, rs2);
      return;
    }
    for (int i = 0; i < 4; ++i) {
      std::vector<View> srcs;
      srcs.push_back(xin);
      if (i > 0) srcs.push_back(grow.v(0, i * gc));
      run_conv(r.c[i], srcs, H, W, grow.v(i * gc, gc));
    }
    run_conv(r.c[4], {xin, grow.v(0, 4 * gc)}, H, W, out, xin, 0.2f, res2, rs2);
  }

  void run_rrdb(const Rrdb& rr, View x0, View out, int H, int W, Scratch& sc) {
    const int nf = cfg.rrdb_nf;
    const View none = mkview(nullptr, 0, 0, 0);
    run_rdb(rr.r[0], x0, sc.rgrow, H, W, sc.t1.v(0, nf), none, 0.f, &sc.fatp);
    run_rdb(rr.r[1], sc.t1.v(0, nf), sc.rgrow, H, W, sc.t2.v(0, nf), none, 0.f, &sc.fatp);
    run_rdb(rr.r[2], sc.t2.v(0, nf), sc.rgrow, H, W, out, x0, 0.2f, &sc.fatp);
  }

  // ConditionalFlow.get_conditional_feature_SR / _Rescaling (ConditionalFlow.py:99-110) -> cfbuf
  // HCF_TRUNK_MB=n (experiment, profiles/r03_notes.md): the RRDB trunk runs n samples at a time so that a dense block's
  // 192-channel slab (78.6 MB per sample at 320 x 320) stays inside the 256 MB MALL between its five convs. Off by default:
  // the persistent Winograd kernels then see 3..6 units per CU and lose more to the ragged last round than the cache gives.
  void run_cond_features(const CondFlow& cf, std::vector<View> u, int H, int W, const Buf& cfbuf, Scratch& sc) {
    static const int trunk_mb = getenv("HCF_TRUNK_MB") ? atoi(getenv("HCF_TRUNK_MB")) : 0;
    if (trunk_mb > 0 && trunk_mb < B_ && !taping && !an_active) {
      const int Bfull = B_;
      auto shiftv = [&](View v, int b0) { v.p += (size_t)b0 * (H >> v.up) * (W >> v.up) * v.cs; return v; };
      auto shiftb = [&](Buf b, int b0) { if (b.p) b.p += (size_t)b0 * H * W * b.cs; return b; };
      for (int b0 = 0; b0 < Bfull; b0 += trunk_mb) {
        B_ = std::min(trunk_mb, Bfull - b0);
        std::vector<View> ub;
        for (const View& v : u) ub.push_back(shiftv(v, b0));
        Scratch sb = sc;
        sb.t1 = shiftb(sc.t1, b0); sb.t2 = shiftb(sc.t2, b0); sb.x = shiftb(sc.x, b0); sb.f0 = shiftb(sc.f0, b0);
        sb.rgrow = shiftb(sc.rgrow, b0); sb.fatp = shiftb(sc.fatp, b0);
        run_cond_features_all(cf, ub, H, W, shiftb(cfbuf, b0), sb);
      }
      B_ = Bfull;
      return;
    }
    run_cond_features_all(cf, u, H, W, cfbuf, sc);
  }
  void run_cond_features_all(const CondFlow& cf, std::vector<View> u, int H, int W, const Buf& cfbuf, Scratch& sc) {
    const int nf = cfg.rrdb_nf;
    run_conv(cf.conv_first, u, H, W, sc.f0.v(0, nf));
    View cur = sc.f0.v(0, nf);
    const View f1 = sr() ? cfbuf.v(0, nf) : sc.x.v(0, nf);
    for (size_t n = 0; n < cf.trunk0.size(); ++n) {
      run_rrdb(cf.trunk0[n], cur, f1, H, W, sc);
      cur = f1;
    }
    if (sr() && cf.trunk0.empty()) {
      HCF_LAUNCH(launch_copy_view(cur, f1, B_, H, W, st));
      cur = f1;
    }
    const View x = sc.x.v(0, nf);
    for (size_t n = 0; n < cf.trunk1.size(); ++n) {
      run_rrdb(cf.trunk1[n], cur, x, H, W, sc);
      cur = x;
    }
    const View f2 = sr() ? cfbuf.v(nf, nf) : cfbuf.v(0, nf);
    run_conv(cf.trunk_conv1, {cur}, H, W, f2, sc.f0.v(0, nf), 1.0f);
  }

  Scratch alloc_scratch(int H, int W) {
    Scratch sc;
    int hid = std::max(cfg.hidden, cfg.c_hidden);
    int fo = 4;
    for (const Level& lv : levels) {
      for (const Step& s : lv.steps) fo = std::max(fo, s.f_out);
      for (const Step& s : lv.cf.steps) fo = std::max(fo, s.f_out);
      fo = std::max(fo, lv.cf.Ca * 2);
    }
    int dense_hid = 0;     // DenseBlock coupling nets need a growth slab, FCN ones do not
    for (const Level& lv : levels) {
      for (const Step& s : lv.steps) if (!s.fcn) dense_hid = std::max(dense_hid, s.hid);
      for (const Step& s : lv.cf.steps) if (!s.fcn) dense_hid = std::max(dense_hid, s.hid);
    }
    const int nf = cfg.rrdb_nf;
    sc.h1 = alloc(B_, H, W, hid);
    sc.h2 = alloc(B_, H, W, hid);
    sc.hout = alloc(B_, H, W, fo);
    sc.grow = alloc(B_, H, W, std::max(4, 4 * dense_hid));
    sc.t1 = alloc(B_, H, W, nf);
    sc.t2 = alloc(B_, H, W, nf);
    sc.x = alloc(B_, H, W, nf);
    sc.f0 = alloc(B_, H, W, nf);
    sc.rgrow = alloc(B_, H, W, 4 * cfg.rrdb_gc);
    sc.fatp = alloc(B_, H, W, cfg.rrdb_gc);          // stored partial sum of the fat dense-block launches
    sc.zpad = alloc(B_, H, W, 16);                   // z1 of a conditional coupling net, zero padded (Step::c1w)
    int dwp = 0;
    for (const Level& lv : levels) {
      for (const Step& s : lv.steps) dwp = std::max(dwp, s.dw_pad);
      for (const Step& s : lv.cf.steps) dwp = std::max(dwp, s.dw_pad);
    }
    sc.zpadd = alloc(B_, H, W, std::max(4, dwp));    // ... of a DenseBlock coupling net (Step::cw)
    return sc;
  }

  int ensure_arena(size_t need) {
    if (need <= arena.cap) return HCF_OK;
    if (arena.base) {
      hipStreamSynchronize(st);
      hipFree(arena.base);
      arena.base = nullptr;
      arena.cap = 0;
    }
    void* p = nullptr;
    if (hipMalloc(&p, need) != hipSuccess) return fail(HCF_ERR_NOMEM, "hipMalloc failed for the activation arena");
    arena.base = (char*)p;
    arena.cap = need;
    return HCF_OK;
  }

  // ---------------------------------------------------------------- inverse pass
  // FlowNet.reverse_flow (FlowNet_SR_x4.py:106-123, FlowNet_SR_x8.py:121-144, FlowNet_Rescaling_x4.py:111-128)
  void pass_inverse(const float* lr, const float* const* eps, int n_eps, float tau, uint64_t seed, int64_t sample0, float* out,
                    int B, int h, int w, uint32_t flags) {
    B_ = B;
    arena.top = 0;
    const int L = cfg.L;
    // HCF_FLAG_KEEP_COND / HCF_FLAG_REUSE_COND: the deepest level's conditional features and prior-head output depend on lr
    // only (FlowNet_SR_x4.py:113-115, FlowNet_SR_x8.py:129): they sit at fixed arena offsets (first allocations of the pass) and
    // survive until another kind of pass, another shape or a parameter change touches the arena.
    const bool keep_c = (flags & (HCF_FLAG_KEEP_COND | HCF_FLAG_REUSE_COND)) != 0;
    const bool reuse_c = (flags & HCF_FLAG_REUSE_COND) != 0 && cond_cache_ok(B, h, w);
    Buf hkeep;
    hkeep.p = nullptr;
    std::vector<Buf> cfb(L);
    Buf zprev;       // z buffer of the level processed before (deeper)
    zprev.p = nullptr;
    for (int level = L - 1; level >= 0; --level) {
      const Level& lv = levels[level];
      const CondFlow& cf = lv.cf;
      const int H = h << (L - 1 - level), W = w << (L - 1 - level);
      cfb[level] = alloc(B, H, W, cond_ch());
      if (level == L - 1 && keep_c) hkeep = alloc(B, H, W, cf.Ca * 2);
      Buf z = alloc(B, H, W, lv.C);
      const bool cached = reuse_c && level == L - 1;
      if (level == L - 1) {
        HCF_LAUNCH(launch_nchw_to_nhwc(lr, z.v(0, 3), B, 3, H, W, st));
      } else {
        // squeeze^-1 of the deeper level lands in z[:, :ns]   (Basic.py:143-157 / :479-487)
        const Level& dp = levels[level + 1];
        if (cfg.squeeze == HCF_SQUEEZE_HAAR)
          HCF_LAUNCH(launch_haar_inv(zprev.all(), z.v(0, lv.ns), B, dp.C, H / 2, W / 2, st));
        else
          HCF_LAUNCH(launch_unsqueeze(zprev.all(), z.v(0, lv.ns), B, dp.C, H / 2, W / 2, st));
      }
      const size_t mark = arena.top;
      Scratch sc = alloc_scratch(H, W);
      Buf a = alloc(B, H, W, cf.Ca);
      // conditional features: u = cat(z, up2(cf_{l+1}), up4(cf_{l+2}))  (FlowNet_SR_x8.py:132-137)
      std::vector<View> u;
      u.push_back(z.v(0, lv.ns));
      for (int l2 = level + 1; l2 < L; ++l2) u.push_back(cfb[l2].v(0, cond_ch(), l2 - level));
      if (!cached) run_cond_features(cf, u, H, W, cfb[level], sc);
      const View cfv = cfb[level].v(0, cond_ch());
      // prior: a = mean + exp(logs) * eps   (ConditionalFlow.py:61-64 / 88-91)
      if (!cached) {
        run_conv(cf.head, {cfv}, H, W, sc.hout.v(0, cf.Ca * 2));
        if (level == L - 1 && keep_c) HCF_LAUNCH(launch_copy_view(sc.hout.v(0, cf.Ca * 2), hkeep.all(), B, H, W, st));
      }
      {
        GaussArgs g;
        memset(&g, 0, sizeof(g));
        g.B = B; g.H = H; g.W = W; g.C = cf.Ca;
        g.h = (cached || (level == L - 1 && keep_c)) ? hkeep.all() : sc.hout.v(0, cf.Ca * 2);
        g.rescale = sr() ? 0 : 1;
        const int draw = L - 1 - level;
        g.eps = (eps && draw < n_eps) ? eps[draw] : nullptr;
        g.tau = tau; g.seed = seed; g.offset = (uint64_t)draw; g.b0 = sample0;
        g.out = a.all();
        HCF_LAUNCH(launch_gauss_sample(g, st));
      }
      zpad_valid = false;
      for (int k = (int)cf.steps.size() - 1; k >= 0; --k) run_step_inverse(cf.steps[k], a, &cfv, H, W, sc, k > 0 ? &cf.steps[k - 1] : nullptr);
      zpad_valid = false;
      // Split reverse: z = cat(z, a)   (Basic.py:498-499)
      HCF_LAUNCH(launch_copy_view(a.all(), z.v(lv.ns, cf.Ca), B, H, W, st));
      for (int k = (int)lv.steps.size() - 1; k >= 0; --k) run_step_inverse(lv.steps[k], z, nullptr, H, W, sc);
      arena.top = mark;     // scratch of this level is dead; z and cf stay
      zprev = z;
      if (level == 0)
        HCF_LAUNCH(launch_unsqueeze_nchw(z.all(), out, B, lv.C, H, W, cfg.squeeze == HCF_SQUEEZE_HAAR ? 1 : 0,
                                         (flags & HCF_FLAG_NO_CLAMP) ? 0 : 1, st));
    }
    if (!dry() && rc == HCF_OK) {
      if (keep_c) { cc_valid = true; cc_B = B; cc_h = h; cc_w = w; cc_f16 = use_f16; cc_base = arena.base; }
      else cc_valid = false;
    }
  }
  // state of the kept conditional features (see pass_inverse)
  bool cc_valid = false, cc_f16 = false;
  int cc_B = 0, cc_h = 0, cc_w = 0;
  char* cc_base = nullptr;
  bool cond_cache_ok(int B, int h, int w) const {
    return cc_valid && cc_B == B && cc_h == h && cc_w == w && cc_f16 == use_f16 && cc_base == arena.base && arena.base != nullptr;
  }

  // ---------------------------------------------------------------- forward pass
  // FlowNet.normal_flow (FlowNet_SR_x4.py:84-101, FlowNet_SR_x8.py:91-116, FlowNet_Rescaling_x4.py:89-106)
  void pass_forward(const float* hr, const float* lr, const float* noise, float* out_lr, float* out_nll,
                    float* out_logdet, float* out_z, float* out_z1, float* out_z2, int B, int H0, int W0, uint32_t flags) {
    B_ = B;
    arena.top = 0;
    const int L = cfg.L;
    const bool want_ld = sr();
    // partial-sum slots
    int nslots = 0;
    for (int level = 0; level < L; ++level) {
      const int H = H0 >> (level + 1), W = W0 >> (level + 1);
      const int nb = step_blocks_per_sample(H, W);
      for (const Step& s : levels[level].steps) if (s.mode == CPL_AFFINE) nslots += nb;
      for (const Step& s : levels[level].cf.steps) if (s.mode == CPL_AFFINE) nslots += nb;
      nslots += nb;                      // gaussian logp
    }
    nslots += step_blocks_per_sample(H0 >> L, W0 >> L);   // Dirac term
    float* partial = nullptr;
    int pslot = 0;
    if (want_ld) {
      partial = arena.alloc((size_t)B * nslots);
      HCF_LAUNCH(launch_fill(partial, (size_t)B * nslots, 0.f, st));
    }
    std::vector<Buf> zb(L), cfb(L);
    for (int level = 0; level < L; ++level) {
      const Level& lv = levels[level];
      const int H = H0 >> (level + 1), W = W0 >> (level + 1);
      zb[level] = alloc(B, H, W, lv.C);
      cfb[level] = alloc(B, H, W, cond_ch());
      if (level == 0) {
        HCF_LAUNCH(launch_nchw_squeeze(hr, noise, cfg.quant, zb[0].all(), B, cfg.in_nc, H0, W0,
                                       cfg.squeeze == HCF_SQUEEZE_HAAR ? 1 : 0, st));
      } else {
        const Level& up = levels[level - 1];
        if (cfg.squeeze == HCF_SQUEEZE_HAAR)
          HCF_LAUNCH(launch_haar_fwd(zb[level - 1].v(0, up.ns), zb[level].all(), B, up.ns, H * 2, W * 2, st));
        else
          HCF_LAUNCH(launch_squeeze(zb[level - 1].v(0, up.ns), zb[level].all(), B, up.ns, H * 2, W * 2, st));
      }
      const size_t mark = arena.top;
      Scratch sc = alloc_scratch(H, W);
      for (size_t k = 0; k < lv.steps.size(); ++k)
        run_step_forward(lv.steps[k], zb[level], nullptr, H, W, sc, partial, nslots, pslot);
      arena.top = mark;
    }
    // hierarchical conditional prior, deepest level first (FlowNet_SR_x4.py:95-99)
    for (int level = L - 1; level >= 0; --level) {
      const Level& lv = levels[level];
      const CondFlow& cf = lv.cf;
      const int H = H0 >> (level + 1), W = W0 >> (level + 1);
      const size_t mark = arena.top;
      Scratch sc = alloc_scratch(H, W);
      Buf a = alloc(B, H, W, cf.Ca);
      std::vector<View> u;
      u.push_back(zb[level].v(0, lv.ns));
      for (int l2 = level + 1; l2 < L; ++l2) u.push_back(cfb[l2].v(0, cond_ch(), l2 - level));
      run_cond_features(cf, u, H, W, cfb[level], sc);
      const View cfv = cfb[level].v(0, cond_ch());
      HCF_LAUNCH(launch_copy_view(zb[level].v(lv.ns, cf.Ca), a.all(), B, H, W, st));
      for (size_t k = 0; k < cf.steps.size(); ++k) run_step_forward(cf.steps[k], a, &cfv, H, W, sc, partial, nslots, pslot);
      run_conv(cf.head, {cfv}, H, W, sc.hout.v(0, cf.Ca * 2));
      GaussArgs g;
      memset(&g, 0, sizeof(g));
      g.B = B; g.H = H; g.W = W; g.C = cf.Ca;
      g.h = sc.hout.v(0, cf.Ca * 2);
      g.out = a.all();
      if (sr()) {
        g.partial = partial + pslot;
        g.partial_stride = nslots;
        pslot += step_blocks_per_sample(H, W);
        HCF_LAUNCH(launch_gauss_logp(g, st));
      } else {
        g.rescale = 1;
        g.aux = (level == 0) ? out_z1 : out_z2;
        if (g.aux) HCF_LAUNCH(launch_gauss_encode(g, st));
      }
      arena.top = mark;
    }
    const int h = H0 >> L, w = W0 >> L;
    const View zlr = zb[L - 1].v(0, 3);
    if (sr()) {
      if (out_z) HCF_LAUNCH(launch_nhwc_to_nchw(zlr, out_z, B, 3, h, w, 0, st));
      float* pp = partial + pslot;
      pslot += step_blocks_per_sample(h, w);
      HCF_LAUNCH(launch_quant_logp(zlr, lr, out_lr, B, h, w, lr ? pp : nullptr, nslots, st));
      if (pslot > nslots) fail(HCF_ERR_STATE, "internal: partial slot overflow");
      // data-independent log-det terms: -ln(quant) HW + sum over steps of (sum(actnorm logs) + slogdet W) * pixels
      // (summed here, after the steps ran: an ActNorm init pass changes them on the way)
      double ld_const = -log((double)cfg.quant) * (double)H0 * W0;
      for (int level = 0; level < L; ++level) {
        const double px = (double)(H0 >> (level + 1)) * (W0 >> (level + 1));
        for (const Step& s : levels[level].steps) ld_const += s.ld_const * px;
        for (const Step& s : levels[level].cf.steps) ld_const += s.ld_const * px;
      }
      HCF_LAUNCH(launch_reduce_partials(partial, nslots, nslots, B, ld_const, (double)H0 * W0, out_logdet, out_nll, st));
    } else {
      HCF_LAUNCH(launch_nhwc_to_nchw(zlr, out_lr, B, 3, h, w, (flags & HCF_FLAG_NO_CLAMP) ? 0 : 1, st));
    }
  }

#include "hcf_engine_train.inc"

  // Sizing: a dry walk of the pass measures the arena it needs; the result is cached per (pass kind, shape, flags, precision),
  // so a steady-state call walks the graph ONCE and only enqueues (no allocation, no host-device synchronisation).
  std::map<std::vector<long long>, size_t> plan_peak;
  // f16x3 range flag: sticky on the device, mirrored into pinned host memory by an async copy at the end of every f16x3 pass;
  // hcf_check_range() is the one call that waits for it.
  int* ovf_host = nullptr;
  hipEvent_t ovf_ev = nullptr;
  bool ovf_pending = false;

  template <class F>
  int run_pass(F&& body, hipStream_t stream, uint32_t flags, std::vector<long long> key) {
    pass_flags = flags;
    if (!finalized) return fail(HCF_ERR_STATE, "hcf_finalize() has not been called");
    if (hipSetDevice(device) != hipSuccess) return fail(HCF_ERR_HIP, "hipSetDevice failed");
    rc = HCF_OK;
    st = stream;
    prof_last_seq = ~0ull;      // conv profiling: a pass never chains its first conv to the last conv of the PREVIOUS pass (host time)
    use_f16 = (precision == PREC_F16X3) && !an_active;   // also during the sizing run: fusion decisions must not differ
    key.push_back((long long)flags);
    key.push_back(use_f16 ? 1 : 0);
    size_t need = 0;
    auto it = an_active ? plan_peak.end() : plan_peak.find(key);
    if (it != plan_peak.end()) {
      need = it->second;
    } else {
      arena.dry = true;
      arena.peak = 0;
      body();
      arena.dry = false;
      if (rc != HCF_OK) return rc;
      need = arena.peak;
      if (!an_active) plan_peak[key] = need;
    }
    if (ensure_arena(need) != HCF_OK) return rc;
    if (use_f16 && !ovf_flag) {
      if (hipMalloc((void**)&ovf_flag, 256) != hipSuccess) return fail(HCF_ERR_NOMEM, "hipMalloc failed for the overflow flag");
      if (hipMemsetAsync(ovf_flag, 0, 256, st) != hipSuccess) return fail(HCF_ERR_HIP, "hipMemsetAsync failed");
      ovf_clear = false;
    }
    if (use_f16 && ovf_clear) {      // a read flag is cleared on the stream of the NEXT pass (not on the stream of the last one)
      if (hipMemsetAsync(ovf_flag, 0, sizeof(int), st) != hipSuccess) return fail(HCF_ERR_HIP, "hipMemsetAsync failed");
      ovf_clear = false;
    }
    body();
    if (use_f16 && rc == HCF_OK && !(pass_flags & HCF_FLAG_NO_RANGE_CHECK)) {
      // an activation beyond the f16 range (|x| >= 65504) cannot be split; the flag stays raised until hcf_check_range() reads it
      if (!ovf_host && hipHostMalloc((void**)&ovf_host, 64, hipHostMallocDefault) != hipSuccess)
        return fail(HCF_ERR_NOMEM, "hipHostMalloc failed for the range-flag mirror");
      if (!ovf_ev && hipEventCreateWithFlags(&ovf_ev, hipEventDisableTiming) != hipSuccess)
        return fail(HCF_ERR_HIP, "hipEventCreate failed");
      if (hipMemcpyAsync(ovf_host, ovf_flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipEventRecord(ovf_ev, st) != hipSuccess)
        return fail(HCF_ERR_HIP, "enqueueing the range-flag read-back failed");
      ovf_pending = true;
    }
    use_f16 = false;
    an_active = false;
    an_pending.clear();
    return rc;
  }
  // 1: some f16x3 pass since the last check saw an input beyond the f16 range (its outputs are invalid: re-run it with
  // HCF_PRECISION_EXACT); 0: none. Waits for the passes enqueued so far.
  int check_range(int* overflowed, uint32_t* sample_slots = nullptr) {
    if (overflowed) *overflowed = 0;
    if (sample_slots) *sample_slots = 0;
    if (ovf_latch() != HCF_OK) return rc;
    if (ovf_sticky) {
      if (overflowed) *overflowed = 1;
      if (sample_slots) *sample_slots = (ovf_unattributed || !ovf_slots) ? 0x3fffffffu : ovf_slots;
      n_fallbacks++;
      ovf_sticky = false;
      ovf_slots = 0;
      ovf_unattributed = false;
    }
    return HCF_OK;
  }
  // Fold the pending read-back of the device flag into the host-side latch. Called by check_range and before anything that
  // clears the device flag for its own use (taped passes, backward passes), so an unread overflow of an earlier inference pass
  // is never lost.
  int ovf_latch() {
    if (!ovf_pending) return HCF_OK;
    if (hipSetDevice(device) != hipSuccess) return fail(HCF_ERR_HIP, "hipSetDevice failed");
    if (hipEventSynchronize(ovf_ev) != hipSuccess) return fail(HCF_ERR_HIP, "hipEventSynchronize failed");
    ovf_pending = false;
    if (*ovf_host) {
      ovf_sticky = true;
      const uint32_t v_ = (uint32_t)*ovf_host, slots_ = (v_ >> 1) & 0x3fffffffu;
      ovf_slots |= slots_;                                         // bit 1 + (sample mod 30) of the device flag (hcf_conv_f16x3.hip)
      // "all samples": bit 31 = a writer that could not name its sample (reserved for it: none of today's four writers needs it), or a
      // read-back with bit 0 alone. Latched separately, so that slots named by ANOTHER kernel or pass cannot mask it.
      if ((v_ & 0x80000000u) || slots_ == 0) ovf_unattributed = true;
      *ovf_host = 0;
      ovf_clear = true;
    }
    return HCF_OK;
  }
  bool ovf_sticky = false, ovf_clear = false, ovf_unattributed = false;
  uint32_t ovf_slots = 0;
  uint32_t pass_flags = 0;
};

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int hcf_create(const hcf_config* cfg, hcf_engine** out) {
  if (!cfg || !out) return HCF_ERR_ARG;
  *out = nullptr;
  if (cfg->L < 2 || cfg->L > 3 || (1 << cfg->L) != cfg->scale || cfg->in_nc != 3) return HCF_ERR_UNSUPPORTED;
  if (cfg->kind != HCF_KIND_SR && cfg->kind != HCF_KIND_RESCALING) return HCF_ERR_ARG;
  if (cfg->kind == HCF_KIND_RESCALING && cfg->L != 2) return HCF_ERR_UNSUPPORTED;
  if (cfg->rrdb_nf < 4 || cfg->rrdb_nf > 96 || cfg->rrdb_gc < 1 || cfg->rrdb_gc > 96 || cfg->hidden < 1 ||
      cfg->hidden > 96 || cfg->c_hidden < 1 || cfg->c_hidden > 96 || (cfg->rrdb_nf & 3) || (cfg->rrdb_gc & 3) ||
      (cfg->hidden & 3) || (cfg->c_hidden & 3))
    return HCF_ERR_UNSUPPORTED;
  for (int l = 0; l < cfg->L; ++l)
    if (cfg->after[l] < 0 || cfg->after[l] > cfg->K[l]) return HCF_ERR_ARG;
  hcf_engine* e = new (std::nothrow) hcf_engine();
  if (!e) return HCF_ERR_NOMEM;
  e->cfg = *cfg;
  e->spec_mode = true;
  e->rc = HCF_OK;
  const int r = e->build();
  if (r != HCF_OK) { delete e; return r; }
  for (const Spec& s : e->specs) {
    HostTensor t;
    t.shape = s.shape;
    e->params[s.key] = t;
  }
  *out = e;
  return HCF_OK;
}

void hcf_destroy(hcf_engine* e) {
  if (!e) return;
  if (e->device >= 0) hipSetDevice(e->device);
  e->free_weights();
  if (e->arena.base) hipFree(e->arena.base);
  if (e->ovf_flag) hipFree(e->ovf_flag);
  if (e->ovf_host) hipHostFree(e->ovf_host);
  if (e->ovf_ev) hipEventDestroy(e->ovf_ev);
  if (e->stats_dev) hipFree(e->stats_dev);
  if (e->probe_dev) hipFree(e->probe_dev);
  if (e->garena.base) hipFree(e->garena.base);
  for (auto& t : e->slots) { if (t.a.base) hipFree(t.a.base); if (t.g.base) hipFree(t.g.base); }
  if (e->wg_scratch) hipFree(e->wg_scratch);
  if (e->wg_jobs_dev) hipFree(e->wg_jobs_dev);
  if (e->wg_stream) hipStreamSynchronize(e->wg_stream);      // (the stream belongs to the process' pool: aux_stream)
  if (e->wg_ev) hipEventDestroy(e->wg_ev);
  if (e->wg_done) hipEventDestroy(e->wg_done);
  if (e->dg_stream) hipStreamSynchronize(e->dg_stream);
  if (e->dg_ev) hipEventDestroy(e->dg_ev);
  if (e->dg_done) hipEventDestroy(e->dg_done);
  if (e->axpy_jobs_dev) hipFree(e->axpy_jobs_dev);
  if (e->sum_jobs_dev) hipFree(e->sum_jobs_dev);
  if (e->rt.blob) hipFree(e->rt.blob);
  if (e->rt.wino) hipFree(e->rt.wino);
  for (auto& pr : e->prof_events) { hipEventDestroy(pr.e0); hipEventDestroy(pr.e1); }
  delete e;
}

const char* hcf_last_error(const hcf_engine* e) { return e ? e->err.c_str() : "null engine"; }

int hcf_aux_stream(int32_t device, int32_t slot, hcf_stream_t* out) {
  if (!out || slot < 0 || slot > 1) return HCF_ERR_ARG;
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return HCF_ERR_HIP;
  if (device >= 0 && device != cur && hipSetDevice(device) != hipSuccess) return HCF_ERR_HIP;
  hipStream_t st = hcf::aux_stream(slot);
  if (device >= 0 && device != cur) hipSetDevice(cur);
  if (!st) return HCF_ERR_HIP;
  *out = (hcf_stream_t)st;
  return HCF_OK;
}

int hcf_param_count(const hcf_engine* e) { return e ? (int)e->specs.size() : HCF_ERR_ARG; }

int hcf_param_info(const hcf_engine* e, int index, const char** key, int32_t* ndim, int64_t shape[4]) {
  if (!e || index < 0 || index >= (int)e->specs.size()) return HCF_ERR_ARG;
  const Spec& s = e->specs[index];
  if (key) *key = s.key.c_str();
  if (ndim) *ndim = (int32_t)s.shape.size();
  if (shape)
    for (size_t i = 0; i < 4; ++i) shape[i] = i < s.shape.size() ? s.shape[i] : 1;
  return HCF_OK;
}

int hcf_set_param(hcf_engine* e, const char* key, const float* host_data, const int64_t* shape, int32_t ndim) {
  if (!e || !key || !host_data || !shape || ndim < 1 || ndim > 4) return HCF_ERR_ARG;
  auto it = e->params.find(key);
  if (it == e->params.end()) return e->fail(HCF_ERR_KEY, std::string("unexpected key in state_dict: ") + key), HCF_ERR_KEY;
  HostTensor& t = it->second;
  if ((int)t.shape.size() != ndim) return e->fail(HCF_ERR_SHAPE, std::string("size mismatch for ") + key), HCF_ERR_SHAPE;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (t.shape[i] != shape[i]) return e->fail(HCF_ERR_SHAPE, std::string("size mismatch for ") + key), HCF_ERR_SHAPE;
    n *= (size_t)shape[i];
  }
  t.data.assign(host_data, host_data + n);
  t.set = true;
  e->finalized = false;
  return HCF_OK;
}

int hcf_finalize(hcf_engine* e, int device) {
  if (!e) return HCF_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return e->fail(HCF_ERR_HIP, "hipSetDevice failed (no GPU?)");
  if (e->device >= 0 && e->device != device && e->arena.base) {
    hipSetDevice(e->device);
    hipFree(e->arena.base);
    e->arena.base = nullptr;
    e->arena.cap = 0;
    hipSetDevice(device);
  }
  hipDeviceSynchronize();      // packed weights of a previous finalize may still be in use
  e->free_weights();
  e->train_ready = false;
  e->cc_valid = false;
  e->invalidate_tapes();
  e->host_stale = false;
  e->wino_stale = false;       // build() re-packs the Winograd form from the host weights
  e->fat_stale = false;
  e->device = device;
  e->spec_mode = false;
  e->rc = HCF_OK;
  e->err.clear();
  const int r = e->build();
  if (r != HCF_OK) { e->free_weights(); return r; }
  e->finalized = true;
  return HCF_OK;
}

int hcf_inverse_ex(hcf_engine* e, const float* lr, const float* const* eps, int32_t n_eps, float tau, uint64_t seed,
                   int64_t first_sample, float* out_hr, int32_t B, int32_t h, int32_t w, uint32_t flags, hcf_stream_t stream) {
  if (!e || !lr || !out_hr || B < 1 || h < 1 || w < 1 || first_sample < 0) return HCF_ERR_ARG;
  return e->run_pass([&]() { e->pass_inverse(lr, eps, n_eps, tau, seed, first_sample, out_hr, B, h, w, flags); },
                     (hipStream_t)stream, flags, {1, B, h, w});
}

int hcf_inverse(hcf_engine* e, const float* lr, const float* const* eps, int32_t n_eps, float tau, uint64_t seed,
                float* out_hr, int32_t B, int32_t h, int32_t w, uint32_t flags, hcf_stream_t stream) {
  return hcf_inverse_ex(e, lr, eps, n_eps, tau, seed, 0, out_hr, B, h, w, flags, stream);
}

int hcf_check_range(hcf_engine* e, int32_t* overflowed) {
  if (!e) return HCF_ERR_ARG;
  int o = 0;
  const int r = e->check_range(&o);
  if (overflowed) *overflowed = o;
  return r;
}

int hcf_check_range_samples(hcf_engine* e, int32_t* overflowed, uint32_t* sample_slots) {
  if (!e) return HCF_ERR_ARG;
  int o = 0;
  uint32_t m = 0;
  const int r = e->check_range(&o, &m);
  if (overflowed) *overflowed = o;
  if (sample_slots) *sample_slots = m;
  return r;
}

int hcf_forward_sr(hcf_engine* e, const float* hr, const float* lr, const float* noise, float* out_lr, float* out_nll,
                   float* out_logdet, float* out_z, int32_t B, int32_t H, int32_t W, hcf_stream_t stream) {
  if (!e || !hr || B < 1 || H < 1 || W < 1) return HCF_ERR_ARG;
  if (e->cfg.kind != HCF_KIND_SR) return e->fail(HCF_ERR_STATE, "hcf_forward_sr on a rescaling engine"), HCF_ERR_STATE;
  const int m = 1 << e->cfg.L;
  if (H % m || W % m) return e->fail(HCF_ERR_SHAPE, "H, W must be divisible by the scale (squeeze2d assert, Basic.py:136)"), HCF_ERR_SHAPE;
  return e->run_pass([&]() { e->cc_valid = false; e->pass_forward(hr, lr, noise, out_lr, out_nll, out_logdet, out_z, nullptr, nullptr, B, H, W, 0); },
                     (hipStream_t)stream, 0, {2, B, H, W, lr ? 1 : 0, noise ? 1 : 0, out_z ? 1 : 0});
}

int hcf_forward_rescale(hcf_engine* e, const float* hr, float* out_lr, float* out_z1, float* out_z2, int32_t B, int32_t H,
                        int32_t W, uint32_t flags, hcf_stream_t stream) {
  if (!e || !hr || !out_lr || B < 1 || H < 1 || W < 1) return HCF_ERR_ARG;
  if (e->cfg.kind != HCF_KIND_RESCALING) return e->fail(HCF_ERR_STATE, "hcf_forward_rescale on an SR engine"), HCF_ERR_STATE;
  const int m = 1 << e->cfg.L;
  if (H % m || W % m) return e->fail(HCF_ERR_SHAPE, "H, W must be divisible by 4"), HCF_ERR_SHAPE;
  return e->run_pass([&]() { e->cc_valid = false; e->pass_forward(hr, nullptr, nullptr, out_lr, nullptr, nullptr, nullptr, out_z1, out_z2, B, H, W, flags); },
                     (hipStream_t)stream, flags, {3, B, H, W});
}

int hcf_set_precision(hcf_engine* e, int32_t mode) {
  if (!e || (mode != PREC_EXACT && mode != PREC_F16X3)) return HCF_ERR_ARG;
  e->precision = mode;
  return HCF_OK;
}

int hcf_get_precision(const hcf_engine* e) { return e ? e->precision : HCF_ERR_ARG; }

int64_t hcf_fallback_count(const hcf_engine* e) { return e ? e->n_fallbacks : -1; }

size_t hcf_workspace_bytes(const hcf_engine* e) {      // inference arena + the training tapes' activation / gradient arenas
  if (!e) return 0;
  size_t n = e->arena.cap + e->garena.cap;
  for (const auto& t : e->slots) n += t.a.cap + t.g.cap;
  return n;
}
size_t hcf_weight_bytes(const hcf_engine* e) { return e ? e->weight_bytes : 0; }

int hcf_train_forward_sr(hcf_engine* e, const float* hr, const float* lr, const float* noise, float* out_lr,
                         float* out_nll, float* out_logdet, int32_t B, int32_t H, int32_t W, hcf_stream_t stream) {
  if (!e || !hr || !lr || !noise || !out_lr || !out_nll || !out_logdet || B < 1 || H < 1 || W < 1) return HCF_ERR_ARG;
  const int m = 1 << e->cfg.L;
  if (H % m || W % m) return e->fail(HCF_ERR_SHAPE, "H, W must be divisible by the scale (squeeze2d assert, Basic.py:136)");
  return e->run_train_forward(hr, lr, noise, out_lr, out_nll, out_logdet, B, H, W, (hipStream_t)stream);
}

int hcf_train_backward(hcf_engine* e, float grad_nll, float* dparams, int64_t numel, hcf_stream_t stream) {
  if (!e || numel < 0) return HCF_ERR_ARG;
  hcf_engine::BwdIn in = {1, grad_nll, nullptr, nullptr, nullptr, nullptr};
  return e->run_backward(in, dparams, (size_t)numel, (hipStream_t)stream);
}

int hcf_train_backward_phase(hcf_engine* e, int32_t phase, float grad_nll, float* dparams, int64_t numel, hcf_stream_t stream) {
  if (!e || numel < 0 || phase < 0 || phase > 1) return HCF_ERR_ARG;
  hcf_engine::BwdIn in = {1, grad_nll, nullptr, nullptr, nullptr, nullptr};
  return e->run_backward(in, dparams, (size_t)numel, (hipStream_t)stream, phase);
}

int hcf_train_select_tape(hcf_engine* e, int32_t slot) {
  if (!e || slot < 0 || slot > 1) return HCF_ERR_ARG;
  e->cur_slot = slot;
  return HCF_OK;
}

int hcf_train_forward_rescale(hcf_engine* e, const float* hr, float* out_lr, float* out_z1, float* out_z2, int32_t B,
                              int32_t H, int32_t W, uint32_t flags, hcf_stream_t stream) {
  if (!e || !hr || !out_lr || !out_z1 || !out_z2 || B < 1 || H < 1 || W < 1) return HCF_ERR_ARG;
  if (H % 4 || W % 4) return e->fail(HCF_ERR_SHAPE, "H, W must be divisible by 4");
  return e->run_train_forward_rescale(hr, out_lr, out_z1, out_z2, B, H, W, flags, (hipStream_t)stream);
}

int hcf_train_backward_rescale(hcf_engine* e, const float* g_lr, const float* g_z1, const float* g_z2, float* dparams,
                               int64_t numel, hcf_stream_t stream) {
  if (!e || numel < 0) return HCF_ERR_ARG;
  hcf_engine::BwdIn in = {3, 0.f, g_lr, nullptr, g_z1, g_z2};
  return e->run_backward(in, dparams, (size_t)numel, (hipStream_t)stream);
}

int hcf_train_inverse(hcf_engine* e, const float* lr, const float* const* eps, int32_t n_eps, float tau, uint64_t seed,
                      float* out_hr, int32_t B, int32_t h, int32_t w, uint32_t flags, hcf_stream_t stream) {
  if (!e || !lr || !out_hr || B < 1 || h < 1 || w < 1) return HCF_ERR_ARG;
  return e->run_train_inverse(lr, eps, n_eps, tau, seed, out_hr, B, h, w, flags, (hipStream_t)stream);
}

int hcf_train_backward_inverse(hcf_engine* e, const float* grad_out, float* dparams, int64_t numel, float* grad_lr,
                               hcf_stream_t stream) {
  if (!e || numel < 0) return HCF_ERR_ARG;
  hcf_engine::BwdIn in = {2, 1.f, grad_out, grad_lr, nullptr, nullptr};
  return e->run_backward(in, dparams, (size_t)numel, (hipStream_t)stream);
}

int hcf_bind_param_device(hcf_engine* e, const char* key, const float* dev_ptr) {
  if (!e || !key) return HCF_ERR_ARG;
  if (!e->params.count(key)) return e->fail(HCF_ERR_KEY, std::string("unknown parameter: ") + key);
  if (e->dev_src[key] != dev_ptr) ++e->refresh_gen;      // the cached refresh job tables hold this pointer
  e->dev_src[key] = dev_ptr;
  return HCF_OK;
}

int hcf_refresh_from_device(hcf_engine* e, hcf_stream_t stream) {
  if (!e) return HCF_ERR_ARG;
  if (e->ensure_train_ready() != HCF_OK) return e->rc;     // the transposed packs exist before the first refresh
  return e->refresh_from_device((hipStream_t)stream);
}

int hcf_actnorm_init_request(hcf_engine* e, const char* const* prefixes, int32_t n) {
  if (!e || n < 0 || (n > 0 && !prefixes)) return HCF_ERR_ARG;
  if (!e->finalized) return e->fail(HCF_ERR_STATE, "hcf_finalize() has not been called");
  e->an_pending.clear();
  e->an_fitted.clear();
  for (int i = 0; i < n; ++i) {
    if (!prefixes[i]) return HCF_ERR_ARG;
    const std::string k(prefixes[i]);
    if (!e->params.count(k + ".bias") || !e->params.count(k + ".logs")) return e->fail(HCF_ERR_KEY, "not an ActNorm: " + k);
    if (e->host_stale) {        // the "is the bias still zero" rule (ActNorms.py:33-35) must see the current values
      for (const char* suf : {".bias", ".logs"}) {
        auto ds = e->dev_src.find(k + suf);
        std::vector<float>& h = e->params[k + suf].data;
        if (ds != e->dev_src.end() && ds->second &&
            hipMemcpy(h.data(), ds->second, sizeof(float) * h.size(), hipMemcpyDeviceToHost) != hipSuccess)
          return e->fail(HCF_ERR_HIP, "hcf_actnorm_init_request: D2H copy failed");
      }
    }
    e->an_pending.insert(k);
  }
  e->an_active = n > 0;
  return HCF_OK;
}

int hcf_get_param(hcf_engine* e, const char* key, float* out, int64_t numel) {
  if (!e || !key || !out) return HCF_ERR_ARG;
  auto it = e->params.find(key);
  if (it == e->params.end() || !it->second.set) return e->fail(HCF_ERR_KEY, std::string("unknown or unset parameter: ") + key);
  if ((int64_t)it->second.data.size() != numel) return e->fail(HCF_ERR_SHAPE, std::string("size mismatch for parameter: ") + key);
  const std::string skey(key);
  const size_t dot = skey.rfind('.');
  const bool fitted = dot != std::string::npos && e->an_fitted.count(skey.substr(0, dot)) > 0;
  if (e->host_stale && !fitted) {          // the authoritative copy is the caller's device tensor
    auto ds = e->dev_src.find(key);
    if (ds != e->dev_src.end() && ds->second &&
        hipMemcpy(it->second.data.data(), ds->second, sizeof(float) * (size_t)numel, hipMemcpyDeviceToHost) != hipSuccess)
      return e->fail(HCF_ERR_HIP, "hcf_get_param: D2H copy failed");
  }
  memcpy(out, it->second.data.data(), sizeof(float) * (size_t)numel);
  return HCF_OK;
}

int hcf_debug_range_probe(hcf_engine* e, int32_t enable) {
  if (!e) return HCF_ERR_ARG;
  if (enable) {
    if (e->device >= 0 && hipSetDevice(e->device) != hipSuccess) return e->fail(HCF_ERR_HIP, "hipSetDevice failed");
    if (!e->probe_dev && hipMalloc((void**)&e->probe_dev, sizeof(float) * 2 * hcf_engine::kProbeCap) != hipSuccess)
      return e->fail(HCF_ERR_NOMEM, "hipMalloc failed for the range probe");
    if (hipMemset(e->probe_dev, 0, sizeof(float) * 2 * hcf_engine::kProbeCap) != hipSuccess) return e->fail(HCF_ERR_HIP, "hipMemset failed");
    e->probe_recs.clear();
  }
  e->probe_on = enable != 0;
  return HCF_OK;
}

int hcf_debug_range_probe_read(hcf_engine* e, int32_t index, char* key, int32_t key_cap, float* maxima, int32_t* info) {
  if (!e || index < 0 || !key || key_cap < 1 || !maxima || !info) return HCF_ERR_ARG;
  if (index >= (int)e->probe_recs.size()) return HCF_ERR_KEY;              /* past the last record */
  if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(maxima, e->probe_dev + 2 * (size_t)index, 2 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    return e->fail(HCF_ERR_HIP, "range probe read-back failed");
  const auto& r = e->probe_recs[index];
  snprintf(key, (size_t)key_cap, "%s", r.key.c_str());
  info[0] = r.cin; info[1] = r.cout; info[2] = r.H; info[3] = r.W; info[4] = r.f16; info[5] = r.wino;
  return HCF_OK;
}

int hcf_profile_convs(hcf_engine* e, int enable) {
  if (!e) return HCF_ERR_ARG;
  if (enable && !e->prof) { e->prof_used = 0; e->prof_last_seq = ~0ull; }     // records survive a disable so they can be read afterwards
  e->prof = enable != 0;
  return HCF_OK;
}

int hcf_conv_time_ms(hcf_engine* e, int32_t taps, int32_t nt, int32_t kind, int32_t reset, double* total_ms,
                     int64_t* launches, double* flops, double* bytes) {
  if (!e) return HCF_ERR_ARG;
  double tot = 0, fl = 0, by = 0;
  int64_t n = 0;
  for (size_t i = 0; i < e->prof_used; ++i) {
    const auto& r = e->prof_events[i];
    if ((taps && r.taps != taps) || (nt && r.nt != nt) || (kind >= 0 && r.kind != kind)) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) return HCF_ERR_HIP;
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.chained ? e->prof_events[i - 1].e1 : r.e0, r.e1) != hipSuccess) return HCF_ERR_HIP;
    tot += ms;
    fl += r.flops;
    by += r.bytes;
    n++;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  if (reset) { e->prof_used = 0; e->prof_last_seq = ~0ull; }
  return HCF_OK;
}

}  // extern "C"
