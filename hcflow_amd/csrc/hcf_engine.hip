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
struct Rdb { Conv c[5]; Conv ca[2], cb[2]; bool fat[2] = {false, false}; Conv::TPack gt[5]; bool gather = false;
             float* gtw[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; };      // gtw: the gather packs in Winograd form (hcf_engine_train.inc)
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

#include "hcf_engine_build.inc"      // planning: packs, derived packs, the module-tree walk
#include "hcf_engine_run.inc"        // routing + execution: run_conv, blocks, pass_inverse / pass_forward
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
