/* hcflow.h -- C ABI of the MI355X-native HCFlow forward / inverse engine (libhcflow_hip.so).
 *
 * The reference (JingyunLiang/HCFlow) is pure PyTorch: its drop-in boundary is the Python class
 * looked up by codes/models/networks.py:36-41 (define_G -> HCFlowNet_SR / HCFlowNet_Rescaling with
 * forward(hr, lr, z, u, eps_std, add_gt_noise, step, reverse, training),
 * codes/models/modules/HCFlowNet_SR_arch.py:34-42, HCFlowNet_Rescaling_arch.py:26-36). There is no
 * FFI in the reference, so this C ABI sits directly beneath that class: hcflow_amd/arch.py keeps
 * the reference's module surface and state_dict and forwards every call to the entry points
 * below (last row). Each entry point names the reference code it replaces.
 *
 * Conventions: plain pointers and sizes, no torch types. Return 0 on success, a negative HCF_ERR_*
 * code otherwise (never throws); hcf_last_error() gives a message. Image tensors are dense NCHW
 * fp32 DEVICE pointers owned by the caller; parameters are passed as HOST pointers and packed /
 * uploaded by hcf_finalize(). All work is enqueued on the caller's HIP stream; hcf_inverse / hcf_forward_sr /
 * hcf_forward_rescale never synchronise with the device once the workspace has reached its steady-state size (the first call
 * per shape sizes and may allocate it): a steady-state call walks the layer graph once and returns, so it can be captured into
 * a HIP graph. The f16x3 range flag is read back asynchronously; hcf_check_range() is the call that waits for it.
 */
#ifndef HCFLOW_H_
#define HCFLOW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HCF_OK 0
#define HCF_ERR_ARG (-1)
#define HCF_ERR_HIP (-2)
#define HCF_ERR_STATE (-3)
#define HCF_ERR_KEY (-4)
#define HCF_ERR_SHAPE (-5)
#define HCF_ERR_UNSUPPORTED (-6)
#define HCF_ERR_NOMEM (-7)

/* enum values used in hcf_config */
#define HCF_KIND_SR 0
#define HCF_KIND_RESCALING 1
#define HCF_SQUEEZE_CHECKERBOARD 0
#define HCF_SQUEEZE_HAAR 1
#define HCF_PERM_INVCONV 0
#define HCF_PERM_NONE 1
#define HCF_COUPLING_AFFINE 0
#define HCF_COUPLING_AFFINE3SHIFT 1
#define HCF_NN_FCN 0
#define HCF_NN_DENSEBLOCK 1

/* flags for hcf_inverse / hcf_forward_* */
#define HCF_FLAG_NO_CLAMP 1u        /* return the flow output before torch.clamp(.,0,1) (parity tests) */
#define HCF_FLAG_NO_RANGE_CHECK 2u  /* f16x3 mode: do not even enqueue the read-back of the range flag (hcf_check_range) */
#define HCF_FLAG_KEEP_COND 4u       /* hcf_inverse: compute the deepest level's conditional features + prior head and keep them */
#define HCF_FLAG_REUSE_COND 8u      /* hcf_inverse: the caller asserts that lr (contents, B, h, w) is the one of the previous
                                     * KEEP / REUSE call: the kept features are used if still valid (else recomputed and kept).
                                     * They depend on lr only (FlowNet_SR_x4.py:113-115, FlowNet_SR_x8.py:129), so a tau sweep or
                                     * repeated sampling of one LR batch skips that level's RRDB trunks; results are bit-identical. */

/* numerics of the convolutions (hcf_set_precision) */
#define HCF_PRECISION_EXACT 0       /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32) */
#define HCF_PRECISION_F16X3 1       /* fp32-equivalent: 3 f16 MFMAs per product block, fp32 accumulate */

typedef void* hcf_stream_t;        /* a hipStream_t */
typedef struct hcf_engine hcf_engine;

/* The yml network_G block as the reference reads it in FlowNet.__init__
 * (FlowNet_SR_x4.py:17-27, FlowNet_SR_x8.py, FlowNet_Rescaling_x4.py:15-29) and
 * ConditionalFlow.__init__ (ConditionalFlow.py:15-41). */
typedef struct hcf_config {
  int32_t kind;             /* HCF_KIND_* : HCFlowNet_SR or HCFlowNet_Rescaling */
  int32_t scale;            /* 4 or 8 (= 2^L) */
  int32_t in_nc;            /* network_G.in_nc (3) */
  float quant;              /* opt.quant (SR) / datasets.train.quant (rescaling) */
  int32_t L;                /* flowDownsampler.L : 2 or 3 */
  int32_t K[4];             /* flowDownsampler.K per level */
  int32_t after[4];         /* splitOff.after_flowstep per level */
  int32_t squeeze;          /* HCF_SQUEEZE_* */
  int32_t perm, coupling, nn_module, hidden;          /* main flow steps */
  int32_t c_perm, c_coupling, c_nn_module, c_hidden;  /* conditional (splitOff) flow steps */
  int32_t rrdb_nb[2];
  int32_t rrdb_nf, rrdb_gc;
  int32_t lu_decomposed;    /* FlowStep(LU_decomposed=...) (FlowStep.py:9-10,20): every invertible 1x1 conv holds the factors
                             * l, log_s, u (parameters) and p, sign_s (buffers) of W = P (L o mask + I) (U o mask^T + diag(sign_s
                             * exp(log_s))) instead of `weight` (Permutations.py:41-57,78-92); dlogdet = sum(log_s) * pixels */
} hcf_config;

/* ---- life cycle -------------------------------------------------------------------------- */
/* Replaces HCFlowNet_SR.__init__ / HCFlowNet_Rescaling.__init__ -> FlowNet.__init__
 * (HCFlowNet_SR_arch.py:12-31). Touches no device. */
int hcf_create(const hcf_config* cfg, hcf_engine** out);
void hcf_destroy(hcf_engine* e);
const char* hcf_last_error(const hcf_engine* e);

/* state_dict introspection: the keys / shapes the reference modules register, in registration
 * order (SURVEY.md 8b "State-dict (strict)"). shape has up to 4 entries. */
int hcf_param_count(const hcf_engine* e);
int hcf_param_info(const hcf_engine* e, int index, const char** key, int32_t* ndim, int64_t shape[4]);

/* Replaces nn.Module.load_state_dict(strict=True) as used by BaseModel.load_network
 * (codes/models/base_model.py:96-120): one call per tensor with a HOST fp32 pointer. */
int hcf_set_param(hcf_engine* e, const char* key, const float* host_data, const int64_t* shape, int32_t ndim);

/* Pack weights for the kernels (implicit-GEMM layout, ActNorm folded into conv epilogues, W^-1 in
 * fp64 as Permutations.py:74 does, slogdet hoisted out of the pass) and upload to `device`.
 * Must be called after the last hcf_set_param and again whenever parameters change. */
int hcf_finalize(hcf_engine* e, int device);

/* ---- the hot path -------------------------------------------------------------------------- */
/* netG(lr=lr, z=None, u=None, eps_std=tau, reverse=True): HCFlowNet_SR.reverse_flow_diracLR
 * (HCFlowNet_SR_arch.py:70-75) and HCFlowNet_Rescaling.reverse_flow_diracLR
 * (HCFlowNet_Rescaling_arch.py:49-54) -> FlowNet.reverse_flow (FlowNet_SR_x4.py:106-123).
 *   lr      [B,3,h,w]  device
 *   eps     n_eps device pointers, sampling order (deepest level first), each [B,C_l,h_l,w_l] and
 *           ALREADY N(0,tau) distributed (what GaussianDiag.sample draws, Basic.py:98-99); an
 *           entry (or the whole array) may be NULL -> drawn on device (Philox, `seed`) * tau
 *   out_hr  [B,3,h*scale,w*scale] device */
int hcf_inverse(hcf_engine* e, const float* lr, const float* const* eps, int32_t n_eps, float tau, uint64_t seed,
                float* out_hr, int32_t B, int32_t h, int32_t w, uint32_t flags, hcf_stream_t stream);
/* The same for a SHARD of a larger batch (batch-sharded multi-GPU sampling): sample b of this call is sample
 * first_sample + b of the batch, and the device Philox draws are those of that global sample -- N shards with the same seed
 * draw exactly the eps of the one-GPU batch (the reference seeds every rank identically, train_HCFlow.py:43-46, which would give
 * every shard the same eps). The IMAGES are bit-identical when the shards take the same kernel schedule as the full batch
 * (always for equal per-sample sizes at the BASELINE shapes; a layer's Winograd / direct choice follows how many units a launch
 * has, so very small or very large shards agree to fp32 rounding instead). hcf_inverse = first_sample 0. */
int hcf_inverse_ex(hcf_engine* e, const float* lr, const float* const* eps, int32_t n_eps, float tau, uint64_t seed,
                   int64_t first_sample, float* out_hr, int32_t B, int32_t h, int32_t w, uint32_t flags, hcf_stream_t stream);

/* netG(hr=hr, lr=lr, reverse=False) for HCFlowNet_SR: normal_flow_diracLR (HCFlowNet_SR_arch.py:47-67).
 *   hr [B,3,H,W], lr [B,3,H/scale,W/scale] (NULL -> no Dirac term), noise [B,3,H,W] U[0,1) or NULL
 *   (NULL -> no dequantisation noise is added)
 *   out_lr [B,3,H/scale,W/scale] = clamp(Quant(z)); out_nll [1]; out_logdet [B] objective per sample
 *   (nullable); out_z [B,3,H/scale,W/scale] pre-quantisation latent (nullable) */
int hcf_forward_sr(hcf_engine* e, const float* hr, const float* lr, const float* noise, float* out_lr, float* out_nll,
                   float* out_logdet, float* out_z, int32_t B, int32_t H, int32_t W, hcf_stream_t stream);

/* netG(hr=hr, reverse=False) for HCFlowNet_Rescaling: normal_flow_diracLR
 * (HCFlowNet_Rescaling_arch.py:39-46): out_lr = clamp(LR^), out_z1 [B,6,H/2,W/2], out_z2 [B,21,H/4,W/4] */
int hcf_forward_rescale(hcf_engine* e, const float* hr, float* out_lr, float* out_z1, float* out_z2, int32_t B, int32_t H,
                        int32_t W, uint32_t flags, hcf_stream_t stream);

/* Convolution numerics. HCF_PRECISION_EXACT (default): fp32 MFMA, an exact fp32 fma chain.
 * HCF_PRECISION_F16X3: every fp32 product is formed from f16 hi/lo parts (a_hi b_hi + a_hi b_lo +
 * a_lo b_hi, error ~2^-22) on the 16x faster f16 matrix cores with fp32 accumulation; deviation
 * from fp64 on the full nets equals plain fp32's. An input beyond the f16 range (|x| >= 65504) cannot be
 * split: it raises a sticky device flag whose value every f16x3 inference pass copies to pinned host memory asynchronously.
 * hcf_check_range waits for the passes enqueued so far and reports (and clears) it: *overflowed = 1 means the outputs of the
 * f16x3 passes since the last check are invalid and must be recomputed with HCF_PRECISION_EXACT (hcflow_amd/arch.py does that
 * by default after every pass; `set_range_check("lazy")` defers it). hcf_fallback_count = number of raised checks. The taped
 * training passes (hcf_train_*) check and re-run by themselves (they synchronise anyway). */
int hcf_set_precision(hcf_engine* e, int32_t mode);
int hcf_get_precision(const hcf_engine* e);
int hcf_check_range(hcf_engine* e, int32_t* overflowed);
/* The same check with the affected SAMPLES: every op of the path is per-sample (HCFlowNet_SR_arch.py:70-75, thops.sum(dim=[1,2,3])),
 * so an out-of-range activation invalidates only the samples whose tiles saw it. *sample_slots: bit (b mod 30) is set for every
 * sample index b (within its call's batch) that was flagged since the last check -- a superset for B > 30 (slots alias); all 30
 * bits when a flagging kernel could not name its sample (device flag: bit 0 = overflow, bits 1..30 = sample slots, bit 31 =
 * "unattributed", latched on its own so that slots named by another kernel or pass cannot mask it). The caller re-runs only those samples with HCF_PRECISION_EXACT
 * (hcflow_amd/arch.py: one B = 1 pass per flagged sample through hcf_inverse_ex with its sample offset) instead of the whole
 * batch. Reports and clears like hcf_check_range (one fallback counted). */
int hcf_check_range_samples(hcf_engine* e, int32_t* overflowed, uint32_t* sample_slots);
int64_t hcf_fallback_count(const hcf_engine* e);

/* Side stream `slot` (0 / 1) of `device` (-1: the current one): the process holds at most two per device (low priority,
 * non-blocking, created on first use, never destroyed) and every engine uses them -- the training pass for its weight-gradient and
 * conditional-feature-gradient work, the module for the two half batches of a split inference call (hcflow_amd/arch.py:
 * set_streams(2)). Replaces nothing in the reference (it runs on one stream; codes/models/HCFlow_SR_model.py:184-205, 209-262);
 * exported so that the host side does not create further streams of its own: HIP spreads streams over four hardware queues. */
int hcf_aux_stream(int32_t device, int32_t slot, hcf_stream_t* out);

/* bytes of device workspace currently held (activations arena) and of packed weights */
size_t hcf_workspace_bytes(const hcf_engine* e);
size_t hcf_weight_bytes(const hcf_engine* e);

/* ---- NLL training step (reference: HCFlow_SR_model.optimize_parameters, HCFlow_SR_model.py:195-202:
 *      `_, nll = netG(hr, lr, reverse=False); nll.backward()`) -------------------------------------------------
 * hcf_train_forward_sr = hcf_forward_sr (same outputs; hr / lr / noise are required) that additionally keeps every
 * intermediate tensor in HBM and records the backward pass. hcf_train_backward then writes
 *   d(grad_nll * nll) / d(parameter)   for EVERY parameter, concatenated in hcf_param_info order (= state_dict
 * order, each tensor flattened), into `dparams` (device buffer of `numel` = total parameter count floats); the
 * caller slices it into its .grad tensors. One backward per forward; `lr` must stay alive in between. The convs follow
 * hcf_set_precision (f16x3: forward, data-gradient and weight-gradient convs on the split kernels, the latter only for
 * layers whose taped forward was range-checked; exact: fp32 MFMA). Every per-channel and split-K sum is reduced in a fixed
 * order: the gradients are bit-reproducible run to run (no floating-point atomics). SR nets only. */
int hcf_train_forward_sr(hcf_engine* e, const float* hr, const float* lr, const float* noise, float* out_lr,
                         float* out_nll, float* out_logdet, int32_t B, int32_t H, int32_t W, hcf_stream_t stream);
int hcf_train_backward(hcf_engine* e, float grad_nll, float* dparams, int64_t numel, hcf_stream_t stream);
/* The same backward pass in two calls, for a caller whose gradient all-reduce overlaps the backward pass (the reference trains under
 * DistributedDataParallel, HCFlow_SR_model.py:33-36, whose buckets are reduced as their gradients arrive). phase 0 runs the part
 * of the pass that was taped last -- the output terms and the level-0 conditional flow -- and completes every parameter-gradient
 * reduction it enqueued on `stream`: when it returns (stream order), the slices of `dparams` of all parameters whose key starts with
 * "flow.level0_condFlow." (about half of an SR x4 net) are FINAL and nothing later touches them. phase 1 (same `dparams`) runs the
 * rest. The two calls together write exactly what hcf_train_backward writes, bit for bit. */
int hcf_train_backward_phase(hcf_engine* e, int32_t phase, float grad_nll, float* dparams, int64_t numel, hcf_stream_t stream);

/* Gradients through the REVERSE (sampling) path (reference: the HR pixel / feature / GAN losses of the HCFlow+ / ++
 * recipes, HCFlow_SR_model.py:207-255: `fake_H = netG(lr=, eps_std=, reverse=True)` followed by a loss on fake_H and
 * backward()). hcf_train_inverse = hcf_inverse (same arguments and output) with a tape; hcf_train_backward_inverse
 * takes dL/d(out_hr) (device [B,3,H,W], the clamp's gradient mask is applied inside) and writes dL/d(parameter) for
 * every parameter like hcf_train_backward. */
int hcf_train_inverse(hcf_engine* e, const float* lr, const float* const* eps, int32_t n_eps, float tau, uint64_t seed,
                      float* out_hr, int32_t B, int32_t h, int32_t w, uint32_t flags, hcf_stream_t stream);
int hcf_train_backward_inverse(hcf_engine* e, const float* grad_out, float* dparams, int64_t numel, float* grad_lr,
                               hcf_stream_t stream);     /* grad_lr: optional device [B,3,h,w], receives dL/d lr */

/* Rescaling net (reference: one generator step of HCFlow_Rescaling_model.optimize_parameters, :212-256 --
 * `fake_LR, z1, z2 = netG(hr, reverse=False)`; losses on all three; `fake_H = netG(lr=Quant(fake_LR), reverse=True)`;
 * loss on fake_H; ONE backward through both passes). hcf_train_forward_rescale = hcf_forward_rescale with a tape;
 * hcf_train_backward_rescale takes dL/d(out_lr), dL/d z1, dL/d z2 (device, each nullable; the clamp mask of out_lr is
 * applied inside). Both passes of the step must keep their tapes alive at once: hcf_train_select_tape picks the slot
 * (0 or 1) that the following taped passes / backward calls use. With hcf_train_backward_inverse's grad_lr the
 * gradient flows from the inverse pass into the forward pass (through the caller's straight-through Quant). */
int hcf_train_select_tape(hcf_engine* e, int32_t slot);
int hcf_train_forward_rescale(hcf_engine* e, const float* hr, float* out_lr, float* out_z1, float* out_z2, int32_t B,
                              int32_t H, int32_t W, uint32_t flags, hcf_stream_t stream);
int hcf_train_backward_rescale(hcf_engine* e, const float* g_lr, const float* g_z1, const float* g_z2, float* dparams,
                               int64_t numel, hcf_stream_t stream);

/* In-place parameter updates on the GPU (optimiser steps; reference: base_model / HCFlow_SR_model keep netG's
 * parameters on the device and Adam updates them in place, HCFlow_SR_model.py:108-125,202). After hcf_finalize,
 * hcf_bind_param_device tells the engine where each parameter lives in DEVICE memory (fp32, contiguous, same shape
 * as hcf_param_info reports; the pointer must stay valid); hcf_refresh_from_device then rewrites every weight pack
 * and table from those tensors with device kernels (host work only for the invertible 1x1 convs' fp64 inverse and
 * log-determinant: one small D2H + H2D round trip) -- milliseconds instead of the host repack of hcf_finalize. */
int hcf_bind_param_device(hcf_engine* e, const char* key, const float* dev_ptr);
int hcf_refresh_from_device(hcf_engine* e, hcf_stream_t stream);

/* ActNorm data-dependent initialisation (reference: _ActNorm.initialize_parameters, ActNorms.py:29-43, reached from
 * _ActNorm.forward when `not self.inited` in train() mode, :78-80). hcf_actnorm_init_request() arms the NEXT
 * hcf_forward_sr / hcf_forward_rescale call: each listed ActNorm (state_dict prefix, e.g.
 * "flow.layers.1.actnorm" or "flow.layers.1.affine.f.conv1.actnorm") whose stored bias is all zero gets
 * bias = -mean, logs = log(1 / (sqrt(var) + 1e-6)) of the tensor that reaches it, per channel over (B, H, W), in
 * forward order, and the pass continues with the fitted values (the pass runs on the exact fp32 kernels and
 * synchronises the stream once per fitted layer). Afterwards hcf_get_param() returns the fitted tensors (HOST
 * buffer of `numel` floats; works for any parameter key) so that the caller can store them in its own module. */
int hcf_actnorm_init_request(hcf_engine* e, const char* const* prefixes, int32_t n);
int hcf_get_param(hcf_engine* e, const char* key, float* host_out, int64_t numel);

/* timing hook used by bench.py: when enabled, every conv launch is bracketed by HIP events on the
 * launch stream. hcf_conv_time_ms() sums the recorded launches of one kernel variant
 * (taps in {9,1} = 3x3 / 1x1, nt = N tiles of 32 output channels; 0 = any; kind = 0 plain conv, 1 with the fused
 * 1x1 second layer, 2 with the fused flow-step tail, 3 reading an upsampled source, 4 the Winograd form of the f16x3
 * conv, 5 the persistent small-K FCN conv1 + conv2 kernel (hcf_conv_fcn.hip), 6 the Winograd form of a conditional FCN
 * conv1 with the 1x1 conv2 in its epilogue, 7 the 32 -> 32 completion of a fat dense-block launch (adds a stored partial), -1 any): total time, launch count,
 * algorithmic FLOPs (2 * taps * cin * cout per output pixel) and algorithmic HBM bytes (each source window,
 * residual and the weights read once, the output written once). reset != 0 clears the records. */
int hcf_profile_convs(hcf_engine* e, int enable);
int hcf_conv_time_ms(hcf_engine* e, int32_t taps, int32_t nt, int32_t kind, int32_t reset, double* total_ms,
                     int64_t* launches, double* flops, double* bytes);

/* ---- the optimiser step of the training caller (reference: HCFlow_SR_model.py:118-120 / HCFlow_Rescaling_model.py:140-142
 * build torch.optim.Adam(optim_params, lr, weight_decay, betas) over netG's ~1500 parameter tensors and step it once per
 * iteration, optimize_parameters :202) as ONE launch. param / exp_avg / exp_avg_sq: device fp32 buffers of one layout (every
 * parameter tensor in a slot at a multiple-of-4 offset; 16-byte aligned bases). chunks_dev: DEVICE table, one entry per
 * <= HCF_ADAM_CHUNK consecutive elements of a tensor: its gradient slice (device fp32, any alignment), the slot offset
 * (floats, multiple of 4) and the length. Arithmetic of torch.optim.Adam with amsgrad = False, maximize = False, L2 weight
 * decay: g += wd p; m += (g - m)(1 - beta1); v = v beta2 + (1 - beta2) g g; p -= lr / (1 - beta1^step) * m /
 * (sqrt(v) / sqrt(1 - beta2^step) + eps); step >= 1 is the count AFTER this update. Enqueues on `stream`, no sync. */
#define HCF_ADAM_CHUNK 4096
typedef struct hcf_adam_chunk {
  const float* grad;
  uint32_t offset;
  uint32_t n;
} hcf_adam_chunk;
int hcf_adam_step(float* param, float* exp_avg, float* exp_avg_sq, const hcf_adam_chunk* chunks_dev, int32_t n_chunks,
                  double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, hcf_stream_t stream);

/* ---- validation metrics around the path (reference: the per-image metric block of test_HCFlow.py:103-182) ------
 * hcf_metric_psnr_ssim: util.tensor2img (utils/util.py:790-816) on gt and sr, then util.calculate_psnr_ssim(gt, sr,
 * crop_border) (:898-982; Y channel as data/util.py:209-230) and, for scale > 1, the same on
 * imresize(gt, 1/scale), imresize(sr, 1/scale) with crop 0 (utils/imresize.py, MATLAB bicubic with antialiasing).
 * gt, sr: device [B,3,H,W] RGB fp32; out: HOST double [B][8] = psnr, ssim, psnr_y, ssim_y, then the four
 * down-scaled ("bicHR") values (0 when scale <= 1). float64 arithmetic. hcf_metric_imresize_down returns the
 * down-scaled tensor2img image itself (HOST double [B][3 (B,G,R)][ceil(H/scale)][ceil(W/scale)], 0..255 units). */
int hcf_metric_psnr_ssim(const float* gt, const float* sr, int32_t B, int32_t H, int32_t W, int32_t crop_border,
                         int32_t scale, double* out, hcf_stream_t stream);
int hcf_metric_imresize_down(const float* x, int32_t B, int32_t H, int32_t W, int32_t scale, double* out,
                             hcf_stream_t stream);

/* ---- per-op entry points (unit parity tests; tensors are device NCHW fp32) ------------------- */
/* F.conv2d(x, w, stride 1, padding k/2) with the fused epilogue
 *   y = res2 + rs2 * (res1 + rs1 * act((conv + bias) * scale))
 * w is a HOST pointer in PyTorch layout [cout, cin, k, k]; bias/scale HOST [cout] or NULL;
 * x is the channel-concatenation of n_src device tensors [B, src_c[i], H >> src_up[i], W >> src_up[i]]
 * (nearest-upsampled by 2^src_up[i]); res1/res2 device [B,cout,H,W] or NULL. k in {1,3}. */
int hcf_op_conv2d(const float* const* src, const int32_t* src_c, const int32_t* src_up, int32_t n_src, int32_t B,
                  int32_t H, int32_t W, const float* w, const float* bias, const float* scale, int32_t cout, int32_t k,
                  int32_t act, const float* res1, float rs1, const float* res2, float rs2, float* out,
                  hcf_stream_t stream);

/* Backward of hcf_op_conv2d without epilogue (torch.autograd of F.conv2d(cat(up(src_i)), w, bias, 1, k/2); the
 * reference reaches it through loss.backward() in HCFlow_SR_model.optimize_parameters, HCFlow_SR_model.py:195-202):
 * g = dL/dy, device [B,cout,H,W]. dsrc[i] (device [B, src_c[i], H >> up, W >> up]) may be NULL; dw is a HOST buffer
 * [cout, cin, k, k], dbias a HOST buffer [cout]; either may be NULL. Data gradients run on the forward conv
 * kernels with transposed / flipped weights (process-wide op precision). The weight gradient runs on
 * hcf_conv_wgrad.hip: fp32 MFMA in exact mode, the f16x3 matrix-core kernel (g scaled by a power of two from
 * max |g|, both operands split) in f16x3 mode; split-K partial tiles are added in a fixed order (deterministic). */
int hcf_op_conv2d_backward(const float* const* src, const int32_t* src_c, const int32_t* src_up, int32_t n_src,
                           int32_t B, int32_t H, int32_t W, const float* w, int32_t cout, int32_t k, const float* g,
                           float* const* dsrc, float* dw, float* dbias, hcf_stream_t stream);

/* Basic.squeeze2d / unsqueeze2d factor 2 (Basic.py:127-157) and HaarDownsampling (Basic.py:470-487) */
int hcf_op_squeeze2d(const float* x, float* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t haar, hcf_stream_t stream);
int hcf_op_unsqueeze2d(const float* x, float* out, int32_t B, int32_t C4, int32_t H, int32_t W, int32_t haar, hcf_stream_t stream);
/* FlowStep.reverse_flow tail / normal_flow head+coupling on given tensors (FlowStep.py:40-64):
 *   inverse: out = actnorm^-1( Winv @ coupling^-1(z, h) );  forward: zmid = W @ actnorm(z) then
 *   out = coupling(zmid, h), logdet[b] = sum logscale.  mat: HOST [C,C] (W for forward; the entry
 *   inverts it in fp64 for inverse) or NULL; bias/logs HOST [C]; mode/ns as hcf_common.h StepArgs. */
int hcf_op_step_inverse(const float* z, const float* h, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                        int32_t hC, int32_t mode, int32_t ns, const float* mat, const float* an_bias,
                        const float* an_logs, hcf_stream_t stream);
int hcf_op_step_forward_head(const float* z, float* out, int32_t B, int32_t C, int32_t H, int32_t W, const float* mat,
                             const float* an_bias, const float* an_logs, hcf_stream_t stream);
int hcf_op_step_forward_couple(const float* z, const float* h, float* out, float* logdet, int32_t B, int32_t C,
                               int32_t H, int32_t W, int32_t hC, int32_t mode, int32_t ns, hcf_stream_t stream);
/* GaussianDiag.logp / sample (Basic.py:78-101) with (mean, logs) = h[:,0::2], h[:,1::2] */
int hcf_op_gauss_logp(const float* h, const float* x, float* out_logp, int32_t B, int32_t C, int32_t H, int32_t W,
                      hcf_stream_t stream);
int hcf_op_gauss_sample(const float* h, const float* eps, float tau, uint64_t seed, float* out, int32_t B, int32_t C,
                        int32_t H, int32_t W, int32_t rescale, hcf_stream_t stream);

/* ---- auxiliary nets of the HCFlow+ / ++ recipes (reference: Discriminator_VGG_160 / VGGFeatureExtractor,
 *      codes/models/modules/discriminator_vgg_arch.py:68-157; used by HCFlow_SR_model.py:75-95,219-285) ------------------
 * Stride-1 "same" convolution (k = 3 or 1) and its gradients on DEVICE tensors with the flow's own conv / weight-gradient
 * kernels: x, y, g, dx are dense NHWC fp32 [B][H][W][cs] (cs % 4 == 0, 16-byte aligned), w is the PyTorch-layout weight
 * [cout][cin][k][k] in device memory (packs are rebuilt on the device per call), bias [cout] or NULL.
 *   hcf_aux_conv2d:          y[..., 0:cout] = act(conv(x[..., 0:cin], w) + bias), act 0 none / 1 relu / 2 leaky relu 0.2
 *   hcf_aux_conv2d_backward: dx[..., 0:cin] = dL/dx (nullable), dw = dL/dw given g = dL/d(conv output)
 * `work`: device scratch of hcf_aux_conv2d_workspace(...) bytes; with HCF_PRECISION_F16X3 (3x3 forward only) the int at
 * work[0] is raised when an input leaves the f16 range (the caller zeroes it before a network pass and reads it after).
 * The discriminator's 4x4 stride-2 convs run as squeeze2d + 3x3 conv with a re-indexed weight (hcflow_amd/gan.py). */
size_t hcf_aux_conv2d_workspace(int32_t cin, int32_t cout, int32_t k, int32_t B, int32_t H, int32_t W);
int hcf_aux_conv2d(const float* x, int32_t cs_in, int32_t cin, int32_t B, int32_t H, int32_t W, const float* w,
                   const float* bias, int32_t cout, int32_t k, int32_t act, float* y, int32_t cs_out, void* work,
                   size_t work_bytes, int32_t precision, hcf_stream_t stream);
int hcf_aux_conv2d_backward(const float* x, int32_t cs_in, int32_t cin, int32_t B, int32_t H, int32_t W, const float* w,
                            int32_t cout, int32_t k, const float* g, int32_t cs_g, float* dx, int32_t cs_dx, float* dw,
                            void* work, size_t work_bytes, int32_t precision, hcf_stream_t stream);

/* Range-headroom probe of the f16x3 path (tools/range_headroom.py): while enabled, every conv launch of the following passes
 * also records max |x| over its input windows and -- for layers that own a Winograd pack -- max |B^T d B| over the F(2x2,3x3)
 * input patches, i.e. the values the split has to represent (f16 limit 65504). hcf_debug_range_probe_read returns record
 * `index` (launch order): key = state-dict key of the conv's weight, maxima[2] = {max |x|, max |V| (0 if not Winograd)},
 * info[6] = {cin, cout, H, W, ran on f16x3, has a Winograd pack}; HCF_ERR_KEY past the last record. Debug / evidence only. */
int hcf_debug_range_probe(hcf_engine* e, int32_t enable);
int hcf_debug_range_probe_read(hcf_engine* e, int32_t index, char* key, int32_t key_cap, float* maxima, int32_t* info);

/* Shader clock INSIDE the dominant kernel: while enabled, block 0 of every launch of the 64-channel Winograd kernel adds its life
 * in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime) to two device counters (scalar registers only, one block);
 * hcf_debug_last_clock_mhz() synchronises and returns 100 * cycles / ticks since the enable (0 before any launch; without the
 * probe: the clock of the last hcf_bench_conv call of a TIMERS build). bench.py reads it inside its single-stream timed region
 * (roofline.clock): the board runs this workload at its power cap, the held clock is part of the roofline fraction. */
int hcf_debug_clock_probe(int32_t enable);
double hcf_debug_last_clock_mhz(void);
/* tools/conv_bench.py --ablate: timing-only ablations of the f16x3 kernel (results invalid); 0 = off */
int hcf_debug_set_ablation(int32_t bits);
/* precision used by the per-op entry points hcf_op_conv2d / hcf_bench_conv (process-wide test knob) */
int hcf_op_set_precision(int32_t mode);
/* tools/conv_bench.py: times `iters` back-to-back launches of the conv kernel on random NHWC slabs */
int hcf_bench_conv(int32_t B, int32_t H, int32_t W, const int32_t* src_c, int32_t n_src, int32_t cout, int32_t k,
                   int32_t iters, double* ms_per_launch, double* flops_per_launch, hcf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HCFLOW_H_ */
