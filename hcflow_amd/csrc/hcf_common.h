// Shared declarations for the HCFlow MI355X engine (gfx950 only).
//
// Data layout: every activation lives in HBM as dense NHWC fp32, [B][H][W][cs] with channel
// stride cs = roundup4(C) so that a pixel's channel vector is 16-byte aligned and float4-loadable.
// A `View` names a channel window [c0, c0+n) of such a tensor, optionally read through a nearest
// upsample by 2^up (the reference's F.interpolate(mode='nearest'), FlowNet_SR_x4.py:98,117, is
// folded into the consumer's addressing and never materialised).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>

#define HCF_OK 0
#define HCF_ERR_ARG (-1)
#define HCF_ERR_HIP (-2)
#define HCF_ERR_STATE (-3)
#define HCF_ERR_KEY (-4)
#define HCF_ERR_SHAPE (-5)
#define HCF_ERR_UNSUPPORTED (-6)
#define HCF_ERR_NOMEM (-7)

namespace hcf {

struct View {
  float* p;   // base of the NHWC tensor
  int cs;     // channel stride (floats per pixel)
  int c0;     // first channel of the window
  int n;      // channels in the window
  int up;     // log2 nearest-upsample factor when read as a conv source (0 = none)
};

static inline View mkview(float* p, int cs, int c0, int n, int up = 0) {
  View v; v.p = p; v.cs = cs; v.c0 = c0; v.n = n; v.up = up; return v;
}

constexpr int kMaxSrc = 3;

// ---- fused conv (3x3 pad 1, or 1x1) ---------------------------------------------------------
// out[pix][oc] = res2 + rs2 * ( res1 + rs1 * act( (sum_k A[pix,k] W[k,oc] + bias[oc]) * scale[oc] ) )
// (terms with a null residual are skipped). Covers nn.Conv2d+bias(+LeakyReLU), Basic.Conv2d
// (+ActNorm+ReLU), Conv2dZeros (*exp(3 logs)), RDB "x5*0.2+x" and RRDB "out*0.2+x" epilogues.
struct ConvArgs {
  View src[kMaxSrc];
  int nsrc;
  int B, H, W;             // output resolution (sources with up>0 are read at H>>up, W>>up)
  const float* wpack;      // [nchunk][taps][2][npad][8]  (see pack_conv_weights)
  int nchunk;              // ceil(Kvirtual / 16)
  const float* bias;       // [npad]
  const float* scale;      // [npad]
  int act;                 // 0 none, 1 relu, 2 leaky relu 0.2
  View out;                // n = cout
  View res1; float rs1;    // res1.p == nullptr -> skipped
  View res2; float rs2;
  int* ovf;                // f16x3 kernel only: device flag raised when an input exceeds the f16 range
  int any_up;              // set by the launcher: some source window is read through an upsample
  // f16x3 kernel, optional fused second layer (FCN conv2: 1x1, 64 -> 64): out = act2((W2 * act(layer1) + bias2) * scale2)
  const float* w2;         // f16x3 pack of the 1x1 weights (taps = 1) or nullptr
  const float* bias2; const float* scale2; int act2;
  // f16x3 kernel, optional fused inverse flow-step tail (tC > 0): z <- actnorm^-1(W^-1 coupling^-1(z, h = this conv))
  View tz; View tzo; const float* tmat; const float* tbias; const float* tmul; int tC, tns, tmode;
  float* tzpad; int tzpad_n;       // ... and StepArgs::zpad16 / zpad_n
  const float* zeros;      // f16x3 kernel: >= 64 bytes of zeros in device memory (out-of-image halo reads)
  const float* in_max;     // f16x3 kernel, optional (training): device float = max |x| of source 0 -> power-of-two input scaling
  unsigned long long* dbg; // HCF_CONV_TIMERS builds only (tools/conv_bench.py): phase timers of a few mid-grid blocks
  int vec_epi;             // f16x3 kernel, set by the launcher: out / residual views allow 16-byte accesses -> LDS-transposed epilogue
  // Winograd kernels only ("fat" dense-block launches, hcf_engine.hip run_rdb): out2.p != null -> output channels [32, 64) go to
  // out2 with activation act_t2 instead of `out`; res1_pre != 0 -> res1 is a stored partial sum added BEFORE bias / activation
  View out2; int act_t2; int res1_pre;
  // 64-channel Winograd kernel only: the 1x1 second layer of an FCN coupling net in its epilogue (hcf_conv_wino.h Args::f_w);
  // wf1x1 = wino::pack_weights_1x1_frag of the 64 x 64 weights, bias2 / scale2 / act2 as for the fused f16x3 form
  const void* wf1x1;
  // Winograd kernels only: the pack's channel tiles (1 / 2) when the real output width is not 32 / 64 (a DenseBlock coupling net's
  // last conv: 3 .. 42 channels, computed as a zero-padded tile and stored up to the next multiple of 4); 0: out.n / 32
  int wino_ntile;
  // f16x3 kernel, training (gather-form data gradient of a dense block, hcf_engine_train.inc): this conv's output IS the complete
  // dL/dy of the conv that produced `fb_y`; its epilogue applies that conv's epilogue backward on the spot -- out = dL/dpre =
  // dL/dy * act'(y) (activation fb_act, no residuals, unit scale) -- and leaves what conv_epilogue_bwd_kernel would: per-block partial
  // sums of dL/dpre per channel in fb_part ([grid blocks][2][n]; second row: sums of dz * y when fb_zy, else zero; with fb_scale dL/dpre =
  // dz * scale[c], dz = dL/dy * act'(y), as conv_epilogue_bwd_kernel), max |dL/dpre| in fb_max, and
  // max(|dL/dpre|, *in_max) in fb_max2 (a slot OTHER than in_max, which every block of this launch reads: the running max moves on).
  View fb_y; int fb_act; float* fb_part; float* fb_max; float* fb_max2;
  const float* fb_scale; int fb_zy;        // producer's per-channel scale (null: none) and whether the second partial row (sum of dz * y) is wanted
  // set by the f16x3 launcher (scaled training variant only): 0, or W + 1 -- the 8 x 32 tiles walk STRIPS, the B images of a tile
  // row side by side with one zero column between them, instead of every image's own ceil(W / 32) tiles (narrow images: a 40-wide
  // image fills 40 of 64 tile columns, the strip 640 of 656). strip_magic = 2^32 / strip_w + 1 (exact division of < 2^16).
  int strip_w; unsigned strip_magic;
};
// grid of the scaled f16x3 variant for a [B, H, W] tensor (= rows of ConvArgs::fb_part when tile_h is asked for), strips and the
// 4-row tiles of small grids included
int conv_f16x3_scaled_blocks(int B, int H, int W, int* strip_w = nullptr, int* tile_h = nullptr);

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };
// Branch-free activation for unrolled epilogues: a runtime `if (act == ...)` chain per accumulator compiles to a chain of
// scalar branches PER VALUE (~6 us of a 64-channel tile's epilogue). One select on a per-launch slope instead; NaN
// propagates (as torch's relu / leaky_relu do), relu(-inf) = 0.
__device__ __forceinline__ float act_slope(int act) { return act == ACT_RELU ? 0.f : act == ACT_LRELU ? 0.2f : 1.f; }
__device__ __forceinline__ float apply_act(float v, float slope) {
  // v > 0 or NaN -> v; otherwise slope * v, with relu's -inf clamped first so that 0 * -inf cannot make a NaN
  const float lo = (slope == 0.f) ? -3.0e38f : -INFINITY;          // loop-invariant
  return !(v <= 0.f) ? v : slope * fmaxf(v, lo);
}
enum { PREC_EXACT = 0, PREC_F16X3 = 1 };

int launch_conv(const ConvArgs& a, int taps, hipStream_t st);
// fp32-equivalent conv on f16 MFMA (3-term split, hcf_conv_f16x3.hip); wpack = f16x3 pack
int launch_conv_f16x3(const ConvArgs& a, int taps, hipStream_t st);
// FCN conv1 (3x3, <= 16 input channels = one K chunk) + conv2 (1x1 64 -> 64) as one persistent launch with resident weights
// (hcf_conv_fcn.hip); a.res1 = optional pre-activation term of conv1 (64 channels). HCF_ERR_UNSUPPORTED: use the generic kernel
int launch_fcn12(const ConvArgs& a, hipStream_t st);
// Winograd F(2x2,3x3) form of the same f16x3 conv (hcf_conv_wino.hip): eligible layers only (pack size 0 otherwise);
// HCF_ERR_UNSUPPORTED when the call cannot take it (upsampled source, unaligned views, > 2^24 pixels): use the direct kernel
size_t pack_conv_weights_wino(const float* w, int cin, int cout, const int* srcs, int nsrc, std::vector<float>& out, int min_cin = -1);
size_t pack_conv_weights_1x1_frag(const float* w, std::vector<float>& out);
int launch_conv_wino(const ConvArgs& a, const void* wpack_wino, hipStream_t st);
int wino_clock_probe(int enable);            // hcf_conv_wino.hip: in-kernel clock probe of the 64-channel kernel
double wino_clock_probe_mhz();
int conv_wino_grid(int B, int H, int W, int ntile_n);           // blocks of such a launch (= rows of partial sums of its fused epilogue backward)
bool conv_wino_rounds_ok(int B, int H, int W, int ntile_n);     // false: launch_conv_wino would hand this launch to the direct kernel
int launch_repack_wino(const float* w_dev, int cin, int cout, int cout_tile, void* pk, hipStream_t st);   // pack rebuilt from device weights
// One Winograd pack rebuilt from device weights. Element (oc, ic) of the PACK comes from row oc of `w` (oc < split) or row oc - split
// of `w2`, rows `ld` / `ld2` floats apart (0: cin * 9, a plain [cout][cin][3][3] tensor), input channel ic of the pack = channel
// ic of the row (z1_pad == 0) or, for the packs whose first source is z1 zero-padded to z1_pad channels: ic < z1_n -> ic, ic < z1_pad
// -> zero, else ic - z1_pad + z1_n. The derived packs of hcf_engine.hip (fat dense-block pairs, padded-z1 FCN / DenseBlock convs) are
// all of this shape. frag1x1 != 0: the job is the lane-order pack of a 1x1 64 -> 64 layer instead (wino::pack_weights_1x1_frag).
struct RepackWinoJob { const float* w; void* pk; int cin, cout, cout_tile; long long blk0;                             // blk0: first block of the job (ascending)
                       const float* w2; int split, ld, ld2, z1_n, z1_pad, frag1x1;
                       // tr != 0 (gather-form data-gradient packs of a dense block, hcf_engine_train.inc): the job rewrites input channels
                       // [k0, k0 + kn) of a pack of `cin` input channels from ONE forward conv's weight, transposed and flipped:
                       // L[n][k][t] = w[(k - k0) * ld + (tr_off + n) * 9 + (8 - t)], n < cout
                       int tr, k0, kn, tr_off; };
int launch_repack_wino_batch(const RepackWinoJob* jobs_dev, int njobs, long long nblocks, hipStream_t st);

// ---- conv weight gradient (training path) ---------------------------------------------------------------------
// dW[oc][ic][tap] += sum_pixels X[pixel + tap][ic] * G[pixel][oc]   (PyTorch weight layout, fp32 atomics)
struct WgradArgs {
  View src[kMaxSrc];       // the forward conv's input windows (cat order), optional nearest upsample
  int nsrc;
  View g;                  // gradient w.r.t. the conv's pre-activation output, n = cout
  int B, H, W;
  int taps;                // 9 (3x3, pad 1) or 1
  float* dw;               // [cout][cin_total][taps], accumulated (caller zero-initialises)
  int negate;              // dW -= ... instead of += (the inverse 1x1 conv's weight gradient)
  const float* g_max;      // nullable (device): max |g|; non-null selects the f16x3 matrix-core kernel (g scaled by a power of two)
  float* part;             // scratch for the per-block partial tiles, >= conv_wgrad_scratch_floats(a) floats
  size_t part_cap;         // capacity of `part` in floats
  int blocks_hint;         // f16x3 kernel: blocks per launch to aim for (0: 256 = one per CU). The engine asks for 160 when the
                           // weight gradients run on their own stream BESIDE the data-gradient chain, so the two do not share CUs
  int cin_total, tpb;      // set by the launcher
  int strip_w;             // set by the launcher (f16x3 kernel): 0, or W + 1: the tiles walk STRIPS -- the B images of a tile row side
  unsigned strip_magic;    // by side with one zero column between them (virtual width B (W + 1)) -- instead of every image's own
                           // ceil(W / 32) tiles: a 40-wide image fills 40 of 64 tile columns, the strip 640 of 656. Pixels are the
                           // reduction dimension here, so only the walk changes. strip_magic = 2^32 / strip_w + 1 (exact division).
  int dbg;                 // timing ablations (HCF_WG_DBG): 1 no MFMAs, 2 no epilogue, 4 no global loads
};
size_t conv_wgrad_scratch_floats(const WgradArgs& a, int* nblk_x = nullptr, int* tpb = nullptr);
// The fixed-order reduction of a launch's partial tiles as a JOB: launch_conv_wgrad(a, st, &job) leaves the partial tiles in a.part
// and returns the reduce step instead of launching it; launch_wgrad_reduce_batch runs any number of them as ONE launch (the training
// step: ~630 reduce launches of ~7 us become ~40). Same summation order per dW element as the stand-alone reduce.
struct WgradReduceJob { WgradArgs a; int nx, nicb, nocb, nbx; long long blk0; };
int launch_conv_wgrad(const WgradArgs& a, hipStream_t st, WgradReduceJob* defer = nullptr);
constexpr int kWgBatchMax = 5;
struct WgradBatchArgs { WgradArgs a[kWgBatchMax]; int blk0[kWgBatchMax + 1], nicb[kWgBatchMax], nocb[kWgBatchMax], n; };   // kernel argument
int launch_conv_wgrad_batch(const WgradArgs* jobs, int n, hipStream_t st, WgradReduceJob* defer);
int launch_wgrad_reduce_batch(const WgradReduceJob* jobs_dev, int njobs, long long nblocks, hipStream_t st);
int launch_absmax(const View& g, int B, int H, int W, float* out, hipStream_t st);   // *out = max |g| (out zero-initialised)
int launch_wino_vmax(const View& g, int B, int H, int W, float* out, hipStream_t st);   // *out = max |B^T d B| over the F(2x2,3x3) patches

// ---- device-side weight repack (hcf_repack.hip) -------------------------------------------------------------------
struct RepackArgs {
  const float* w;          // PyTorch-layout weight in device memory
  int cin_w, taps;         // second dimension of w, taps (9 / 1)
  int transposed, off;     // 1: data-gradient pack of w's input-channel block [off, off + cout)
  int cout;                // output channels of the pack (forward: cout; transposed: block size)
  int srcs[kMaxSrc], nsrc; // input windows of the packed conv (transposed: one window = the forward cout)
  int nchunk, npad;
  float* pk;               // exact pack or nullptr
  _Float16* pk16;          // f16x3 pack or nullptr
};
int launch_repack_conv(const RepackArgs& a, hipStream_t st);
// batched forms for the per-step refresh of a training run (thousands of tiny repacks -> three launches): job tables live
// in device memory; prefix[j] = first block of job j, prefix[njobs] = total blocks
struct RepackEpiJob { int kind, cout; const float* b; const float* l; float* bias; float* scale; };
struct CopyJob { const float* src; float* dst; int n; };
int launch_repack_conv_batch(const RepackArgs* jobs, const long long* prefix, int njobs, long long nblocks, hipStream_t st);
int launch_repack_epilogue_batch(const RepackEpiJob* jobs, int njobs, hipStream_t st);
int launch_copy_jobs(const CopyJob* jobs, int njobs, hipStream_t st);      // dst[0..n) = src[0..n) per job (gather / scatter of small tables)
int launch_repack_epilogue(int kind, const float* b, const float* l, int cout, float* bias, float* scale, hipStream_t st);

// ---- flow-step glue --------------------------------------------------------------------------
enum { CPL_AFFINE = 0, CPL_SHIFT3 = 1 };

struct StepArgs {
  int B, H, W;
  int C;                 // channels of z
  int ns;                // CPL_AFFINE: channels [ns, C) are transformed, h = (shift, scale) interleaved
                         // CPL_SHIFT3: channels [0, 3) are shifted by h[0:3], ns unused
  int mode;
  View z;                // input
  View h;                // coupling-network output
  View out;              // output
  const float* mat;      // C x C row-major (W^-1 for inverse, W for forward) or nullptr (no permutation)
  const float* an_bias;  // [C]
  const float* an_mul;   // [C]  exp(-logs) (inverse) or exp(logs) (forward)
  float* partial;        // forward couple: per-block partial sums of logscale, [B][nblk]; else nullptr
  int partial_stride;    // floats between consecutive samples in `partial`
  View aux;              // forward head, training tape: also store the ActNorm output (input of W); p == nullptr: no
  float* zpad16; int zpad_n;   // inverse tail: also store out[:, :zpad_n] as a zero-padded 16-channel tensor (or nullptr)
};

int launch_step_tail_inv(const StepArgs& a, hipStream_t st);     // coupling^-1, W^-1, actnorm^-1
int launch_step_head_fwd(const StepArgs& a, hipStream_t st);     // actnorm, W           (h unused)
int launch_step_couple_fwd(const StepArgs& a, hipStream_t st);   // coupling (in place capable) + sum logscale
int step_blocks_per_sample(int H, int W);
int step_cmax(int C);   // register-array bucket (8/12/24/48) used by the step kernels; -1 if C > 48

// ---- backward kernels of the training path (hcf_train.hip) ---------------------------------
// Backward of the fused conv epilogue  y = res2 + rs2 * (res1 + rs1 * act((acc + bias) * scale)):
//   g2 += gy ;  g1 += gy * rs2 ;  dz = gy * rs2 * rs1 * act'(y) ;  gpre = dz * scale  (= dL/d acc)
//   sum_pre[c] += sum_pixels gpre          (conv bias / ActNorm bias / Conv2dZeros bias gradient)
//   sum_zy[c]  += zy_mult * sum_pixels dz * y   (ActNorm logs: 1, Conv2dZeros logs: 3; act in {none, relu} there)
struct EpiBwdArgs {
  int B, H, W;
  View gy, y;              // n = cout; y is read only when act != none or sum_zy != nullptr
  const float* scale;      // [npad] forward epilogue scale
  int act;
  float rs1, rs2;          // as in the forward call (ignored for a missing residual)
  int has1, has2;          // the forward conv had res1 / res2 (rs1 / rs2 are meaningful)
  View g1, g2;             // gradient buffers of res1 / res2 (p may be nullptr: that gradient is not needed)
  View gpre;
  float* sum_pre;          // nullable
  float* sum_zy;           // nullable
  float zy_mult;
  float* absmax;           // nullable: max |gpre| as float bits (atomicMax on the int view; zero-initialised)
  float* absmax2;          // nullable: a second slot raised the same way (shared by the convs of one dense block: the running
                           // max over the gradients a gather-form data-gradient conv reads, hcf_engine_train.inc)
  const float* carry2;     // nullable: a slot whose value is folded into absmax2 as well (the running max so far: the next conv of the
                           // dense block reads absmax2 while later kernels raise their own slot, so the running max moves slot to slot)
  float* part;             // per-block partial sums [conv_epilogue_bwd_blocks()][2][gy.n] (sum_pre, sum_zy): reduced in a fixed
                           // order by launch_sum_jobs, so parameter gradients are bit-reproducible; nullptr: fp32 atomics into sum_*
};
int launch_conv_epilogue_bwd(const EpiBwdArgs& a, hipStream_t st);
int conv_epilogue_bwd_blocks(int B, int H, int W);

// Fixed-order reduction of per-block partial sums into parameter gradients: dst0[c] += sum_b part[b][0][c],
// dst1[c] += mult1 * sum_b part[b][1][c] (double accumulation, blocks in index order); one block per job.
struct SumJob { const float* part; int nblk, n, pstride; float* dst0; float* dst1; float mult1; };
int launch_sum_jobs(const SumJob* jobs_dev, int njobs, hipStream_t st);

struct StepBwdArgs {
  int B, H, W, C, ns, mode;
  // coupling backward: (gzout, zout, h) -> gzb (=), gh (=)
  View gzout, zout, h, gzb, gh;
  float gobj;              // dL/d(objective) of every sample (the sum-of-logscale term feeds the log-det)
  // head backward: (gzb, za) -> gzin (=), sums for ActNorm bias / logs
  View za, gzin;
  const float* matT;       // [CMAX][CMAX] W^T (row c = column c of W) or nullptr (no permutation)
  const float* an_mul;     // exp(logs)
  float* g_bias;           // [C] +=
  float* g_logs;           // [C] +=
  float* part;             // per-block partials [B * step_blocks_per_sample][2][step_cmax(C)] (bias, logs) or nullptr (atomics)
};
int launch_step_couple_bwd(const StepBwdArgs& a, hipStream_t st);
int launch_step_head_bwd(const StepBwdArgs& a, hipStream_t st);

// backward of one INVERSE flow step (FlowStep.reverse_flow, FlowStep.py:53-64):  z -> zc = coupling^-1(z, h) -> y = W^-1 zc
// -> x = y e^-s - b. Given gx: gy = gx e^-s, gzc = W^-T gy, gz (coupling^-1 backward), gh; sums for bias / logs;
// y and gzc are materialised for the 1x1 weight-gradient kernel (dW = -sum gzc y^T).
struct StepInvBwdArgs {
  int B, H, W, C, ns, mode;
  View gx, x, zc, h;           // in
  View gz, gh, gzc, y;         // out (=)
  const float* matInvT;        // [CMAX][CMAX], row j = column j of W^-1, or nullptr
  const float* an_bias;        // b
  const float* mul_fwd;        // e^s
  const float* mul_inv;        // e^-s
  float* g_bias;               // [C] +=
  float* g_logs;               // [C] +=
  float* part;                 // as StepBwdArgs::part
};
int launch_step_inv_bwd(const StepInvBwdArgs& a, hipStream_t st);
int launch_mask_unit_range(View z, View g, int B, int H, int W, hipStream_t st);      // g = (0 <= z <= 1) ? g : 0

struct PriorBwdArgs {
  int B, H, W, C;          // C = channels of the latent; h has 2C (mean = h[0::2], s = h[1::2])
  View a, h, ga, gh;       // ga (=), gh (=)
  float gobj;
  int rescale;             // 0: logs = s (SR); 1: logs = 0.318 atan(2 s) (rescaling net, ConditionalFlow.py:78,90)
  const float* gz_nchw;    // encode backward: dL/dz of z = (a - mean) e^-logs, NCHW [B,C,H,W] (nullptr: zero)
};
int launch_gauss_logp_bwd(const PriorBwdArgs& a, hipStream_t st);
// a = mean + e^logs * eps (SR prior sample): gh[2c] = ga[c], gh[2c+1] = ga[c] * (a[c] - mean)   (ga in, gh out)
int launch_gauss_sample_bwd(const PriorBwdArgs& a, hipStream_t st);
// z = (a - mean) e^-logs (rescaling forward, ConditionalFlow.py:76-80): ga (=), gh (=) from gz_nchw
int launch_gauss_encode_bwd(const PriorBwdArgs& a, hipStream_t st);
// gz[pix][c] += g_nchw[b][c][pix] (masked where z is outside [0,1] when clamp01): backward of an NCHW (clamped) output
int launch_add_nchw_grad(const float* g_nchw, View z, View gz, int B, int H, int W, int clamp01, hipStream_t st);
// out = (0 <= raw <= 1) ? g : 0 over n floats (flat NCHW tensors)
int launch_mask_flat(const float* g, const float* raw, float* out, size_t n, hipStream_t st);
// d/dz of logp(lr; mean := Quant(z), logs = -6) with the straight-through Quant: gz += gobj * (lr - q(z)) * e^12
int launch_quant_logp_bwd(View z, const float* lr_nchw, View gz, int B, int H, int W, float gobj, hipStream_t st);
int launch_add_view(View in, View out, int B, int H, int W, float alpha, hipStream_t st);      // out += alpha * in
int launch_add_const(float* p, size_t n, float v, hipStream_t st);                             // p[i] += v
int launch_axpy(const float* x, float* y, size_t n, float alpha, hipStream_t st);              // y += alpha * x
// many of the two above in one launch (the data-independent log-det terms of a training step: 2 small updates per flow step)
struct AxpyJob { const float* x; float* y; int n; float alpha; };                              // y += alpha * (x ? x : 1)
int launch_axpy_jobs(const AxpyJob* jobs_dev, int njobs, hipStream_t st);
// LU-decomposed invertible 1x1 conv (Permutations.py:78-86: W = P L U', L = l o mask + I, U' = u o mask^T + diag(sign_s e^log_s)):
// chain rule of dL/dW into the factors, A = P^T dW:  dl += strict_lower(A U'^T),  du += strict_upper(L^T A),
// dlog_s[i] += (L^T A)[i][i] * U'[i][i].  All matrices [C][C] row-major, C <= 48; one block.
struct LuChainArgs { const float *dW, *P, *L, *U; float *dl, *du, *dlog_s; int C; };
int launch_lu_chain(const LuChainArgs& a, hipStream_t st);

// ---- Gaussian prior / misc elementwise -------------------------------------------------------
struct GaussArgs {
  int B, H, W, C;        // C = channels of the latent (h has 2C: mean = h[0::2], s = h[1::2])
  View h;
  int rescale;           // 0: logs = s (SR, ConditionalFlow.py:54,62); 1: logs = 0.318 atan(2 s) (:78,90)
  // sample
  const float* eps;      // NCHW [B,C,H,W] already N(0,tau) (injected), or nullptr -> device Philox
  float tau; uint64_t seed; uint64_t offset;
  int64_t b0;            // device draws: sample b of this call is sample b0 + b of the (sharded) batch, so shards reproduce the full batch
  View out;              // sample: latent out.  logp/encode: latent in
  float* aux;            // encode (rescale fwd): NCHW output z = (a-mean) exp(-logs)
  float* partial; int partial_stride;
};
int launch_gauss_sample(const GaussArgs& a, hipStream_t st);
int launch_gauss_logp(const GaussArgs& a, hipStream_t st);      // SR forward: partial sums of log p
int launch_gauss_encode(const GaussArgs& a, hipStream_t st);    // rescaling forward

int launch_nchw_to_nhwc(const float* src, View dst, int B, int C, int H, int W, hipStream_t st);
int launch_nhwc_to_nchw(View src, float* dst, int B, int C, int H, int W, int clamp01, hipStream_t st);
// squeeze2d / unsqueeze2d, factor 2, channel order c*4 + i*2 + j (Basic.py:127-157)
int launch_squeeze(View in, View out, int B, int C, int H, int W, hipStream_t st);     // in: [H,W,C] -> out [H/2,W/2,4C]
int launch_unsqueeze(View in, View out, int B, int C4, int H, int W, hipStream_t st);  // in: [H,W,C4] -> out [2H,2W,C4/4]
// fused boundary forms
int launch_nchw_squeeze(const float* src, const float* noise, float quant, View out, int B, int C, int H, int W,
                        int haar, hipStream_t st);   // (src + noise/quant) -> squeeze/haar -> NHWC
int launch_unsqueeze_nchw(View in, float* dst, int B, int C4, int H, int W, int haar, int clamp01, hipStream_t st);
int launch_haar_fwd(View in, View out, int B, int C, int H, int W, hipStream_t st);
int launch_haar_inv(View in, View out, int B, int C4, int H, int W, hipStream_t st);
int launch_copy_view(View in, View out, int B, int H, int W, hipStream_t st);
int launch_copy_pad16(View in, float* out16, int B, int H, int W, hipStream_t st);   // [B, H, W, 16]: channels of `in` (<= 16), then zeros
int launch_copy_pad(View in, View out, int B, int H, int W, hipStream_t st);         // out.n (a multiple of 16) channels: `in`'s, then zeros
// SR forward tail: zq = round(clamp(z,0,1)*255)/255 ; lr_hat NCHW = zq ; partial += logp(lr; mean zq, logs -6)
int launch_quant_logp(View z, const float* lr_nchw, float* lr_hat_nchw, int B, int H, int W,
                      float* partial, int partial_stride, hipStream_t st);
// out[b] = cst + sum_j partial[b*stride + j], j < n  (double accumulation); optional nll = mean(-out)/(ln2*pixels)
int launch_reduce_partials(const float* partial, int stride, int n, int B, double cst, double pixels,
                           float* out_logdet, float* out_nll, hipStream_t st);
int launch_fill(float* p, size_t n, float v, hipStream_t st);
// out[0..n) = per-channel sum, out[n..2n) = per-channel sum of squares over all B*H*W pixels of the window (double)
int launch_channel_stats(View v, int B, int H, int W, double* out, hipStream_t st);
// backward of the nearest upsample: out[b, y, x, c] (+)= sum over the 2^up x 2^up block of `in` ([B, Ho << up, Wo << up])
int launch_sumpool(View in, View out, int B, int Ho, int Wo, int up, int accumulate, hipStream_t st);

}  // namespace hcf
