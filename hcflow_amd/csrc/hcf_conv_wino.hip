// Winograd F(2x2, 3x3) form of the f16x3 convolution (hcf_conv_wino.h) behind the engine's ConvArgs: used for plain 3x3
// convs whose source windows are multiples of 16 channels, with >= 64 input channels and 32 / 64 output channels -- the
// convs of the residual dense blocks and the 64 -> 64 trunk convs (RRDBNet_arch.py:18-34, 84-97), where it beats the direct
// kernel (profiles/r02_notes.md).
#include "hcf_common.h"
#include <cstdlib>
#include "hcf_conv_wino.h"
#include <atomic>

namespace hcf {

// w: PyTorch [cout][cin][3][3]; bytes of the pack (or 0 when the layer is not eligible / a weight leaves the f16 range)
size_t pack_conv_weights_wino(const float* w, int cin, int cout, const int* srcs, int nsrc, std::vector<float>& out, int min_cin_arg) {
  out.clear();
  static const int min_cin_env = getenv("HCF_WINO_MIN_CIN") ? atoi(getenv("HCF_WINO_MIN_CIN")) : 64;    // experiment knob, read once (bench A/B: 128 -> 112.2, 96 -> 113.5, 64 -> 114.5 img/s)
  const int min_cin = min_cin_arg >= 0 ? min_cin_arg : min_cin_env;      // (the completion launches of the fat schedule: 32 input channels)
  if (!w || cin < min_cin || (cout != 32 && cout != 64) || nsrc < 1 || nsrc > 3) return 0;
  int sum = 0;
  for (int i = 0; i < nsrc; ++i) {
    if (srcs[i] < 16 || (srcs[i] & 15)) return 0;
    sum += srcs[i];
  }
  if (sum != cin) return 0;
  std::vector<uint16_t> pk;
  if (!(cout == 64 ? wino::pack_weights_wino64(w, cin, cout, pk) : wino::pack_weights_wino(w, cin, cout, pk))) return 0;
  out.resize((pk.size() * 2 + 3) / 4);
  memcpy(out.data(), pk.data(), pk.size() * 2);
  return pk.size() * 2;
}

// lane-order pack of a 1x1 64 -> 64 layer for the fused epilogue of the 64-channel kernel (w: [64][64]); 0 = not representable
size_t pack_conv_weights_1x1_frag(const float* w, std::vector<float>& out) {
  out.clear();
  std::vector<uint16_t> pk;
  if (!w || !wino::pack_weights_1x1_frag(w, pk)) return 0;
  out.resize(pk.size() / 2);
  memcpy(out.data(), pk.data(), pk.size() * 2);
  return pk.size() * 2;
}

// Device-side rebuild of a Winograd pack from the PyTorch-layout weight in device memory (after an optimiser step): the same
// arithmetic as wino::pack_weights_wino, one thread per (output channel, input channel).
__device__ __forceinline__ void repack_wino_one(const RepackWinoJob& jb, int idx) {
#pragma clang fp contract(off)
  const int cin = jb.cin, cout = jb.cout, cout_tile = jb.cout_tile;
  uint16_t* __restrict__ pk = reinterpret_cast<uint16_t*>(jb.pk);
  if (jb.frag1x1) {                              // lane-order pack of a 1x1 64 -> 64 layer: idx = (mt, ks, lane, e)
    if (idx >= 4096) return;
    const int e = idx & 7, ln = (idx >> 3) & 63, ks = (idx >> 9) & 3, mt = idx >> 11;
    const int oc = mt * 32 + (ln & 31), ic = 16 * ks + 8 * (ln >> 5) + e;
    const float x = jb.w[oc * 64 + ic];
    const _Float16 hi = (_Float16)x;
    const _Float16 p0 = (_Float16)((float)hi * 2048.f), p1 = (_Float16)((x - (float)hi) * 2048.f);
    const size_t o = ((size_t)((mt * 4 + ks) * 2 + 0) * 64 + ln) * 8 + e;
    pk[o] = __builtin_bit_cast(uint16_t, p0);
    pk[o + 512] = __builtin_bit_cast(uint16_t, p1);
    return;
  }
  const int kn = jb.tr ? jb.kn : cin;
  if (idx >= kn * cout) return;
  const int oc = idx / kn, ic = idx - oc * kn + (jb.tr ? jb.k0 : 0);
  float g[9];
  if (jb.tr) {
    const float* src = jb.w + (size_t)(ic - jb.k0) * jb.ld + (size_t)(jb.tr_off + oc) * 9;
    for (int t = 0; t < 9; ++t) g[t] = src[8 - t];
  } else {
    const int ld = jb.ld ? jb.ld : cin * 9, ld2 = jb.ld2 ? jb.ld2 : cin * 9;
    const float* row = (!jb.w2 || oc < jb.split) ? jb.w + (size_t)oc * ld : jb.w2 + (size_t)(oc - jb.split) * ld2;
    const int col = jb.z1_pad == 0 ? ic : (ic < jb.z1_n ? ic : ic < jb.z1_pad ? -1 : ic - jb.z1_pad + jb.z1_n);
    for (int t = 0; t < 9; ++t) g[t] = col >= 0 ? row[(size_t)col * 9 + t] : 0.f;
  }
  const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  double t[4][3], U[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) U[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
  const int nchunk = cin / 16, nt = oc >> 5, n = oc & 31, c = ic >> 4, h = (ic >> 3) & 1, e = ic & 7;
  for (int xi = 0; xi < 4; ++xi)
    for (int nu = 0; nu < 4; ++nu) {
      const double u = U[xi][nu] * ((xi == 2) ? -1.0 : 1.0) * ((nu == 2) ? -1.0 : 1.0);
      const float x = (float)u;
      const _Float16 hi = (_Float16)x;       // (|x| 2^11 beyond the f16 range becomes inf: the conv raises the range flag)
      const _Float16 p0 = (_Float16)((float)hi * 2048.f), p1 = (_Float16)((float)((u - (double)(float)hi) * 2048.0));
      const size_t o = (cout_tile == 64)   // v4 layout: piece ((pos * 2 + ntile) * 2 + plane) of the chunk's 64
          ? (size_t)c * (wino::W4_BYTES / 2) + (size_t)(((xi * 4 + nu) * 2 + nt) * 2) * 512 + (size_t)h * 256 + (size_t)n * 8 + e
          : ((size_t)(nt * nchunk + c) * wino::W_BYTES) / 2 + (size_t)(((xi * 4 + nu) * 2 + 0) * 2 + h) * 256 + (size_t)n * 8 + e;
      pk[o] = __builtin_bit_cast(uint16_t, p0);
      pk[o + 512] = __builtin_bit_cast(uint16_t, p1);
    }
}

__global__ __launch_bounds__(256) void repack_wino_kernel(const RepackWinoJob j) {
  repack_wino_one(j, blockIdx.x * 256 + threadIdx.x);
}
// every Winograd pack of the net in one launch (the training step's refresh: ~500 packs): block -> job by binary search over blk0
__global__ __launch_bounds__(256) void repack_wino_batch_kernel(const RepackWinoJob* __restrict__ jobs, int njobs) {
  const long long blk = blockIdx.x;
  int lo = 0, hi = njobs;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (jobs[mid].blk0 <= blk) lo = mid; else hi = mid;
  }
  const RepackWinoJob j = jobs[lo];
  repack_wino_one(j, (int)(blk - j.blk0) * 256 + threadIdx.x);
}
int launch_repack_wino_batch(const RepackWinoJob* jobs_dev, int njobs, long long nblocks, hipStream_t st) {
  if (!jobs_dev || njobs < 1 || nblocks < 1 || nblocks > 0x7fffffffLL) return HCF_ERR_ARG;
  hipLaunchKernelGGL(repack_wino_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, jobs_dev, njobs);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// cout_tile = the pack's width (32 / 64; > cout for the zero-padded tiles, whose extra rows stay as the host pack left them)
int launch_repack_wino(const float* w_dev, int cin, int cout, int cout_tile, void* pk, hipStream_t st) {
  if (!w_dev || !pk || cin < 16 || (cin & 15) || (cout_tile != 32 && cout_tile != 64) || cout < 1 || cout > cout_tile) return HCF_ERR_ARG;
  RepackWinoJob j;
  memset(&j, 0, sizeof(j));
  j.w = w_dev; j.pk = pk; j.cin = cin; j.cout = cout; j.cout_tile = cout_tile; j.split = cout;
  hipLaunchKernelGGL(repack_wino_kernel, dim3((unsigned)((cin * cout + 255) / 256)), dim3(256), 0, st, j);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

// In-kernel clock probe of the 64-channel Winograd kernel (bench.py: the dominant family's shader clock inside the timed region):
// per device two counters the kernel's block 0 adds to; enable = allocate / zero, read = synchronise + copy back.
static unsigned long long* g_wino_clk[64] = {nullptr};
static bool g_wino_clk_on = false;
int wino_clock_probe(int enable) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return HCF_ERR_HIP;
  if (enable) {
    if (!g_wino_clk[dev] && hipMalloc(&g_wino_clk[dev], 64) != hipSuccess) { g_wino_clk[dev] = nullptr; return HCF_ERR_NOMEM; }
    if (hipDeviceSynchronize() != hipSuccess || hipMemset(g_wino_clk[dev], 0, 64) != hipSuccess) return HCF_ERR_HIP;
  }
  g_wino_clk_on = enable != 0;
  return HCF_OK;
}
double wino_clock_probe_mhz() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_wino_clk[dev]) return 0.0;
  unsigned long long h[2] = {0, 0};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, g_wino_clk[dev], 16, hipMemcpyDeviceToHost) != hipSuccess || !h[1]) return 0.0;
  return 100.0 * (double)h[0] / (double)h[1];
}

// the round-occupancy rule of launch_conv_wino, for callers that must know BEFORE they commit to a schedule (fat launches)
static int wino_ncu() {
  static int ncu_dev[64] = {0};               // per device (a process may drive several GPUs: nn.DataParallel replicas)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!ncu_dev[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    ncu_dev[dev] = v;
  }
  return ncu_dev[dev];
}
// blocks of a launch of the 32- (ntile_n = 1) / 64-channel (2) kernel = rows of partial sums its fused epilogue backward leaves
int conv_wino_grid(int B, int H, int W, int ntile_n) {
  const long long nunits = (ntile_n == 2) ? (long long)B * ((W + 31) / 32) * ((H + 7) / 8) : (long long)B * ((W + 31) / 32) * ((H + 15) / 16);
  const int ncu = wino_ncu();
  return (int)(nunits < ncu ? nunits : ncu);
}
bool conv_wino_rounds_ok(int B, int H, int W, int ntile_n) {
  const int ncu = wino_ncu();
  const long long nunits = (ntile_n == 2) ? (long long)B * ((W + 31) / 32) * ((H + 7) / 8) : (long long)B * ((W + 31) / 32) * ((H + 15) / 16);
  const long long rounds = (nunits + ncu - 1) / ncu;
  return !(rounds >= 2 && nunits * 100 < rounds * ncu * 75) && (long long)B * H * W < (1LL << 24);
}

int launch_conv_wino(const ConvArgs& a, const void* wpack_wino, hipStream_t st) {
  if (!wpack_wino || a.nsrc < 1 || a.nsrc > 3 || a.w2 || a.tC > 0) return HCF_ERR_UNSUPPORTED;
  wino::Args w;
  memset(&w, 0, sizeof(w));
  int cin = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    if (a.src[i].up != 0) return HCF_ERR_UNSUPPORTED;
    w.src[i] = wino::Src{a.src[i].p, a.src[i].cs, a.src[i].c0, a.src[i].n};
    cin += a.src[i].n;
  }
  for (int i = a.nsrc; i < 3; ++i) w.src[i] = w.src[0];
  w.nsrc = a.nsrc;
  w.B = a.B; w.H = a.H; w.W = a.W;
  w.wpack = reinterpret_cast<const char*>(wpack_wino);
  w.nchunk = cin / 16;
  w.ntile_n = a.wino_ntile > 0 ? a.wino_ntile : a.out.n / 32;
  w.cout = a.wino_ntile > 0 ? ((a.out.n + 3) & ~3) : a.out.n;      // (the padded tile's extra channels are zeros: stored up to the next 4)
  if (a.wino_ntile > 0 ? (w.cout > 32 * w.ntile_n || w.cout > a.out.cs - a.out.c0) : (w.ntile_n * 32 != a.out.n)) return HCF_ERR_UNSUPPORTED;
  w.bias = a.bias; w.scale = a.scale; w.act = a.act;
  w.out = a.out.p; w.out_cs = a.out.cs; w.out_c0 = a.out.c0;
  if (a.res1.p && a.res1_pre) { w.pre = a.res1.p; w.pre_cs = a.res1.cs; w.pre_c0 = a.res1.c0; }
  else if (a.res1.p) { w.res1 = a.res1.p; w.res1_cs = a.res1.cs; w.res1_c0 = a.res1.c0; w.rs1 = a.rs1; }
  if (a.wf1x1) {
    if (a.res1.p || a.res2.p || a.out2.p || !a.bias2 || !a.scale2) return HCF_ERR_UNSUPPORTED;
    w.f_w = reinterpret_cast<const char*>(a.wf1x1); w.f_bias = a.bias2; w.f_scale = a.scale2; w.f_act = a.act2;
  }
  if (a.out2.p) {
    w.out2 = a.out2.p; w.out2_cs = a.out2.cs; w.out2_c0 = a.out2.c0; w.act2 = a.act_t2;
    w.out2_split = (w.ntile_n == 1) ? 16 : 0;          // one 32-channel tile: its upper half (a 16-channel dense block's fat launch)
  }
  if (a.res2.p) {
    if (!a.res1.p || a.res1_pre) return HCF_ERR_UNSUPPORTED;
    w.res2 = a.res2.p; w.res2_cs = a.res2.cs; w.res2_c0 = a.res2.c0; w.rs2 = a.rs2;
  }
  // data-gradient form (bwd_conv): scaled input, optionally the producer's epilogue backward in the 32-channel kernel's epilogue
  w.in_max = a.in_max;
  if (a.fb_y.p) {
    if (!a.in_max || a.fb_scale || a.fb_zy || a.fb_max2 == a.in_max || !a.fb_part || a.res1.p || a.res2.p || w.ntile_n != 1) return HCF_ERR_UNSUPPORTED;
    w.fb_y = a.fb_y.p; w.fb_y_cs = a.fb_y.cs; w.fb_y_c0 = a.fb_y.c0; w.fb_act = a.fb_act;
    w.fb_part = a.fb_part; w.fb_max = a.fb_max; w.fb_max2 = a.fb_max2;
  }
  w.ovf = a.ovf;
  w.zeros = reinterpret_cast<const char*>(a.zeros);
  if (g_wino_clk_on) {
    int dev_ = 0;
    if (hipGetDevice(&dev_) == hipSuccess && dev_ >= 0 && dev_ < 64) w.clk = g_wino_clk[dev_];
  }
  static const int top_wait = getenv("HCF_WINO_TOP_WAIT") ? atoi(getenv("HCF_WINO_TOP_WAIT")) : 0;     // A/B knob, read once
  w.top_wait = top_wait;
  // Consecutive launches walk their units in opposite directions (HCF_WINO_REV=0: all forward): the next conv of a dense block reads
  // what this one read and wrote, and the part touched last is still in the MALL when the walk starts there (profiles/r05_notes.md section 9).
  const char* const rev_env = getenv("HCF_WINO_REV");                 // (read per launch: tests switch it inside one process)
  const int rev_mode = rev_env ? atoi(rev_env) : 1;
  // (per enqueuing THREAD: the two half batches of a split call are enqueued by two threads, one engine each -- a process-wide
  //  counter would hand one of them the even and the other the odd numbers, i.e. no alternation inside either stream)
  static thread_local unsigned rev_flip = 0;
  w.rev = rev_mode ? (int)(rev_flip++ & 1u) : 0;
  const int ncu = wino_ncu();
  // One persistent block per CU walks units of 16 x 32 pixels x 32 channels (8 x 32 x 64 for 64 output channels): a grid of a few rounds with a ragged last one
  // (below 75 % occupancy of the rounds) loses what the kernel gains -- the direct kernel takes those. (The 160 x 160 level of
  // config 2, 800 / 1600 units on 256 CUs = 78 / 89 %, measured equal / slightly better here, stays.)
  if (!conv_wino_rounds_ok(a.B, a.H, a.W, w.ntile_n)) return HCF_ERR_UNSUPPORTED;
  const int r = wino::launch(w, ncu, st, w.ntile_n == 2 ? 4 : 2);      // 64 output channels: the two-tile kernel and its pack layout
  return r == 0 ? HCF_OK : r == -6 ? HCF_ERR_UNSUPPORTED : r == -2 ? HCF_ERR_HIP : HCF_ERR_ARG;
}

}  // namespace hcf
