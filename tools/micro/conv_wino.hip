// Stand-alone harness for the Winograd f16x3 convolution (hcflow_amd/csrc/hcf_conv_wino.h): checks it against a naive fp64
// direct convolution of the same fp32 inputs and times the RDB shapes of config 2 (B = 16, 320^2 / 160^2).
//   hipcc -O3 --offload-arch=gfx950 -I hcflow_amd/csrc -I tools/micro tools/micro/conv_wino.hip -o build/micro/conv_wino && build/micro/conv_wino
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "hcf_conv_wino.h"
#include "hcf_conv_wino6.h"        // version 10: F(4x4, 3x3), 64 output channels (round 6; tools/micro only, profiles/r06_notes.md)
// (versions 1, 5, 6 / 7, 8 / 9 of the series -- first kernel, ping-pong, pipelined row phase, 1-D form -- were removed in round 6:
//  their measurements are in profiles/r03..r05_notes.md, the sources in the history before this commit)

using namespace hcf::wino;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__host__ __device__ inline float hashf(uint64_t i, uint32_t salt) {
  uint64_t x = i * 0x9E3779B97F4A7C15ull + salt * 0xBF58476D1CE4E5B9ull;
  x ^= x >> 31; x *= 0x94D049BB133111EBull; x ^= x >> 29;
  return ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * 2.0f;
}
__global__ void fill_f32(float* p, long long n, uint32_t salt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = hashf((uint64_t)i, salt);
}
// inputs: source s = channels [0, n_s) of its own NHWC tensor with stride cs_s
__global__ void ref_conv(const float* s0, int cs0, int n0, const float* s1, int cs1, int n1, const float* w, const float* bias,
                         const float* scale, int act, const float* res1, int res1_cs, float rs1, int B, int H, int W, int cout,
                         float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H * W * cout) return;
  const int oc = (int)(i % cout); const long long pix = i / cout;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
  const int cin = n0 + n1;
  double s = 0;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = y + dy - 1, xx = x + dx - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const size_t p = (size_t)((size_t)b * H + yy) * W + xx;
      for (int ic = 0; ic < cin; ++ic) {
        const double a = ic < n0 ? (double)s0[p * cs0 + ic] : (double)s1[p * cs1 + (ic - n0)];
        s += a * (double)w[((size_t)oc * cin + ic) * 9 + dy * 3 + dx];
      }
    }
  float v = (float)((s + (double)bias[oc]) * (double)scale[oc]);
  if (act == 1) v = v > 0 ? v : 0; else if (act == 2) v = v > 0 ? v : 0.2f * v;
  if (res1) v = v * rs1 + res1[pix * res1_cs + oc];
  out[i] = v;
}

// second layer of the fused form: out[p][oc] = act((sum_ic h[p][ic] w2[oc][ic] + bias2[oc]) * scale2[oc]), fp64
__global__ void ref_1x1(const float* h, const float* w2, const float* bias2, const float* scale2, int act, long long npix, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * 64) return;
  const int oc = (int)(i & 63); const long long pix = i >> 6;
  double s = 0;
  for (int ic = 0; ic < 64; ++ic) s += (double)h[pix * 64 + ic] * (double)w2[oc * 64 + ic];
  float v = (float)((s + (double)bias2[oc]) * (double)scale2[oc]);
  if (act == 1) v = v > 0 ? v : 0; else if (act == 2) v = v > 0 ? v : 0.2f * v;
  out[i] = v;
}

struct Prob { const char* name; int B, H, W, n0, n1, cout, act; bool res; bool fuse = false; };

int main(int argc, char** argv) {
  int only = argc > 1 ? atoi(argv[1]) : -1;
  const int version = argc > 2 ? atoi(argv[2]) : 2;
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  if (getenv("NCU")) ncu = atoi(getenv("NCU"));      // fewer resident blocks: how much of a unit's time is contention for HBM / fabric?
  const Prob probs[] = {
      {"check small 64->32  lrelu", 2, 24, 40, 64, 0, 32, 2, false},
      {"check small 64+128->64 res", 1, 17, 70, 64, 128, 64, 0, true},
      {"check small 64+32->32 relu", 1, 8, 32, 64, 32, 32, 1, false},
      {"check small 16->32 ragged", 3, 5, 7, 16, 0, 32, 0, false},
      {"rdb conv1 L0  64->32  @320", 16, 320, 320, 64, 0, 32, 2, false},
      {"rdb conv2 L0  96->32  @320", 16, 320, 320, 64, 32, 32, 2, false},
      {"rdb conv4 L0 160->32  @320", 16, 320, 320, 64, 96, 32, 2, false},
      {"rdb conv5 L0 192->64  @320", 16, 320, 320, 64, 128, 64, 0, true},
      {"rdb conv4 L0 160->32  @320 B=4", 4, 320, 320, 64, 96, 32, 2, false},
      {"rdb conv1 L1  64->32  @160", 16, 160, 160, 64, 0, 32, 2, false},
      {"rdb conv5 L1 192->64  @160", 16, 160, 160, 64, 128, 64, 0, true},
      // components of the "fat launch" schedule of an RDB (profiles/r03_notes.md): A = conv1 + conv2|x0 (64 -> 64),
      // B / D = completion of conv2 / conv4 (32 -> 32, + stored partial read through the residual slot),
      // C = conv3 + conv4|[x0 x1 x2] (128 -> 64); E = conv5 as today
      {"fat A L0  64->64  @320", 16, 320, 320, 64, 0, 64, 2, false},
      {"fat B L0  32->32  @320 +partial", 16, 320, 320, 32, 0, 32, 2, true},
      {"fat C L0 128->64  @320", 16, 320, 320, 64, 64, 64, 2, false},
      {"rdb conv3 L0 128->32 @320", 16, 320, 320, 64, 64, 32, 2, false},
      {"fat A L1  64->64  @160", 16, 160, 160, 64, 0, 64, 2, false},
      {"fat B L1  32->32  @160 +partial", 16, 160, 160, 32, 0, 32, 2, true},
      {"fat C L1 128->64  @160", 16, 160, 160, 64, 64, 64, 2, false},
      // conditional FCN coupling net: conv1 (3x3, [z1 padded to 16 | 128 features] -> 64, ActNorm, ReLU) + conv2 (1x1 64 -> 64) fused
      {"check fcn 16+128->64 +1x1", 2, 19, 45, 16, 128, 64, 1, false, true},
      {"check fcn 16+128->64 +1x1 b", 1, 8, 32, 16, 128, 64, 1, false, true},
      {"fcn cond L0 144->64 +1x1 @320", 16, 320, 320, 16, 128, 64, 1, false, true},
      {"fcn cond L1 144->64 +1x1 @160", 16, 160, 160, 16, 128, 64, 1, false, true},
      {"fcn cond L0 144->64 alone @320", 16, 320, 320, 16, 128, 64, 1, false, false},
      {"check 64->64 lrelu ragged", 2, 33, 65, 64, 0, 64, 2, false},
      {"check 64+64->64 exact tiles", 3, 16, 32, 64, 64, 64, 1, false},
      {"check 16->64 tiny", 1, 3, 5, 16, 0, 64, 0, true},
      // is the memory side of these kernels HBM or would a MALL-resident working set (256 MB) change it? the same launch repeated:
      // 2 samples of 256 x 256 = 168 MB of input + residual + output (2 full rounds of 256 units) against 16 samples (1.3 GB, 16 rounds)
      {"conv5 192->64 @256 B=2 (168 MB)", 2, 256, 256, 64, 128, 64, 0, true},
      {"conv5 192->64 @256 B=16 (1.3 GB)", 16, 256, 256, 64, 128, 64, 0, true},
      {"fat C 128->64 @256 B=2", 2, 256, 256, 64, 64, 64, 2, false},
      {"fat C 128->64 @256 B=16", 16, 256, 256, 64, 64, 64, 2, false},
      {"conv5 192->64 @256 B=1 (84 MB)", 1, 256, 256, 64, 128, 64, 0, true},
      {"conv5 192->64 @256 B=3 (252 MB)", 3, 256, 256, 64, 128, 64, 0, true},
      {"conv5 192->64 @256 B=4 (336 MB)", 4, 256, 256, 64, 128, 64, 0, true},
      {"conv5 192->64 @256 B=6 (503 MB)", 6, 256, 256, 64, 128, 64, 0, true},
      {"conv5 192->64 @256 B=8 (671 MB)", 8, 256, 256, 64, 128, 64, 0, true},
      // round 6 (F(4x4,3x3), version 10): several samples and units per block with residual / activation / ragged edges
      {"check 64+128->64 res lrelu B=2", 2, 21, 37, 64, 128, 64, 2, true},
      {"check 16+64->64 relu B=3", 3, 16, 33, 16, 64, 64, 1, false},
      {"check 64->64 res B=5 exact", 5, 32, 32, 64, 0, 64, 0, true},
      {"check 64+64->64 B=1 H=7", 1, 7, 100, 64, 64, 64, 2, false},
      {"check 64->64 plain B=2", 2, 16, 32, 64, 0, 64, 0, false},
  };
  const int nprob = sizeof(probs) / sizeof(probs[0]);
  for (int pi = 0; pi < nprob; ++pi) {
    if (only >= 0 && pi != only) continue;
    const Prob& P = probs[pi];
    const bool check = P.B * P.H * P.W <= 8192;
    const long long npix = (long long)P.B * P.H * P.W;
    const int cin = P.n0 + P.n1, cs0 = P.n0 < 64 ? P.n0 : 64, cs1 = 128;   // x lives in a 64-channel tensor, the growth slab has 128
    float *s0, *s1, *out, *res, *dw, *dbias, *dscale; char* wpk; int* ovf;
    CK(hipMalloc(&s0, (size_t)npix * cs0 * 4 + 4096)); CK(hipMalloc(&s1, (size_t)npix * cs1 * 4 + 4096));
    CK(hipMalloc(&out, (size_t)npix * 64 * 4)); CK(hipMalloc(&res, (size_t)npix * 64 * 4));
    CK(hipMalloc(&ovf, 256)); CK(hipMemset(ovf, 0, 256));
    fill_f32<<<(unsigned)((npix * cs0 + 255) / 256), 256>>>(s0, npix * cs0, 7);
    fill_f32<<<(unsigned)((npix * cs1 + 255) / 256), 256>>>(s1, npix * cs1, 9);
    fill_f32<<<(unsigned)((npix * 64 + 255) / 256), 256>>>(res, npix * 64, 11);
    CK(hipMemset(out, 0, (size_t)npix * 64 * 4));
    std::vector<float> w((size_t)P.cout * cin * 9), bias(64, 0.f), scale(64, 1.f);
    for (size_t i = 0; i < w.size(); ++i) w[i] = hashf(i, 3) * 0.5f / sqrtf((float)cin * 9.f);
    for (int i = 0; i < P.cout; ++i) { bias[i] = hashf(i, 5) * 0.1f; scale[i] = 1.f + 0.1f * hashf(i, 6); }
    std::vector<uint16_t> pk;
    const int ver = ((version >= 4) && P.cout != 64) ? 2 : version;         // version 4 / 10 are 64-output-channel kernels
    if (ver == 10 && P.fuse) { printf("%-34s (no fused 1x1 form in version 10)\n", P.name); continue; }
    if (!((ver == 10) ? hcf::wino6::pack_weights_wino6(w.data(), cin, P.cout, pk) : (ver >= 4) ? pack_weights_wino64(w.data(), cin, P.cout, pk) : pack_weights_wino(w.data(), cin, P.cout, pk))) { printf("pack failed\n"); return 1; }
    CK(hipMalloc(&wpk, pk.size() * 2)); CK(hipMemcpy(wpk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbias, 256)); CK(hipMalloc(&dscale, 256));
    CK(hipMemcpy(dbias, bias.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dscale, scale.data(), 256, hipMemcpyHostToDevice));
    Args a; memset(&a, 0, sizeof(a));
    a.src[0] = {s0, cs0, 0, P.n0}; a.nsrc = 1;
    if (P.n1) { a.src[1] = {s1, cs1, 0, P.n1}; a.nsrc = 2; }
    a.src[2] = a.src[0]; if (a.nsrc == 1) a.src[1] = a.src[0];
    a.B = P.B; a.H = P.H; a.W = P.W; a.wpack = wpk; a.nchunk = cin / 16; a.ntile_n = P.cout / 32;
    a.bias = dbias; a.scale = dscale; a.act = P.act; a.out = out; a.out_cs = 64; a.out_c0 = 0; a.cout = P.cout;
    if (P.res) { a.res1 = res; a.res1_cs = 64; a.res1_c0 = 0; a.rs1 = 0.2f; }
    a.ovf = ovf; a.zeros = reinterpret_cast<const char*>(ovf) + 64;
    std::vector<float> w2(64 * 64), bias2(64), scale2(64);
    float *dw2 = nullptr, *dbias2 = nullptr, *dscale2 = nullptr; char* dfw = nullptr;
    if (P.fuse) {
      for (size_t i = 0; i < w2.size(); ++i) w2[i] = hashf(i, 13) * 0.5f / 8.f;
      for (int i = 0; i < 64; ++i) { bias2[i] = hashf(i, 15) * 0.1f; scale2[i] = 1.f + 0.1f * hashf(i, 16); }
      std::vector<uint16_t> fp;
      if (!pack_weights_1x1_frag(w2.data(), fp)) { printf("1x1 pack failed\n"); return 1; }
      CK(hipMalloc(&dfw, fp.size() * 2)); CK(hipMemcpy(dfw, fp.data(), fp.size() * 2, hipMemcpyHostToDevice));
      CK(hipMalloc(&dw2, w2.size() * 4)); CK(hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice));
      CK(hipMalloc(&dbias2, 256)); CK(hipMalloc(&dscale2, 256));
      CK(hipMemcpy(dbias2, bias2.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dscale2, scale2.data(), 256, hipMemcpyHostToDevice));
      a.f_w = dfw; a.f_bias = dbias2; a.f_scale = dscale2; a.f_act = 1;
    }
    unsigned long long* dbg; CK(hipMalloc(&dbg, 256)); CK(hipMemset(dbg, 0, 256)); a.dbg = dbg;
    int rc = (ver == 10 ? hcf::wino6::launch(a, ncu, 0) : launch(a, ncu, 0, ver));
    if (rc != 0) { printf("launch failed %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    if (check) {
      float* ref; CK(hipMalloc(&ref, npix * P.cout * 4));
      ref_conv<<<(unsigned)((npix * P.cout + 255) / 256), 256>>>(s0, cs0, P.n0, s1, cs1, P.n1, dw, dbias, dscale, P.act, P.res ? res : nullptr, 64,
                                                                0.2f, P.B, P.H, P.W, P.cout, ref);
      std::vector<float> hr(npix * P.cout), ho(npix * 64);
      if (P.fuse) {
        float* ref2; CK(hipMalloc(&ref2, npix * 64 * 4));
        ref_1x1<<<(unsigned)((npix * 64 + 255) / 256), 256>>>(ref, dw2, dbias2, dscale2, 1, npix, ref2);
        CK(hipMemcpy(ref, ref2, npix * 64 * 4, hipMemcpyDeviceToDevice));
        hipFree(ref2);
      }
      CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
      double md = 0, refmax = 0;
      for (long long p = 0; p < npix; ++p) for (int c = 0; c < P.cout; ++c) {
        md = fmax(md, fabs((double)hr[p * P.cout + c] - ho[p * 64 + c])); refmax = fmax(refmax, fabs(hr[p * P.cout + c])); }
      int hovf = 0; CK(hipMemcpy(&hovf, ovf, 4, hipMemcpyDeviceToHost));
      printf("%-34s max|diff| %.3e (ref max %.3f)  %s%s\n", P.name, md, refmax, md <= 4e-6 * refmax ? "OK" : "FAIL", hovf ? "  RANGE FLAG" : "");
      if (md > 1e-3 * refmax && getenv("W6_VERBOSE")) {          // where are the wrong values?
        long long nbad = 0; int shown = 0; long long bych[64] = {0};
        for (long long p = 0; p < npix; ++p) for (int c = 0; c < P.cout; ++c) {
          const double d = fabs((double)hr[p * P.cout + c] - ho[p * 64 + c]);
          if (d > 1e-3 * refmax) {
            ++nbad; ++bych[c];
            if (shown < 16) { ++shown; printf("    b %lld y %lld x %lld c %d: ref %.5f got %.5f\n", p / (P.H * P.W), (p / P.W) % P.H, p % P.W, c, hr[p * P.cout + c], ho[p * 64 + c]); }
          }
        }
        long long byab[4][4] = {{0}};
        for (long long p = 0; p < npix; ++p) for (int c = 0; c < P.cout; ++c)
          if (fabs((double)hr[p * P.cout + c] - ho[p * 64 + c]) > 1e-3 * refmax) ++byab[((p / P.W) % P.H) & 3][(p % P.W) & 3];
        printf("    by (y & 3, x & 3):");
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) printf(" %lld", byab[i][j]); printf(" |"); }
        printf("\n    %lld wrong of %lld; per channel:", nbad, npix * P.cout);
        for (int c = 0; c < P.cout; ++c) printf(" %lld", bych[c]);
        printf("\n");
      }
      hipFree(ref);
    } else {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 10;
      for (int i = 0; i < 2; ++i) (ver == 10 ? hcf::wino6::launch(a, ncu, 0) : launch(a, ncu, 0, ver));
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) (ver == 10 ? hcf::wino6::launch(a, ncu, 0) : launch(a, ncu, 0, ver));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters, fl = 2.0 * 9 * cin * P.cout * (double)npix;
      printf("%-34s %9.1f us  %7.1f TF-eq  (%.3f of 833)\n", P.name, us, fl / us / 1e6, fl / us / 1e6 / 833.3);
#if defined(W6_PROF)
      if (ver == 10) { unsigned long long h[16]; CK(hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost));
        if (h[13]) { const double life = (double)h[12];
          printf("    per wave: life %.0f kcyc (%.2f GHz) | top wait %.1f  barrier %.1f  dma issue %.1f  single rows %.1f  row0 %.1f  pair rows %.1f  row1 %.1f  row2 %.1f | exchange %.1f  out0 %.1f  out1+rest %.1f  other %.1f %%\n",
                 life / 1e3 / h[13], life / (double)h[13] / (us * 1e3 * 12), 100 * h[0] / life, 100 * h[1] / life, 100 * h[2] / life, 100 * h[3] / life, 100 * h[4] / life,
                 100 * h[5] / life, 100 * h[6] / life, 100 * h[7] / life, 100 * h[8] / life, 100 * h[9] / life, 100 * h[10] / life, 100 * h[11] / life); } }
#endif
#if defined(WINO_PROF)
      { unsigned long long h[8]; CK(hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost));
        if (h[4]) printf("    per wave: life %.0f kcyc (%.2f GHz)  vmcnt %.1f %%  barrier %.1f %%  setup+issue %.1f %%  loads+transform %.1f %%  epilogue %.1f %%\n",
                         h[2] / 1e3 / h[4], h[2] / (double)h[4] / (us * 1e3), 100.0 * h[0] / h[2], 100.0 * h[1] / h[2], 100.0 * h[5] / h[2], 100.0 * h[6] / h[2], 100.0 * h[3] / h[2]);
        if (h[4] && h[7]) printf("    of the epilogue: %.1f %% of the life waiting at its first barrier\n", 100.0 * h[7] / h[2]); }
#endif
    }
    fflush(stdout);
    hipFree(dfw); hipFree(dw2); hipFree(dbias2); hipFree(dscale2);
    hipFree(s0); hipFree(s1); hipFree(out); hipFree(res); hipFree(ovf); hipFree(wpk); hipFree(dw); hipFree(dbias); hipFree(dscale);
  }
  return 0;
}
