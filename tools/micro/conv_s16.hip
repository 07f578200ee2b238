// Stand-alone development harness for the split16 / LDS-DMA 3x3 convolution (tools/micro/hcf_conv_s16.h; build with -I tools/micro -I hcflow_amd/csrc):
// builds the kernel header against synthetic RDB-shaped problems, checks it against a naive fp64 evaluation of the
// same split operands, and times the RDB shapes of config 2 (B = 16, 320^2 / 160^2).
//   hipcc -O3 --offload-arch=gfx950 -I hcflow_amd/csrc tools/micro/conv_s16.hip -o build/micro/conv_s16 && build/micro/conv_s16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "hcf_conv_s16.h"

using namespace hcf::s16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// deterministic pseudo-random fp32 in [-2, 2)
__host__ __device__ inline float hashf(uint64_t i, uint32_t salt) {
  uint64_t x = i * 0x9E3779B97F4A7C15ull + salt * 0xBF58476D1CE4E5B9ull;
  x ^= x >> 31; x *= 0x94D049BB133111EBull; x ^= x >> 29;
  return ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * 2.0f;
}

// fill a split16p slab: [npix][nrec] records from the fp32 value hashf(pix * C + ch)
__device__ inline size_t rec_off(long long p, int g, int nrec, int hw) {      // planar: [b][g][y][x][64 B]
  const long long b = p / hw, q = p - b * hw;
  return ((size_t)(b * nrec + g) * hw + q) * 64;
}
__global__ void fill_s16(char* slab, long long npix, int nrec, uint32_t salt, int hw) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * nrec) return;
  const long long p = i / nrec; const int g = (int)(i - p * nrec);
  _Float16* rec = reinterpret_cast<_Float16*>(slab + rec_off(p, g, nrec, hw));
  for (int h = 0; h < 2; ++h)
    for (int e = 0; e < 8; ++e) {
      const int ch = 16 * g + rec_channel(h, e);
      const float v = hashf((uint64_t)p * (nrec * 16) + ch, salt);
      const _Float16 hi = (_Float16)v;
      rec[h * 8 + e] = hi;
      rec[16 + h * 8 + e] = (_Float16)(v - (float)hi);
    }
}
__global__ void fill_f32(float* p, long long n, uint32_t salt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = hashf((uint64_t)i, salt);
}

// naive reference on the SAME split operands (value = hi + lo), fp64 accumulation, one thread per output element
__global__ void ref_conv(const char* slab, int nrec, int rec0, int cin, const float* w /*[cout][cin][9]*/, const float* bias,
                         const float* scale, int act, const float* res1, int res1_cs, float rs1, int B, int H, int W, int cout,
                         float* out /*[B*H*W][cout]*/) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H * W * cout) return;
  const int oc = (int)(i % cout); const long long pix = i / cout;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
  double s = 0;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = y + dy - 1, xx = x + dx - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      for (int ic = 0; ic < cin; ++ic) {
        const _Float16* rec = reinterpret_cast<const _Float16*>(slab + rec_off(((long long)b * H + yy) * W + xx, rec0 + (ic >> 4), nrec, H * W));
        int pos = -1;
        for (int h = 0; h < 2; ++h) for (int e = 0; e < 8; ++e) if (rec_channel(h, e) == (ic & 15)) pos = h * 8 + e;
        const double a = (double)(float)rec[pos] + (double)(float)rec[16 + pos];
        s += a * (double)w[((size_t)oc * cin + ic) * 9 + dy * 3 + dx];
      }
    }
  float v = (float)((s + (double)bias[oc]) * (double)scale[oc]);
  if (act == 1) v = v > 0 ? v : 0; else if (act == 2) v = v > 0 ? v : 0.2f * v;
  if (res1) v = v * rs1 + res1[pix * res1_cs + oc];
  out[i] = v;
}

// decode a split16p output window back to fp32 [pix][cout]
__global__ void decode_s16(const char* slab, int nrec, int rec0, long long npix, int cout, float* out, int hw) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * cout) return;
  const int oc = (int)(i % cout); const long long p = i / cout;
  const _Float16* rec = reinterpret_cast<const _Float16*>(slab + rec_off(p, rec0 + (oc >> 4), nrec, hw));
  int pos = -1;
  for (int h = 0; h < 2; ++h) for (int e = 0; e < 8; ++e) if (rec_channel(h, e) == (oc & 15)) pos = h * 8 + e;
  out[i] = (float)rec[pos] + (float)rec[16 + pos];
}

struct Prob { const char* name; int B, H, W, cin, cout, act; bool res, o16, o32; };

int main(int argc, char** argv) {
  int only = argc > 1 ? atoi(argv[1]) : -1;
  int variant = argc > 2 ? atoi(argv[2]) : 0;
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  const Prob probs[] = {
      {"check small 64->32  lrelu  s16 out", 2, 24, 40, 64, 32, 2, false, true, false},
      {"check small 192->64 res    f32+s16", 1, 17, 70, 192, 64, 0, true, true, true},
      {"check small 96->32  relu   f32 out", 1, 8, 32, 96, 32, 1, false, false, true},
      {"check small 64->64  lrelu  f32+s16", 2, 40, 70, 64, 64, 2, false, true, true},
      {"rdb conv1 L0  64->32  @320", 16, 320, 320, 64, 32, 2, false, true, false},
      {"rdb conv2 L0  96->32  @320", 16, 320, 320, 96, 32, 2, false, true, false},
      {"rdb conv4 L0 160->32  @320", 16, 320, 320, 160, 32, 2, false, true, false},
      {"rdb conv5 L0 192->64  @320", 16, 320, 320, 192, 64, 0, true, true, true},
      {"rdb conv4 L0 160->32  @320 B=2", 2, 320, 320, 160, 32, 2, false, true, false},
      {"rdb conv4 L0 160->32  @320 B=4", 4, 320, 320, 160, 32, 2, false, true, false},
      {"rdb conv4 L0 160->32  @320 B=8", 8, 320, 320, 160, 32, 2, false, true, false},
      {"rdb conv1 L1  64->32  @160", 16, 160, 160, 64, 32, 2, false, true, false},
      {"rdb conv5 L1 192->64  @160", 16, 160, 160, 192, 64, 0, true, true, true},
  };
  const int nprob = sizeof(probs) / sizeof(probs[0]);
  for (int pi = 0; pi < nprob; ++pi) {
    if (only >= 0 && pi != only) continue;
    const Prob& P = probs[pi];
    const bool check = P.B * P.H * P.W <= 8192;
    const long long npix = (long long)P.B * P.H * P.W;
    const int nrec_src = 12, cs_bytes = nrec_src * 64;              // 192-channel slab
    const int nchunk = P.cin / 16, ntn = P.cout / 32;
    char *src, *out16; float *out32, *res, *dw, *dbias, *dscale; char* wpk; int* ovf; char* zeros;
    CK(hipMalloc(&src, (size_t)npix * cs_bytes + 4096)); CK(hipMalloc(&out16, (size_t)npix * cs_bytes + 4096));
    CK(hipMalloc(&out32, (size_t)npix * 64 * 4)); CK(hipMalloc(&res, (size_t)npix * 64 * 4));
    CK(hipMalloc(&ovf, 256)); CK(hipMemset(ovf, 0, 256)); zeros = reinterpret_cast<char*>(ovf) + 64;
    fill_s16<<<(unsigned)((npix * nrec_src + 255) / 256), 256>>>(src, npix, nrec_src, 7, P.H * P.W);
    fill_f32<<<(unsigned)((npix * 64 + 255) / 256), 256>>>(res, npix * 64, 11);
    CK(hipMemset(out16, 0, (size_t)npix * cs_bytes)); CK(hipMemset(out32, 0, (size_t)npix * 64 * 4));
    std::vector<float> w((size_t)P.cout * P.cin * 9), bias(64, 0.f), scale(64, 1.f);
    for (size_t i = 0; i < w.size(); ++i) w[i] = hashf(i, 3) * 0.5f / sqrtf((float)P.cin * 9.f);
    for (int i = 0; i < P.cout; ++i) { bias[i] = hashf(i, 5) * 0.1f; scale[i] = 1.f + 0.1f * hashf(i, 6); }
    std::vector<uint16_t> pk;
    const bool wide = (variant & 1) && P.cout == 64;
    if (!pack_weights_s16(w.data(), P.cin, P.cout, pk, wide)) { printf("pack failed\n"); return 1; }
    CK(hipMalloc(&wpk, pk.size() * 2)); CK(hipMemcpy(wpk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbias, 256)); CK(hipMalloc(&dscale, 256));
    CK(hipMemcpy(dbias, bias.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dscale, scale.data(), 256, hipMemcpyHostToDevice));

    Args a; memset(&a, 0, sizeof(a));
    a.src = src; a.src_planes = nrec_src; a.src_rec0 = 0; a.nchunk = nchunk; a.wpack = wpk; a.ntile_n = ntn;
    a.bias = dbias; a.scale = dscale; a.act = P.act; a.B = P.B; a.H = P.H; a.W = P.W; a.ovf = ovf; a.zeros = zeros;
    if (P.o16) { a.out16 = out16; a.out16_planes = nrec_src; a.out16_rec0 = 4; }
    if (P.o32) { a.out32 = out32; a.out32_cs = 64; a.out32_c0 = 0; }
    if (P.res) { a.res1 = res; a.res1_cs = 64; a.res1_c0 = 0; a.rs1 = 0.2f; }
    a.variant = variant;
    unsigned long long* dbg; CK(hipMalloc(&dbg, 64)); CK(hipMemset(dbg, 0, 64)); a.dbg = dbg;
    auto launch_any = [&](const Args& aa) { return wide ? launch_wide(aa, ncu, 0) : launch(aa, ncu, 0); };
    int rc = launch_any(a);
    if (rc != 0) { printf("launch failed %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    if (check) {
      float *ref, *got; CK(hipMalloc(&ref, npix * P.cout * 4)); CK(hipMalloc(&got, npix * P.cout * 4));
      ref_conv<<<(unsigned)((npix * P.cout + 255) / 256), 256>>>(src, nrec_src, 0, P.cin, dw, dbias, dscale, P.act, P.res ? res : nullptr, 64,
                                                                0.2f, P.B, P.H, P.W, P.cout, ref);
      std::vector<float> hr(npix * P.cout), hg(npix * P.cout);
      CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
      double refmax = 0; for (float v : hr) refmax = fmax(refmax, fabs(v));
      if (P.o16) {
        decode_s16<<<(unsigned)((npix * P.cout + 255) / 256), 256>>>(out16, nrec_src, 4, npix, P.cout, got, P.H * P.W);
        CK(hipMemcpy(hg.data(), got, hg.size() * 4, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < hr.size(); ++i) md = fmax(md, fabs((double)hr[i] - hg[i]));
        printf("%-40s s16 out: max|diff| %.3e (ref max %.3f)  %s\n", P.name, md, refmax, md <= 4e-6 * refmax ? "OK" : "FAIL");
      }
      if (P.o32) {
        std::vector<float> ho(npix * 64);
        CK(hipMemcpy(ho.data(), out32, ho.size() * 4, hipMemcpyDeviceToHost));
        double md = 0;
        for (long long p = 0; p < npix; ++p) for (int c = 0; c < P.cout; ++c) md = fmax(md, fabs((double)hr[p * P.cout + c] - ho[p * 64 + c]));
        printf("%-40s f32 out: max|diff| %.3e (ref max %.3f)  %s\n", P.name, md, refmax, md <= 2e-6 * refmax ? "OK" : "FAIL");
      }
      int hovf = 0; CK(hipMemcpy(&hovf, ovf, 4, hipMemcpyDeviceToHost));
      if (hovf) printf("  range flag raised!\n");
      hipFree(ref); hipFree(got);
    } else {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 10;
      for (int i = 0; i < 2; ++i) launch_any(a);
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) launch_any(a);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters, fl = 2.0 * 9 * P.cin * P.cout * (double)npix;
      printf("%-40s %9.1f us  %7.1f TF-eq  (%.3f of 833)\n", P.name, us, fl / us / 1e6, fl / us / 1e6 / 833.3);
#if defined(S16_PROF)
      { unsigned long long h[5]; CK(hipMemcpy(h, dbg, 40, hipMemcpyDeviceToHost));
        if (h[4]) printf("    per wave: life %.0f kcyc  vmcnt wait %.1f %%  barrier wait %.1f %%  epilogue %.1f %%\n", h[2] / 1e3 / h[4],
                         100.0 * h[0] / h[2], 100.0 * h[1] / h[2], 100.0 * h[3] / h[2]); }
#endif
    }
    fflush(stdout);
    hipFree(src); hipFree(out16); hipFree(out32); hipFree(res); hipFree(ovf); hipFree(wpk); hipFree(dw); hipFree(dbias); hipFree(dscale);
  }
  return 0;
}
