// Timing harness of the persistent FCN conv1 + conv2 kernel (hcflow_amd/csrc/hcf_conv_fcn.hip) on the flow-step shapes of
// config 2 (B = 16, 320^2 / 160^2, z1 = 6 / 12 channels), with compile-time ablations (results are garbage: random packs).
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -I hcflow_amd/csrc [-DFCN12_ABL=bits] tools/micro/fcn12_bench.hip -o build/micro/fcn12_bench
//   bits: 1 no output stores, 2 conv2 weights from LDS instead of L1/L2, 8 halo loads from the zero page
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "hcf_conv_fcn.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace hcf;
__global__ void fill(float* p, long long n, unsigned salt, float amp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { unsigned x = (unsigned)i * 2654435761u + salt; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = ((x & 0xffff) / 32768.f - 1.f) * amp; }
}
__global__ void fillh(_Float16* p, long long n, unsigned salt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { unsigned x = (unsigned)i * 2654435761u + salt; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = (_Float16)(((x & 0xffff) / 32768.f - 1.f) * 8.f); }
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  struct P { const char* name; int B, H, W, C, n, pre; } probs[] = {
      {"L0 main  z1=6  @320", 16, 320, 320, 12, 6, 0}, {"L0 cond  z1=3  @320 + pre", 16, 320, 320, 8, 3, 1},
      {"L1 main  z1=12 @160", 16, 160, 160, 24, 12, 0}, {"L1 cond  z1=6  @160 + pre", 16, 160, 160, 12, 6, 1}};
  for (const P& p : probs) {
    const long long npx = (long long)p.B * p.H * p.W;
    float *z, *out, *pre, *bias, *zeros; _Float16 *w1, *w2; int* ovf;
    CK(hipMalloc(&z, npx * p.C * 4)); CK(hipMalloc(&out, npx * 64 * 4)); CK(hipMalloc(&pre, npx * 64 * 4));
    CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&zeros, 4096)); CK(hipMalloc(&w1, 2 * 36864)); CK(hipMalloc(&w2, 65536)); CK(hipMalloc(&ovf, 256));
    CK(hipMemset(zeros, 0, 4096)); CK(hipMemset(ovf, 0, 256));
    fill<<<(unsigned)((npx * p.C + 255) / 256), 256>>>(z, npx * p.C, 1, 1.f);
    fill<<<(unsigned)((npx * 64 + 255) / 256), 256>>>(pre, npx * 64, 2, 1.f);
    fill<<<4, 256>>>(bias, 1024, 3, 0.5f);
    fillh<<<(36864 + 255) / 256, 256>>>(w1, 36864, 4);
    fillh<<<128, 256>>>(w2, 32768, 5);
    ConvArgs a; memset(&a, 0, sizeof(a));
    a.src[0] = mkview(z, p.C, 0, p.n); a.src[1] = a.src[2] = a.src[0]; a.nsrc = 1; a.B = p.B; a.H = p.H; a.W = p.W;
    a.wpack = (const float*)w1; a.nchunk = 1; a.bias = bias; a.scale = bias + 64; a.act = ACT_RELU;
    a.w2 = (const float*)w2; a.bias2 = bias + 128; a.scale2 = bias + 192; a.act2 = ACT_RELU;
    a.out = mkview(out, 64, 0, 64); a.ovf = ovf; a.zeros = zeros;
    a.res1 = p.pre ? mkview(pre, 64, 0, 64) : mkview(nullptr, 0, 0, 0);
    a.res2 = mkview(nullptr, 0, 0, 0);
    int rc = launch_fcn12(a, 0);
    if (rc) { printf("launch rc %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_fcn12(a, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("ABL %2d  %-28s %8.1f us\n", FCN12_ABL, p.name, ms * 1e3 / iters);
    hipFree(z); hipFree(out); hipFree(pre); hipFree(bias); hipFree(zeros); hipFree(w1); hipFree(w2); hipFree(ovf);
  }
  return 0;
}
