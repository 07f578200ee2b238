// Does v_mfma_f32_32x32x16_f16 throughput (or the sustained clock) depend on the operand values?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_subnormal.hip -o gpurun_out/mfma_subnormal && gpurun_out/mfma_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const uint16_t* pat, int npat, int iters, float* out, unsigned long long* clk) {
  f16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      uint16_t bits = pat[(threadIdx.x * 8 + e + 13 * i) % npat];
      a[i][e] = __builtin_bit_cast(_Float16, bits);
    }
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.01f * ((threadIdx.x + e + i) % 17 - 8));
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[u & 1], acc[i], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

int main() {
  const int NP = 4096, blocks = 256 * 8, iters = 20000;
  uint16_t h[NP]; uint16_t* d; float* out; unsigned long long* clk;
  hipMalloc(&d, sizeof(h)); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  const char* names[] = {"normal ~N(0,1)", "subnormal (|x|<6e-5)", "half normal / half subnormal", "zeros", "small normal 2^-12..2^-10"};
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 5; ++mode) {
    uint32_t s = 12345;
    for (int i = 0; i < NP; ++i) {
      s = s * 1664525u + 1013904223u;
      uint16_t sign = (s >> 31) << 15, man = (s >> 8) & 0x3ff;
      uint16_t expn = 13 + ((s >> 20) % 4);                       // 2^-2 .. 2^1
      uint16_t v;
      if (mode == 0) v = sign | (expn << 10) | man;
      else if (mode == 1) v = sign | man;                          // exponent 0: subnormal
      else if (mode == 2) v = (i & 1) ? (sign | man) : (sign | (expn << 10) | man);
      else if (mode == 3) v = 0;
      else v = sign | ((3 + (s >> 20) % 3) << 10) | man;
      h[i] = v;
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(d, NP, 2000, out, clk);                     // warm-up
    hipEventRecord(e0);
    k<<<blocks, 256>>>(d, NP, iters, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("%-32s %8.2f ms  %7.1f TFLOP/s  clk %.0f MHz  cycles/MFMA/wave %.1f\n", names[mode], ms, flop / ms / 1e9,
           hc[0] / (hc[1] / 100.0), (double)hc[0] / (iters * 16.0));
  }
  return 0;
}
