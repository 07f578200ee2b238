// Does the sustained v_mfma_f32_32x32x16_f16 rate depend on how much the operands CHANGE from one MFMA to the next?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_toggle.hip -o build/micro/mfma_toggle && build/micro/mfma_toggle
// mode 0: every MFMA of the loop body uses the same A and B fragment (no operand toggling at the matrix core inputs)
// mode 1: 16 distinct random A fragments x 8 distinct random B fragments, a different pair for consecutive MFMAs
// mode 2: as 1, but the fragments are ALSO rewritten every iteration (xor with a lane/iteration pattern: 2 VALU per MFMA)
// mode 3 / 4: only A / only B cycles; mode 5: a new (A, B) pair every 4th MFMA (the 4 accumulators share it)
// Reports TFLOP/s, the in-kernel clock (s_memtime / s_memrealtime) and busy cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const uint16_t* pat, int npat, int iters, float* out, unsigned long long* clk) {
  f16x8 a[16], b[8];
  for (int i = 0; i < 16; ++i)
    for (int e = 0; e < 8; ++e) a[i][e] = __builtin_bit_cast(_Float16, pat[(threadIdx.x * 8 + e + 131 * i) % npat]);
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) b[i][e] = __builtin_bit_cast(_Float16, pat[(threadIdx.x * 8 + e + 977 * i + 5) % npat]);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int ai = (MODE == 0 || MODE == 4) ? 0 : (MODE == 5) ? (u >> 2) : u;
      const int bi = (MODE == 0 || MODE == 3) ? 0 : (MODE == 5) ? (u >> 2) : (u & 7);
      if (MODE == 2) {   // keep the exponent field: flip mantissa bits only
        u32x4 x = __builtin_bit_cast(u32x4, a[ai]);
        x.x ^= 0x01ff01ffu & (unsigned)(it * 2654435761u + u); x.z ^= 0x00ff00ffu & (unsigned)(it * 40503u + threadIdx.x);
        a[ai] = __builtin_bit_cast(f16x8, x);
      }
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ai], b[bi], acc[u & 3], 0, 0, 0);
    }
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((blockIdx.x & 255) == 7 && threadIdx.x == 0) { atomicAdd(clk, c1 - c0); atomicAdd(clk + 1, r1 - r0); }
}

int main() {
  const int NP = 4096, blocks = 256 * 2, iters = 40000;     // 2 blocks per CU = 2 waves per SIMD, all resident at once
  uint16_t h[NP]; uint16_t* d; float* out; unsigned long long* clk;
  hipMalloc(&d, sizeof(h)); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  uint32_t s = 12345;
  for (int i = 0; i < NP; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = (uint16_t)(((s >> 31) << 15) | ((13 + ((s >> 20) % 4)) << 10) | ((s >> 8) & 0x3ff));   // ~N(0,1)-like magnitudes
  }
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[] = {"same A,B every MFMA", "16 A x 8 B fragments cycling", "cycling + mantissas rewritten",
                         "A cycles (16), B fixed", "A fixed, B cycles (8)", "A,B change every 4th MFMA"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipMemset(clk, 0, 16);
      auto launch = [&](int n) {
        if (mode == 0) k<0><<<blocks, 256>>>(d, NP, n, out, clk);
        else if (mode == 1) k<1><<<blocks, 256>>>(d, NP, n, out, clk);
        else if (mode == 2) k<2><<<blocks, 256>>>(d, NP, n, out, clk);
        else if (mode == 3) k<3><<<blocks, 256>>>(d, NP, n, out, clk);
        else if (mode == 4) k<4><<<blocks, 256>>>(d, NP, n, out, clk);
        else k<5><<<blocks, 256>>>(d, NP, n, out, clk);
      };
      launch(4000);
      hipMemset(clk, 0, 16);
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
      const double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
      const double mhz = hc[1] ? 100.0 * hc[0] / hc[1] : 0;
      // per SIMD: 2 waves x iters x 16 MFMAs in ms -> MFMA slots of 32 cycles
      const double busy = (2.0 * iters * 16 * 32) / (ms * 1e-3 * mhz * 1e6);
      printf("%-34s %8.2f ms  %7.1f TFLOP/s  clk %.0f MHz  MFMA pipe busy %.0f %%\n", names[mode], ms, flop / ms / 1e9, mhz, 100 * busy);
    }
  return 0;
}
