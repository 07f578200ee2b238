// How many independent accumulator chains does ONE wave need to issue v_mfma_f32_32x32x16_f16 at the pipe's rate (32 cycles)?
// NC accumulators are updated round-robin (each MFMA depends on the MFMA NC instructions earlier); 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_chains.hip -o build/micro/mfma_chains && build/micro/mfma_chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NC, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void kern(float* out, int iters, unsigned long long* clk, float seed) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + 0.001f * (lane + i)); b[i] = (_Float16)(0.5f - 0.002f * (lane - i)); }
  f32x16 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 24 / NC; ++k)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) atomicAdd(clk, c1 - c0);
}

template <int NC, int WPS>
static void run(float* out, unsigned long long* clk) {
  const int iters = 2000, blocks = 256;
  hipMemset(clk, 0, 8);
  kern<NC, WPS><<<blocks, 256 * WPS>>>(out, 50, clk, 0.1f);
  hipMemset(clk, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<NC, WPS><<<blocks, 256 * WPS>>>(out, iters, clk, 0.1f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  const double per_wave = (double)h / ((double)iters * 24);          // shader cycles per MFMA of one wave
  const double pf = (double)blocks * 4 * WPS * iters * 24.0 * 32768.0 / (ms * 1e-3) / 1e15;
  printf("chains %d, %d wave(s) per SIMD: %6.1f shader cycles per MFMA and wave = %5.1f per MFMA and SIMD; %6.3f ms; %5.3f PF/s\n", NC, WPS, per_wave,
         per_wave / WPS, ms, pf);
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 1>(out, clk); run<2, 1>(out, clk); run<3, 1>(out, clk); run<4, 1>(out, clk); run<8, 1>(out, clk);
    run<1, 2>(out, clk); run<2, 2>(out, clk); run<4, 2>(out, clk); run<8, 2>(out, clk);
    printf("\n");
  }
  return 0;
}
