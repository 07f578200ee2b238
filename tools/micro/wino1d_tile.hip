// Feasibility probe (round 5): phase loop of a 1-D Winograd form -- F(2,3) along x, direct along y -- of the f16x3 convolution.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/micro/wino1d_tile.hip -o build/micro/wino1d_tile && build/micro/wino1d_tile
// Unit = 8 output rows x 64 columns x 64 output channels, 8 waves: wave w = output row w = 32 patches of 1 x 2 outputs, BOTH channel
// tiles, all four positions (8 accumulators). A "phase" = (16-channel chunk, kernel row dy): the wave reads image row w + dy of its
// patches (4 columns x 8 channels per lane: 8 ds_read_b128), forms V[nu] = B^T d (4 subtractions per channel), splits it into f16
// hi / lo and issues 4 positions x 2 tiles x 3 = 24 MFMAs against 16 weight fragments of the phase's 16 KB (ring of 4 slots).
// Per 256 output pixels and chunk: 1.5 phases = 1.5x the MFMAs of the 2-D form (tools/micro/wino_tile.hip), but 80 instead of 152
// VALU per 24 MFMAs and 30 instead of 102 LDS-DMA pieces per 192 MFMAs. DMA > 0: every phase the block also issues that many 1-KB
// buffer_load ... lds pieces (weights-like: contiguous, L2-resident source) into an unused LDS region, to price the staging.
// Data is random, results are not checked. Reports direct-equivalent TFLOP/s (2*9*Cin*Cout per output pixel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr;

__device__ __forceinline__ void split8(const float (&v)[8], u32x4& h, u32x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 hh = {(_Float16)v[2 * i], (_Float16)v[2 * i + 1]};
    h[i] = __builtin_bit_cast(uint32_t, hh);
    uint32_t lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h[i]), "v"(v[2 * i]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(h[i]), "v"(v[2 * i + 1]));
    l[i] = lo;
  }
}

constexpr int ROWS = 10, HW = 66;
constexpr int A_BYTES = ROWS * HW * 64;            // 42 240: fp32 [row][px][16 ch]
constexpr int WP_BYTES = 4 * 2 * 2 * 1024;         // one phase: [nu][tile][plane] x 1 KB = 16 KB
constexpr int NSLOT = 4;
constexpr int DMA_OFF = 2 * A_BYTES + NSLOT * WP_BYTES;   // 150 016: scratch target of the priced DMA pieces

template <int DMA>
__global__ __launch_bounds__(512, 1) void kern(const uint4* src, int nsrc, float* out, int iters, unsigned long long* clk, const char* wsrc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (2 * A_BYTES + NSLOT * WP_BYTES) / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = src[(i + 977 * blockIdx.x) % nsrc];
  f32x16 acc[4][2];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][n][r] = 0.f;
  // byte offsets of the lane's 4 patch columns (16-byte slots XOR-swizzled per 256-byte row of 4 pixels), k-half `half`
  int off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = 2 * li + j;
    off[j] = (x >> 2) * 256 + (((((x & 3) * 4 + 2 * half)) ^ ((x >> 2) & 7)) << 4);
  }
  constexpr int ROWB = ((HW + 3) / 4) * 256;         // 17 x 256 bytes per image row
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 1 << 20, 0x00020000);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int ph = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy, ++ph) {
      const char* const ib = lds + ((it & 1) ? A_BYTES : 0) + (wave + dy) * ROWB;
      const char* const wb = lds + 2 * A_BYTES + (ph & 3) * WP_BYTES + lane * 16;
      if (DMA > 0) {                                  // this phase's share of the staging: DMA pieces of 1 KB per wave-instruction
#pragma unroll
        for (int q = 0; q < (DMA + 7) / 8; ++q)
          if (8 * q + wave < DMA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr)(lds + DMA_OFF + ((8 * q + wave) & 7) * 1024), 16, lane * 16,
                                                     ((ph * 37 + 8 * q + wave) & 511) * 1024, 0, 0);
      }
      float d[4][8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(ib + off[j]);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(ib + (off[j] ^ 16));
#pragma unroll
        for (int k = 0; k < 4; ++k) { d[j][k] = x0[k]; d[j][4 + k] = x1[k]; }
      }
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          v[k] = (nu == 0) ? d[0][k] - d[2][k] : (nu == 1) ? d[1][k] + d[2][k] : (nu == 2) ? d[1][k] - d[2][k] : d[1][k] - d[3][k];
        u32x4 vh, vl;
        split8(v, vh, vl);
        const f16x8 w00 = *reinterpret_cast<const f16x8*>(wb + (nu * 4 + 0) * 1024);
        const f16x8 w01 = *reinterpret_cast<const f16x8*>(wb + (nu * 4 + 1) * 1024);
        const f16x8 w10 = *reinterpret_cast<const f16x8*>(wb + (nu * 4 + 2) * 1024);
        const f16x8 w11 = *reinterpret_cast<const f16x8*>(wb + (nu * 4 + 3) * 1024);
        acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w00, __builtin_bit_cast(f16x8, vh), acc[nu][0], 0, 0, 0);
        acc[nu][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w10, __builtin_bit_cast(f16x8, vh), acc[nu][1], 0, 0, 0);
        acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w01, __builtin_bit_cast(f16x8, vh), acc[nu][0], 0, 0, 0);
        acc[nu][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w11, __builtin_bit_cast(f16x8, vh), acc[nu][1], 0, 0, 0);
        acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w00, __builtin_bit_cast(f16x8, vl), acc[nu][0], 0, 0, 0);
        acc[nu][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w10, __builtin_bit_cast(f16x8, vl), acc[nu][1], 0, 0, 0);
      }
      if (DMA > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[p][n][r];
  out[(size_t)blockIdx.x * 512 + tid] = s;
  if ((blockIdx.x & 255) == 7 && tid == 0) atomicAdd(clk, t1 - t0);
}

static uint4* g_src; static int g_nsrc; static float* g_out; static unsigned long long* g_clk; static char* g_w;

template <int DMA>
static void run(const char* name) {
  const int dyn = 160 * 1024;
  auto fn = kern<DMA>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)fn);
  const int blocks = 256 * 2, iters = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  fn<<<blocks, 512, dyn>>>(g_src, g_nsrc, g_out, 20, g_clk, g_w);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(g_clk, 0, 8);
    hipEventRecord(e0);
    fn<<<blocks, 512, dyn>>>(g_src, g_nsrc, g_out, iters, g_clk, g_w);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // per block and chunk (iteration): 512 output pixels x 64 channels x 16 input channels x 9 taps x 2
  const double eq = (double)blocks * iters * 512.0 * 64 * 16 * 18 / (best * 1e-3) / 1e12;
  const double mfma = (double)blocks * 8 * iters * 72.0 * 32768.0 / (best * 1e-3) / 1e15;
  printf("%-40s vgpr %3d  %7.3f ms  %5.3f PF/s executed  %6.1f TF-eq (%4.2f of 833)\n", name, fa.numRegs, best, mfma, eq, eq / 833.3);
  fflush(stdout);
}

int main() {
  g_nsrc = 1 << 16;
  std::vector<float> h((size_t)g_nsrc * 4);
  uint32_t s = 12345;
  for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 8388608.0f - 1.0f); }
  hipMalloc(&g_src, h.size() * 4); hipMemcpy(g_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&g_w, 1 << 20); hipMemcpy(g_w, h.data(), 1 << 20, hipMemcpyHostToDevice);
  hipMalloc(&g_out, (size_t)512 * 512 * 4); hipMalloc(&g_clk, 8);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("1-D phase loop, LDS resident");
    run<16>("  + 16 DMA pieces per phase (weights only)");
    run<30>("  + 30 DMA pieces per phase (weights + image)");
    run<48>("  + 48 DMA pieces per phase");
    printf("\n");
  }
  return 0;
}
