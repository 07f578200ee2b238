// Feasibility probe: K loop of a Winograd F(2x2, 3x3) formulation of the f16x3 convolution, operands resident in LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/wino_tile.hip -o build/micro/wino_tile && build/micro/wino_tile
// Per 16-channel chunk a wave (transform row xi, 32 tiles = 4 output rows x 32 px... see below) reads the 8 fp32 pixels
// (2 rows x 4 columns of each 4x4 patch) x 8 channels its four positions (xi, nu = 0..3) need, forms V = B^T d B (8 adds per
// channel), splits V into f16 hi / lo, and issues 12 MFMAs (4 positions x 3 terms) against transformed, pre-split weights.
// 2.25x fewer MFMAs than the direct form; the question is whether VALU + LDS traffic leave that advantage alive on a part that
// is power-limited under dense MFMA. Reports direct-equivalent TFLOP/s (2*9*Cin*Cout per output pixel).
// Data is random and the LDS addressing only mimics the access pattern (conflict-free swizzle), results are not checked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// block: NW waves = 4 transform rows x TG tile groups; tile group g = output rows [4 g, 4 g + 4) x 32 columns (32 tiles of 2x2)
template <int TG, int OCC, int NT>
__global__ __launch_bounds__(256 * TG, OCC) void kern(const uint4* src, int nsrc, float* out, int iters, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int ROWS = 4 * TG + 2, HW = 34;
  constexpr int A_BYTES = ROWS * HW * 64;                    // fp32 [row][px][16 ch]
  constexpr int W_BYTES = 16 * 2 * 2 * 32 * NT * 16;         // [pos][plane][k-half][32 NT oc][8 halves]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
  const int xi = wave & 3, tg = wave >> 2;
  for (int i = tid; i < (A_BYTES + W_BYTES) / 16; i += 256 * TG) reinterpret_cast<uint4*>(lds)[i] = src[(i + 977 * blockIdx.x) % nsrc];
  char* const ldsW = lds + A_BYTES;
  f32x16 acc[4][NT];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][n][r] = 0.f;
  // patch rows used by transform row xi: B^T rows (1,0,-1,0), (0,1,1,0), (0,-1,1,0), (0,1,0,-1)
  const int r1 = (xi == 0) ? 0 : 1, r2 = (xi == 3) ? 3 : 2;
  // signs folded into the (host-transformed) weights: t_j = d[r1][j] + sigma d[r2][j], sigma = +1 for xi = 1, -1 otherwise
  const float sigma = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (xi == 1) ? 1.f : -1.f)));
  const int trow = li >> 4, tcol = li & 15;
  // byte offsets of the 8 pixels x 2 x 16-byte pieces (channels 8 half .. 8 half + 7); 16-byte slots XOR-swizzled per 256-byte row
  int off[2][4];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = 4 * tg + 2 * trow + (rr ? r2 : r1), x = 2 * tcol + j;
      const int px = y * HW + x;
      const int slot = (px & 3) * 4 + half * 2;                 // 16-byte slot inside the 256-byte bank row
      off[rr][j] = (px >> 2) * 256 + ((slot ^ (((px >> 2) & 7) << 1)) << 4);
    }
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_setprio(1);
    // raw pixels: d[rr][j] = 8 channels
    f32x4 d[2][4][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d[rr][j][0] = *reinterpret_cast<const f32x4*>(lds + off[rr][j]);
        d[rr][j][1] = *reinterpret_cast<const f32x4*>(lds + (off[rr][j] ^ 16));
      }
    // t_j = s1 d[r1][j] + s2 d[r2][j];  V_0 = t0 - t2, V_1 = t1 + t2, V_2 = t2 - t1, V_3 = t1 - t3
    f16x8 vh[4], vl[4];
    {
      float t[4][8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) t[j][c] = __builtin_fmaf(sigma, d[1][j][c >> 2][c & 3], d[0][j][c >> 2][c & 3]);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float v[4] = {t[0][c] - t[2][c], t[1][c] + t[2][c], t[1][c] - t[2][c], t[1][c] - t[3][c]};   // (V_2 negated: sign in the weights)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const _Float16 h = (_Float16)v[p];
          vh[p][c] = h;
          vl[p][c] = (_Float16)(v[p] - (float)h);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const char* wp = ldsW + (((xi * 4 + p) * 2) * 2 + half) * (32 * NT * 16) + (n * 32 + li) * 16;
        const f16x8 w1 = *reinterpret_cast<const f16x8*>(wp);
        const f16x8 w2 = *reinterpret_cast<const f16x8*>(wp + 2 * (32 * NT * 16));
        acc[p][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, vh[p], acc[p][n], 0, 0, 0);
        acc[p][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, vh[p], acc[p][n], 0, 0, 0);
        acc[p][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, vl[p], acc[p][n], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[p][n][r];
  out[(size_t)blockIdx.x * 256 * TG + tid] = s;
  if ((blockIdx.x & 255) == 7 && tid == 0) { atomicAdd(clk, c1 - c0); atomicAdd(clk + 1, t1 - t0); }
}

static uint4* g_src; static int g_nsrc; static float* g_out; static unsigned long long* g_clk;

template <int TG, int OCC, int NT>
static void run(const char* name, int bpc) {
  constexpr int ROWS = 4 * TG + 2;
  const int need = ROWS * 34 * 64 + 16 * 2 * 2 * 32 * NT * 16;
  int dyn = (160 * 1024) / bpc - 1024;
  if (dyn < need) { printf("%-44s needs %d B of LDS\n", name, need); return; }
  auto fn = kern<TG, OCC, NT>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)fn);
  const int rounds = 2, blocks = 256 * bpc * rounds, iters = 800;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  fn<<<blocks, 256 * TG, dyn>>>(g_src, g_nsrc, g_out, 40, g_clk);
  float best = 1e30f; double mhz = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(g_clk, 0, 16);
    hipEventRecord(e0);
    fn<<<blocks, 256 * TG, dyn>>>(g_src, g_nsrc, g_out, iters, g_clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; hipMemcpy(hc, g_clk, 16, hipMemcpyDeviceToHost);
    if (ms < best) { best = ms; mhz = hc[1] ? 100.0 * hc[0] / hc[1] : 0; }
  }
  const double mfma = (double)blocks * 4 * TG * iters * 12.0 * NT;
  const double pf = mfma * 32768.0 / (best * 1e-3) / 1e15;
  // direct-equivalent flops: block chunk = 128 TG output pixels x 32 NT channels x 16 input channels x 9 taps x 2
  const double eq = (double)blocks * iters * 128.0 * TG * 32 * NT * 16 * 18 / (best * 1e-3) / 1e12;
  const double busy = (mfma / 1024.0) * 32.0 / (best * 1e-3 * mhz * 1e6);
  printf("%-44s vgpr %3d  %7.3f ms  %5.3f PF/s exec  %6.1f TF-eq (%4.2f of 833)  clk %4.0f MHz  mfma busy %3.0f %%  lds %d\n", name,
         fa.numRegs, best, pf, eq, eq / 833.3, mhz, 100 * busy, need);
  fflush(stdout);
}

int main() {
  g_nsrc = 1 << 16;
  std::vector<float> h((size_t)g_nsrc * 4);
  uint32_t s = 12345;
  for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 8388608.0f - 1.0f); }
  hipMalloc(&g_src, h.size() * 4); hipMemcpy(g_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&g_out, (size_t)256 * 8 * 2 * 512 * 4); hipMalloc(&g_clk, 16);
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 3, 1>("4 waves, 4x32 px x 32 ch, 3 blocks/CU", 3);
    run<1, 2, 1>("4 waves, 4x32 px x 32 ch, 2 blocks/CU", 2);
    run<2, 2, 1>("8 waves, 8x32 px x 32 ch, 1 block/CU", 1);
    run<2, 4, 1>("8 waves, 8x32 px x 32 ch, 2 blocks/CU", 2);
    run<1, 2, 2>("4 waves, 4x32 px x 64 ch, 2 blocks/CU", 2);
    run<2, 2, 2>("8 waves, 8x32 px x 64 ch, 1 block/CU", 1);
    printf("\n");
  }
  return 0;
}
