// How deep is a CU's LDS-DMA path? Every wave issues NP back-to-back 1-KB pieces (buffer_load_dwordx4 ... lds) and stamps s_memtime
// after each issue; then waits (vmcnt(0)) and stamps again. Sources: "hbm" = every piece a distinct 1 KB of a 4 GB buffer (misses),
// "gather" = 16 separate 64-byte runs per piece (the image pattern: one pixel's 16 channels of a 128-channel tensor), "l2" = a 1 MB region.
// Output per configuration: cycles between consecutive issues (median over CUs) for wave 0, and the time until everything landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lptr;
constexpr int NP = 32;

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const char* src, unsigned long long* stamps, int nwaves_active, long long span) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves_active) return;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7ffff000, 0x00020000);
  // piece p of (block, wave): distinct KB
  const unsigned base = (unsigned)(blockIdx.x * 8 + wave) * NP * 1024u;       // span is a power of two: offsets by masking
  unsigned long long t[NP + 2];
  t[0] = __builtin_readcyclecounter();
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    unsigned off;
    const unsigned mask = (unsigned)(span - 1);
    if (MODE == 0) off = (base + p * 1024u) & mask;                                            // contiguous KB, all distinct: HBM
    else if (MODE == 1) off = ((base + p * 1024u) * 8u) & mask;                                // 16 runs of 64 B, 512 B apart
    else off = ((unsigned)(wave * NP + p) * 1024u) & ((1u << 20) - 1);                         // L2-resident
    int vo;
    if (MODE == 1) vo = (lane >> 2) * 512 + (lane & 3) * 16; else vo = lane * 16;
    const int so = (int)(off & 0x7fffffff);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr)(lds + (wave * NP + p) % 128 * 1024), 16, vo, so, 0, 0);
    t[p + 1] = __builtin_readcyclecounter();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  t[NP + 1] = __builtin_readcyclecounter();
  if (lane == 0) {
    unsigned long long* o = stamps + ((size_t)blockIdx.x * 8 + wave) * (NP + 2);
    for (int i = 0; i < NP + 2; ++i) o[i] = t[i] - t[0];
  }
}

int main() {
  const long long span = 1LL << 30;   // 1 GB
  char* src; CK(hipMalloc(&src, (size_t)span + (1 << 20)));
  CK(hipMemset(src, 1, (size_t)span));
  unsigned long long* st; CK(hipMalloc(&st, 256 * 8 * (NP + 2) * 8));
  const char* names[3] = {"hbm contiguous", "hbm gather 16 x 64 B", "l2 resident"};
  for (int mode = 0; mode < 3; ++mode)
    for (int nw : {1, 2, 4, 8}) {
      float kms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(st, 0, 256 * 8 * (NP + 2) * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventRecord(e0));
        auto fn = mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        hipLaunchKernelGGL(fn, dim3(256), dim3(512), 128 * 1024, 0, src, st, nw, span);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&kms, e0, e1));
        CK(hipDeviceSynchronize());
      }
      printf("[kernel wall time %.1f us]\n", kms * 1e3);
      std::vector<unsigned long long> h(256 * 8 * (NP + 2));
      CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
      printf("%-22s %d wave(s)/CU issuing %d pieces each (100 MHz ticks x 10 ns):\n  issue time of piece p, wave 0 (median over CUs):", names[mode], nw, NP);
      for (int p = 0; p <= NP + 1; ++p) {
        std::vector<unsigned long long> v;
        for (int b = 0; b < 256; ++b) v.push_back(h[((size_t)b * 8 + 0) * (NP + 2) + p]);
        std::sort(v.begin(), v.end());
        if (p >= 1) printf(" %llu", v[128]);
      }
      std::vector<unsigned long long> land;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) land.push_back(h[((size_t)b * 8 + w) * (NP + 2) + NP + 1]);
      std::sort(land.begin(), land.end());
      const double med = (double)land[land.size() / 2] * 10e-9;           // seconds (100 MHz)
      printf("\n  all landed (median wave): %.2f us -> %.1f B/ns per CU = %.2f TB/s chip\n", med * 1e6, nw * NP * 1024 / (med * 1e9), nw * NP * 1024 / med * 256 / 1e12);
    }
  return 0;
}
