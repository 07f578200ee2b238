// What does a grid-wide barrier cost on MI355X (8 XCDs, one L2 each)? A persistent chain of dependent stages inside ONE launch
// (a whole dense block, a whole flow step) would replace ~15 us launches by such barriers: the B = 1 latency floor question of
// profiles/r03_notes.md section 7. Every block writes a value per stage, the barrier (release fence + atomic arrive + spin +
// acquire fence) follows, then every block reads its neighbour's value of that stage and checks it.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/grid_sync.hip -o build/micro/grid_sync && build/micro/grid_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512, 1) void chain_kernel(unsigned* bar, float* data, int nstage, int payload, int* bad) {
  extern __shared__ char lds[];                  // (the dynamic LDS request keeps one block per CU, as the conv kernels)
  const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  for (int s = 0; s < nstage; ++s) {
    for (int i = t; i < payload; i += 512) data[(size_t)b * payload + i] = (float)(s * 1000 + b);   // this stage's "output tile"
    __threadfence();                             // release: visible to the other XCDs' L2s
    __syncthreads();
    if (t == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(s + 1) * nb;
      while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __threadfence();                             // acquire
    const int nbr = (b + 37) % nb;               // a block of (most likely) another XCD
    float acc = 0.f;
    for (int i = t; i < payload; i += 512) acc += __builtin_nontemporal_load(&data[(size_t)nbr * payload + i]) - (float)(s * 1000 + nbr);
    if (acc != 0.f) atomicAdd(bad, 1);
    __syncthreads();
  }
}
__global__ void tiny_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

int main() {
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  unsigned* bar; float* data; int* bad;
  const int maxpay = 16384;
  CK(hipMalloc(&bar, 4)); CK(hipMalloc(&data, (size_t)ncu * maxpay * 4)); CK(hipMalloc(&bad, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int payload : {0, 1024, 16384}) {
    for (int nstage : {10, 200}) {
      CK(hipMemset(bar, 0, 4)); CK(hipMemset(bad, 0, 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(chain_kernel, dim3(ncu), dim3(512), 150 * 1024, 0, bar, data, nstage, payload, bad);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      printf("grid barrier chain: %3d blocks, payload %5d floats per block and stage, %3d stages: %8.1f us total, %6.2f us per stage, stale reads %d\n",
             ncu, payload, nstage, ms * 1e3, ms * 1e3 / nstage, hb);
    }
  }
  // the alternative: the same number of dependent (empty) launches in one stream
  for (int n : {10, 200}) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(ncu), dim3(512), 0, 0, data);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dependent launches: %3d empty kernels of %d blocks: %8.1f us total, %6.2f us per launch\n", n, ncu, ms * 1e3, ms * 1e3 / n);
  }
  return 0;
}
