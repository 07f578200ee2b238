// What does a grid-wide barrier cost on MI355X (8 XCDs, one L2 each), and would a persistent chain of dependent stages inside ONE
// launch (a dense block's five convs, a flow step) beat the same stages as dependent launches? Round 4 redo of the round-3 probe,
// which measured the WRONG barrier (one counter, every block polling it with acquire loads, __threadfence on all 512 threads:
// 69-71 us per stage and stale reads). Forms measured here:
//   naive     the round-3 form, kept for reference
//   counter   one monotonic counter; lane 0: release fence, asm vmcnt(0), relaxed arrive; relaxed sc1 poll + s_sleep; ONE acquire
//             fence after the match (MI355X_MICROARCH.md "barrier-counter")
//   xcd       XCD-hierarchical ("barrier-xcd"): per-XCC arrival counter; the last arriver of an XCC is its leader: release fence ->
//             top counter -> polls the top counter for all XCCs -> acquire fence -> bumps its XCC's generation word; everybody else
//             polls its XCC's generation (relaxed) and then issues one agent-scope acquire fence
// Visibility check: every block writes `payload` floats per stage with plain stores and, after the barrier, re-reads the record of
// a block on ANOTHER XCD -- every word, L1-warm (it read the same addresses one stage earlier), under uneven load (odd blocks
// spin a few us before arriving).
// Chain-vs-launches: each stage does `work` iterations of dependent FMAs per thread (a stand-in for a conv unit: ~5 / 15 / 60 us)
// and writes a 16 KB record; the same stage body runs (a) as N dependent launches of 256 blocks and (b) as ONE persistent launch
// with the xcd barrier between stages.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/grid_sync.hip -o build/micro/grid_sync && build/micro/grid_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Bar {
  unsigned top;            // arrivals of XCC leaders (monotonic)
  unsigned pad0[31];
  unsigned xcc_cnt[8 * 32];    // per-XCC arrival counters, one 128-byte line each
  unsigned xcc_gen[8 * 32];    // per-XCC generation words
  unsigned naive;          // the round-3 single counter
  unsigned pad1[31];
  unsigned counter;        // "barrier-counter"
  unsigned pad2[31];
  unsigned timeout;        // a spin gave up
  unsigned pad3[31];
  unsigned live[8];        // in-kernel census: blocks of THIS launch per XCC
  unsigned live_total;
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

// bounded relaxed poll until *p >= want
__device__ __forceinline__ bool poll_ge(gu32* p, unsigned want, gu32* tmo) {
  for (unsigned spins = 0; __hip_atomic_load(p, RLX_AGENT) < want; ++spins) {
    __builtin_amdgcn_s_sleep(1);
    if (spins > (1u << 22)) { __hip_atomic_store(tmo, 1u, RLX_AGENT); return false; }
  }
  return true;
}

// Census of THIS launch (placement is observed, never assumed): blocks per XCC, then a one-off wait until every block has
// registered. Returns this block's XCC population in nper[8] (registers).
__device__ __forceinline__ void census(Bar* bar, int nb, unsigned xcc, int t, unsigned (&nper)[8]) {
  __shared__ unsigned sh[8];
  if (t == 0) {
    __hip_atomic_fetch_add((gu32*)&bar->live[xcc], 1u, RLX_AGENT);
    __hip_atomic_fetch_add((gu32*)&bar->live_total, 1u, RLX_AGENT);
    poll_ge((gu32*)&bar->live_total, (unsigned)nb, (gu32*)&bar->timeout);
    for (int x = 0; x < 8; ++x) sh[x] = __hip_atomic_load((gu32*)&bar->live[x], RLX_AGENT);
  }
  __syncthreads();
  for (int x = 0; x < 8; ++x) nper[x] = sh[x];
}

// form 0: round-3 naive. form 1: counter. form 2: xcd-hierarchical. `epoch` = 1, 2, ... (one per barrier of the launch).
template <int FORM>
__device__ __forceinline__ void grid_barrier(Bar* bar, unsigned epoch, int nb, const unsigned (&nper)[8], unsigned xcc, int t) {
  gu32* const tmo = (gu32*)&bar->timeout;
  if (FORM == 0) {
    __threadfence();
    __syncthreads();
    if (t == 0) {
      __hip_atomic_fetch_add(&bar->naive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = epoch * (unsigned)nb;
      unsigned spins = 0;
      while (__hip_atomic_load(&bar->naive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __threadfence();
    return;
  }
  // every storing wave drains its own stores (they are in this XCD's L2 then), the block meets, ONE lane arrives
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    if (FORM == 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // buffer_wbl2 sc1: this XCD's dirty lines reach memory
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the compiler may drop the wait behind the write-back)
      __hip_atomic_fetch_add((gu32*)&bar->counter, 1u, RLX_AGENT);
      poll_ge((gu32*)&bar->counter, epoch * (unsigned)nb, tmo);
    } else {
      gu32* const cnt = (gu32*)&bar->xcc_cnt[xcc * 32];
      gu32* const gen = (gu32*)&bar->xcc_gen[xcc * 32];
      const unsigned old = __hip_atomic_fetch_add(cnt, 1u, RLX_AGENT);
      if (old + 1 == epoch * nper[xcc]) {                        // last arriver of this XCC: its leader for this epoch
        // every block of this XCC has drained its stores into the shared L2 before arriving: ONE write-back publishes them all
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add((gu32*)&bar->top, 1u, RLX_AGENT);
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += nper[x] ? 1u : 0u;
        poll_ge((gu32*)&bar->top, epoch * nx, tmo);
        __hip_atomic_store(gen, epoch, RLX_AGENT);
      } else {
        poll_ge(gen, epoch, tmo);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // buffer_inv sc1: drop this CU's stale L1 lines
  }
  __syncthreads();
}

template <int FORM>
__global__ __launch_bounds__(512, 1) void chain_kernel(Bar* bar, float* data, int nstage, int payload,
                                                       int work, int uneven, int* bad, float* sink) {
  extern __shared__ char lds[];                  // (the dynamic LDS request keeps one block per CU, as the conv kernels)
  const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  const unsigned xcc = xcc_id();
  unsigned nper[8];
  census(bar, nb, xcc, t, nper);
  const int nbr = (b + 37) % nb;                 // a block on another XCD (blocks are dealt round-robin over the XCCs: 37 % 8 = 5)
  float acc0 = 0.f;
  for (int s = 0; s < nstage; ++s) {
    float v = (float)(s * 1000 + b);
    float w = 1.0f + 1e-7f * t;
    for (int i = 0; i < work; ++i) w = fmaf(w, 1.0000001f, 1e-9f);          // the stage's "compute"
    if (uneven && (b & 1)) for (int i = 0; i < 4000; ++i) w = fmaf(w, 1.0000001f, 1e-9f);   // ~ a few us of arrival skew
    acc0 += w * 0.f;
    for (int i = t; i < payload; i += 512) data[(size_t)b * payload + i] = v;               // this stage's "output tile"
    grid_barrier<FORM>(bar, (unsigned)(2 * s + 1), nb, nper, xcc, t);
    float acc = 0.f;
    for (int i = t; i < payload; i += 512) acc += fabsf(data[(size_t)nbr * payload + i] - (float)(s * 1000 + nbr));   // every word
    if (acc != 0.f) atomicAdd(bad, 1);
    // the consumer reads must be over before the producer overwrites them in the next stage: a second barrier (as a real chain's
    // double buffering would avoid; costed separately by `nstage` x 2 barriers here)
    grid_barrier<FORM>(bar, (unsigned)(2 * s + 2), nb, nper, xcc, t);
  }
  if (acc0 == 123.f) sink[0] = acc0;
}

// the same stage body as its own launch
__global__ __launch_bounds__(512, 1) void stage_kernel(float* data, int s, int payload, int work, int* bad, float* sink, int check) {
  extern __shared__ char lds[];
  const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  const int nbr = (b + 37) % nb;
  if (check && s > 0) {
    float acc = 0.f;
    for (int i = t; i < payload; i += 512) acc += fabsf(data[(size_t)((s - 1) & 1) * nb * payload + (size_t)nbr * payload + i] - (float)((s - 1) * 1000 + nbr));
    if (acc != 0.f) atomicAdd(bad, 1);
  }
  float w = 1.0f + 1e-7f * t;
  for (int i = 0; i < work; ++i) w = fmaf(w, 1.0000001f, 1e-9f);
  float v = (float)(s * 1000 + b);
  for (int i = t; i < payload; i += 512) data[(size_t)(s & 1) * nb * payload + (size_t)b * payload + i] = v;
  if (w == 123.f) sink[0] = w;
}

// persistent chain with double-buffered records: ONE barrier per stage
template <int FORM>
__global__ __launch_bounds__(512, 1) void chain2_kernel(Bar* bar, float* data, int nstage, int payload, int work,
                                                        int* bad, float* sink) {
  extern __shared__ char lds[];
  const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  const unsigned xcc = xcc_id();
  unsigned nper[8];
  census(bar, nb, xcc, t, nper);
  const int nbr = (b + 37) % nb;
  float keep = 0.f;
  for (int s = 0; s < nstage; ++s) {
    if (s > 0) {
      float acc = 0.f;
      for (int i = t; i < payload; i += 512) acc += fabsf(data[(size_t)((s - 1) & 1) * nb * payload + (size_t)nbr * payload + i] - (float)((s - 1) * 1000 + nbr));
      if (acc != 0.f) atomicAdd(bad, 1);
    }
    float w = 1.0f + 1e-7f * t;
    for (int i = 0; i < work; ++i) w = fmaf(w, 1.0000001f, 1e-9f);
    keep += w * 0.f;
    float v = (float)(s * 1000 + b);
    for (int i = t; i < payload; i += 512) data[(size_t)(s & 1) * nb * payload + (size_t)b * payload + i] = v;
    grid_barrier<FORM>(bar, (unsigned)(s + 1), nb, nper, xcc, t);
  }
  if (keep == 123.f) sink[0] = keep;
}

__global__ void census_kernel(unsigned* nper) { if (threadIdx.x == 0) atomicAdd(&nper[xcc_id()], 1u); }
__global__ void tiny_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

int main() {
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  Bar* bar; float* data; int* bad; float* sink; unsigned* nper;
  const int maxpay = 16384;
  CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMalloc(&data, (size_t)2 * ncu * maxpay * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4));
  CK(hipMalloc(&nper, 32)); CK(hipMemset(nper, 0, 32));
  const int LDSB = 150 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  // census: how many of `ncu` co-resident blocks land on each XCC (the barrier needs it; placement is observed, not assumed)
  hipLaunchKernelGGL(census_kernel, dim3(ncu), dim3(64), 0, 0, nper);
  unsigned hn[8]; CK(hipMemcpy(hn, nper, 32, hipMemcpyDeviceToHost));
  printf("census of %d blocks per XCC:", ncu);
  for (int x = 0; x < 8; ++x) printf(" %u", hn[x]);
  printf("\n");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run_chain = [&](int form, int nstage, int payload, int work, int uneven) {
    CK(hipMemset(bar, 0, sizeof(Bar))); CK(hipMemset(bad, 0, 4));
    CK(hipEventRecord(e0));
    if (form == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(ncu), dim3(512), LDSB, 0, bar, data, nstage, payload, work, uneven, bad, sink);
    if (form == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(ncu), dim3(512), LDSB, 0, bar, data, nstage, payload, work, uneven, bad, sink);
    if (form == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(ncu), dim3(512), LDSB, 0, bar, data, nstage, payload, work, uneven, bad, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    Bar hbar; CK(hipMemcpy(&hbar, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    static const char* names[3] = {"naive (r03)", "counter", "xcd-hierarchical"};
    printf("%-17s %3d blocks, payload %5d floats, uneven %d, %3d stages x 2 barriers: %9.1f us total, %6.2f us per barrier, stale reads %d, timeouts %u\n",
           names[form], ncu, payload, uneven, nstage, ms * 1e3, ms * 1e3 / (2 * nstage), hb, hbar.timeout);
  };
  printf("---- barrier cost and visibility (no compute between barriers)\n");
  for (int form = 0; form < 3; ++form)
    for (int payload : {0, 1024, 16384})
      for (int uneven : {0, 1})
        run_chain(form, 100, payload, 0, uneven);
  printf("---- dependent EMPTY launches in one stream\n");
  for (int n : {10, 200}) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(ncu), dim3(512), 0, 0, data);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dependent launches: %3d empty kernels of %d blocks: %8.1f us total, %6.2f us per launch\n", n, ncu, ms * 1e3, ms * 1e3 / n);
  }
  printf("---- a chain of dependent stages (work + 16 KB record per block and stage): N launches vs ONE persistent launch with the xcd barrier\n");
  const int payload = 4096;
  for (int work : {2000, 8000, 32000, 128000}) {
    for (int nstage : {5, 50}) {
      // (a) launches
      CK(hipMemset(bad, 0, 4));
      for (int rep = 0; rep < 2; ++rep) {          // first repetition warms up
        CK(hipEventRecord(e0));
        for (int s = 0; s < nstage; ++s) hipLaunchKernelGGL(stage_kernel, dim3(ncu), dim3(512), LDSB, 0, data, s, payload, work, bad, sink, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float msl; CK(hipEventElapsedTime(&msl, e0, e1));
      int hb1; CK(hipMemcpy(&hb1, bad, 4, hipMemcpyDeviceToHost));
      // (b) persistent chain
      float msc = 0;
      int hb2 = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(bar, 0, sizeof(Bar))); CK(hipMemset(bad, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain2_kernel<2>, dim3(ncu), dim3(512), LDSB, 0, bar, data, nstage, payload, work, bad, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&msc, e0, e1));
        CK(hipMemcpy(&hb2, bad, 4, hipMemcpyDeviceToHost));
      }
      printf("work %6d iters, %2d stages: launches %8.1f us (%6.2f per stage, stale %d) | persistent + xcd barrier %8.1f us (%6.2f per stage, stale %d) | ratio %.3f\n",
             work, nstage, msl * 1e3, msl * 1e3 / nstage, hb1, msc * 1e3, msc * 1e3 / nstage, hb2, msc / msl);
    }
  }
  return 0;
}
