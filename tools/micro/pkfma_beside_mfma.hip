// Reproducer of the round-1 "co-residency fault" (profiles/r02_fault_rootcause.md): packed-fp32 VALU results go wrong while
// OTHER waves on the same SIMD are issuing MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pkfma_beside_mfma.hip -o build/micro/pkfma && build/micro/pkfma
// Waves 0-3 of every 8-wave block run an MFMA loop (or idle), waves 4-7 (waves w and w + 4 share a SIMD) a chain of three packed
// operations that is checked, in the same lane, against the same arithmetic done with scalar VALU instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: v_pk_fma_f32 with the op_sel forms of the faulty stream; 1: plain v_pk_fma_f32; 2: v_pk_mul_f32 + v_pk_add_f32;
//      3: scalar v_fma_f32 against scalar v_fma_f32 (control); 4: MODE 0 with s_nop 4 after every packed instruction;
//      5: three INDEPENDENT v_pk_fma_f32 op_sel_hi:[1,0,1]; 6: dependent chain, op_sel_hi:[1,0,1] only; 7: chain, op_sel:[0,1,0] only
// MF 0: no MFMA waves; 1: v_mfma_f32_32x32x16_f16; 2: v_mfma_f32_32x32x2_f32 (fp32 inputs); 3: v_mfma_f32_16x16x32_f16;
//    4: v_mfma_f32_32x32x16_bf16
template <int MODE, int MF>
__global__ __launch_bounds__(512, 1) void probe(float* sink, unsigned long long* bad, int iters) {
  const int lane = threadIdx.x & 63;
  if ((threadIdx.x >> 8) == 0) {
    if (MF == 0) return;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MF == 1) {
      f16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
      for (int i = 0; i < iters * 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else if (MF == 2) {
      const float a = 0.01f * lane, b = 0.02f * (lane - 7);
      for (int i = 0; i < iters * 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    } else if (MF == 3) {
      f16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < iters * 16; ++i) a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, a4, 0, 0, 0);
      acc[0] = a4[0]; acc[15] = a4[3];
    } else {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      bf16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
      for (int i = 0; i < iters * 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[15];
    return;
  }
  unsigned s = 1234567u + 977u * (blockIdx.x * 512 + threadIdx.x);
  unsigned long long nbad = 0;
  for (int i = 0; i < iters; ++i) {
    float v[8];
    for (int e = 0; e < 8; ++e) { s = s * 1664525u + 1013904223u; v[e] = (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f; }
    f32x2 m0 = {v[0], v[1]}, m1 = {v[2], v[3]}, z = {v[4], v[5]}, c = {v[6], v[7]}, r0, r1, r2;
    float e2l, e2h;
    if (MODE == 0 || MODE == 4) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r0) : "v"(m0), "v"(z), "v"(c));
      if (MODE == 4) asm volatile("s_nop 4");
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r1) : "v"(m1), "v"(z), "v"(r0));
      if (MODE == 4) asm volatile("s_nop 4");
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r2) : "v"(m0), "v"(z), "v"(r1));
      if (MODE == 4) asm volatile("s_nop 4");
      const float e0l = __builtin_fmaf(m0[0], z[0], c[0]), e0h = __builtin_fmaf(m0[1], z[0], c[1]);
      const float e1l = __builtin_fmaf(m1[0], z[1], e0l), e1h = __builtin_fmaf(m1[1], z[1], e0h);
      e2l = __builtin_fmaf(m0[0], z[0], e1l); e2h = __builtin_fmaf(m0[1], z[0], e1h);
    } else if (MODE == 5) {
      f32x2 q0, q1, q2;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(q0) : "v"(m0), "v"(z), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(q1) : "v"(m1), "v"(z), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(q2) : "v"(c), "v"(z), "v"(m0));
      r2[0] = q0[0] + q1[0] + q2[0]; r2[1] = q0[1] + q1[1] + q2[1];
      const float x0 = __builtin_fmaf(m0[0], z[0], c[0]), y0 = __builtin_fmaf(m0[1], z[0], c[1]);
      const float x1 = __builtin_fmaf(m1[0], z[0], c[0]), y1 = __builtin_fmaf(m1[1], z[0], c[1]);
      const float x2 = __builtin_fmaf(c[0], z[0], m0[0]), y2 = __builtin_fmaf(c[1], z[0], m0[1]);
      e2l = x0 + x1 + x2; e2h = y0 + y1 + y2;
    } else if (MODE == 6) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r0) : "v"(m0), "v"(z), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r1) : "v"(m1), "v"(z), "v"(r0));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r2) : "v"(m0), "v"(z), "v"(r1));
      const float e0l = __builtin_fmaf(m0[0], z[0], c[0]), e0h = __builtin_fmaf(m0[1], z[0], c[1]);
      const float e1l = __builtin_fmaf(m1[0], z[0], e0l), e1h = __builtin_fmaf(m1[1], z[0], e0h);
      e2l = __builtin_fmaf(m0[0], z[0], e1l); e2h = __builtin_fmaf(m0[1], z[0], e1h);
    } else if (MODE == 7) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r0) : "v"(m0), "v"(z), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r1) : "v"(m1), "v"(z), "v"(r0));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r2) : "v"(m0), "v"(z), "v"(r1));
      const float e0l = __builtin_fmaf(m0[0], z[1], c[0]), e0h = __builtin_fmaf(m0[1], z[1], c[1]);
      const float e1l = __builtin_fmaf(m1[0], z[1], e0l), e1h = __builtin_fmaf(m1[1], z[1], e0h);
      e2l = __builtin_fmaf(m0[0], z[1], e1l); e2h = __builtin_fmaf(m0[1], z[1], e1h);
    } else if (MODE == 1) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(m0), "v"(z), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(m1), "v"(z), "v"(r0));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r2) : "v"(m0), "v"(z), "v"(r1));
      const float e0l = __builtin_fmaf(m0[0], z[0], c[0]), e0h = __builtin_fmaf(m0[1], z[1], c[1]);
      const float e1l = __builtin_fmaf(m1[0], z[0], e0l), e1h = __builtin_fmaf(m1[1], z[1], e0h);
      e2l = __builtin_fmaf(m0[0], z[0], e1l); e2h = __builtin_fmaf(m0[1], z[1], e1h);
    } else if (MODE == 2) {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(m0), "v"(z));
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r1) : "v"(r0), "v"(c));
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r2) : "v"(r1), "v"(m1));
      float pl = m0[0] * z[0], ph = m0[1] * z[1];
      asm volatile("" : "+v"(pl), "+v"(ph));                       // (no contraction into an FMA)
      float sl = pl + c[0], sh = ph + c[1];
      asm volatile("" : "+v"(sl), "+v"(sh));
      e2l = sl * m1[0]; e2h = sh * m1[1];
    } else {
      float a0, a1, a2, b0, b1, b2;
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(m0[0]), "v"(z[0]), "v"(c[0]));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(m1[0]), "v"(z[1]), "v"(a0));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a2) : "v"(m0[0]), "v"(z[0]), "v"(a1));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(b0) : "v"(m0[1]), "v"(z[0]), "v"(c[1]));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(b1) : "v"(m1[1]), "v"(z[1]), "v"(b0));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(b2) : "v"(m0[1]), "v"(z[0]), "v"(b1));
      r2[0] = a2; r2[1] = b2;
      const float e0l = __builtin_fmaf(m0[0], z[0], c[0]), e0h = __builtin_fmaf(m0[1], z[0], c[1]);
      const float e1l = __builtin_fmaf(m1[0], z[1], e0l), e1h = __builtin_fmaf(m1[1], z[1], e0h);
      e2l = __builtin_fmaf(m0[0], z[0], e1l); e2h = __builtin_fmaf(m0[1], z[0], e1h);
    }
    if (r2[0] != e2l || r2[1] != e2h) ++nbad;
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int MODE, int MF>
static void run(const char* what, float* sink, unsigned long long* bad) {
  const int iters = 4000;
  hipMemset(bad, 0, 8);
  hipLaunchKernelGGL((probe<MODE, MF>), dim3(2048), dim3(512), 0, 0, sink, bad, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  printf("%-44s %-28s mismatches %12llu of %llu (%.3f %%)\n", what, MF == 0 ? "no MFMA waves" : MF == 1 ? "beside mfma_32x32x16_f16" : MF == 2 ? "beside mfma_32x32x2_f32" : MF == 3 ? "beside mfma_16x16x32_f16" : "beside mfma_32x32x16_bf16",
         h, 2048ull * 256 * iters, 100.0 * h / (2048.0 * 256 * iters));
}

int main() {
  float* sink; unsigned long long* bad;
  hipMalloc(&sink, 2048 * 512 * 4); hipMalloc(&bad, 8);
#define ALL(MODE, WHAT) run<MODE, 0>(WHAT, sink, bad); run<MODE, 1>(WHAT, sink, bad); run<MODE, 2>(WHAT, sink, bad); run<MODE, 3>(WHAT, sink, bad); run<MODE, 4>(WHAT, sink, bad);
  ALL(0, "v_pk_fma_f32 with op_sel (fault's stream)")
  ALL(1, "v_pk_fma_f32 plain")
  ALL(2, "v_pk_mul_f32 / v_pk_add_f32")
  ALL(3, "scalar v_fma_f32 (control)")
  ALL(4, "v_pk_fma_f32 op_sel + s_nop 4 after each")
  ALL(5, "3 independent v_pk_fma_f32 op_sel_hi:[1,0,1]")
  ALL(6, "chain, op_sel_hi:[1,0,1] only")
  ALL(7, "chain, op_sel:[0,1,0] only")
  return 0;
}
