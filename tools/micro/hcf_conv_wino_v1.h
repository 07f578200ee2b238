// The FIRST Winograd kernel of the round-2 series (8 waves, wave = (transform row, tile group), 8 x 32 unit, global_load_lds),
// superseded by conv_wino2_kernel / conv_wino4_kernel in hcflow_amd/csrc/hcf_conv_wino.h. Kept only for the comparison table of
// tools/micro/conv_wino.hip (profiles/r02_micro_conv_wino.txt); never part of the library. Include AFTER hcf_conv_wino.h.
#pragma once
#include "hcf_conv_wino.h"

namespace hcf {
namespace wino {
#if defined(__HIPCC__)
template <int RES>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const Args a, const int nunits) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave & 3, tg = wave >> 2;
  const int H = a.H, W = a.W, ntn = a.ntile_n, nchunk = a.nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;

  // ---- DMA slots: instruction I = 8 j + wave (j = 0..6, I < 54) moves pieces [64 I, 64 I + 64) of the stage; I < 22:
  // activation pieces (piece -> swizzled (pixel, part)), else weight pieces. Waves 0..5: j = 0..2 activations, 3..6 weights;
  // waves 6, 7: j = 0, 1 activations, 2..5 weights.
  const int na = (wave < 6) ? 3 : 2;
  int adesc0, adesc1, adesc2;   // (hy << 16) | (hx << 8) | part * 16, or -1: dead piece
  auto mk_adesc = [&](int j) {
    const int pa = (8 * j + wave) * 64 + lane;
#if defined(WINO_ABL) && (WINO_ABL & 4)
    const int R = pa >> 4, sl = (pa & 15);
#else
    const int R = pa >> 4, sl = (pa & 15) ^ ((R & 7) << 1);
#endif
    const int px = R * 4 + (sl >> 2), part = sl & 3;
    const int hy = px / HWP, hx = px - hy * HWP;
    return (px < HHP * HWP) ? ((hy << 16) | (hx << 8) | (part << 4)) : -1;
  };
  adesc0 = mk_adesc(0); adesc1 = mk_adesc(1); adesc2 = mk_adesc(2);
  const gcptr wq = uniform_ptr(a.wpack);
  const gcptr zpage = uniform_ptr(a.zeros);
  const int k0 = __builtin_amdgcn_readfirstlane(a.src[0].n >> 4);
  const int k1 = k0 + __builtin_amdgcn_readfirstlane(a.nsrc > 1 ? (a.src[1].n >> 4) : 0);
  const gcptr sp0 = uniform_ptr(a.src[0].p + a.src[0].c0);
  const gcptr sp1 = uniform_ptr(a.nsrc > 1 ? a.src[1].p + a.src[1].c0 : a.src[0].p);
  const gcptr sp2 = uniform_ptr(a.nsrc > 2 ? a.src[2].p + a.src[2].c0 : a.src[0].p);
  const int csb0 = __builtin_amdgcn_readfirstlane(a.src[0].cs) * 4, csb1 = __builtin_amdgcn_readfirstlane(a.src[1].cs) * 4,
            csb2 = __builtin_amdgcn_readfirstlane(a.src[2].cs) * 4;
  const int woff = ((8 * 2 + wave - 22) * 64 + lane) * 16;      // weight-block byte offset of slot j = 2 seen as a weight slot

  // DMA cursor: unit coordinates + chunk index of the NEXT chunk to fetch
  int upix0 = -1, upix1 = -1, upix2 = -1;   // pixel index in the image of this thread's activation pieces, -1: zero padding / dead
  int ub = 0, uy0 = 0, ux0 = 0, unt = 0, uc = 0;
#define WINO_SETUP_UNIT(U)                                                                         \
  {                                                                                                \
    const int v_ = xcd_remap((U), nunits);                                                         \
    unt = __builtin_amdgcn_readfirstlane(v_ % ntn);                                                \
    const int t_ = v_ / ntn;                                                                       \
    ux0 = __builtin_amdgcn_readfirstlane((t_ % tiles_x) * TW);                                     \
    uy0 = __builtin_amdgcn_readfirstlane(((t_ / tiles_x) % tiles_y) * TH);                         \
    ub = __builtin_amdgcn_readfirstlane(t_ / (tiles_x * tiles_y));                                 \
    uc = 0;                                                                                        \
    { const int y = uy0 + (adesc0 >> 16) - 1, x = ux0 + ((adesc0 >> 8) & 255) - 1;                 \
      upix0 = (adesc0 >= 0 && y >= 0 && y < H && x >= 0 && x < W) ? (ub * H + y) * W + x : -1; }   \
    { const int y = uy0 + (adesc1 >> 16) - 1, x = ux0 + ((adesc1 >> 8) & 255) - 1;                 \
      upix1 = (adesc1 >= 0 && y >= 0 && y < H && x >= 0 && x < W) ? (ub * H + y) * W + x : -1; }   \
    { const int y = uy0 + (adesc2 >> 16) - 1, x = ux0 + ((adesc2 >> 8) & 255) - 1;                 \
      upix2 = (adesc2 >= 0 && y >= 0 && y < H && x >= 0 && x < W) ? (ub * H + y) * W + x : -1; }   \
  }
#if defined(WINO_ABL) && (WINO_ABL & 1)
#define WINO_A_SRC(J, UPIX, ADESC) (wq + (J) * 1024 + lane * 16)
#else
#define WINO_A_SRC(J, UPIX, ADESC) (((UPIX) >= 0) ? sp_ + (size_t)(unsigned)(UPIX) * csb_ + ((ADESC) & 255) : zpage + ((ADESC) & 48))
#endif
  // fetch chunk uc of the cursor's unit into stage STG, advance the cursor. Straight-line code: 7 DMA instructions (6 for waves 6, 7)
#define WINO_ISSUE_CHUNK(STG)                                                                      \
  {                                                                                                \
    char* const sbase_ = lds + ((STG) ? S1_OFF : S0_OFF) + wave * 1024;                            \
    const bool in0_ = uc < k0, in1_ = uc < k1;                                                     \
    const gcptr sp_ = (in0_ ? sp0 : in1_ ? sp1 : sp2) + (size_t)(in0_ ? uc : in1_ ? uc - k0 : uc - k1) * 64; \
    const unsigned csb_ = in0_ ? csb0 : in1_ ? csb1 : csb2;                                        \
    const gcptr wb_ = wq + ((size_t)unt * nchunk + uc) * W_BYTES + woff;                           \
    glds16(WINO_A_SRC(0, upix0, adesc0), sbase_);                                                  \
    glds16(WINO_A_SRC(1, upix1, adesc1), sbase_ + 8192);                                           \
    glds16((na == 3) ? WINO_A_SRC(2, upix2, adesc2) : wb_, sbase_ + 2 * 8192);                     \
    glds16(wb_ + 8192, sbase_ + 3 * 8192);                                                         \
    glds16(wb_ + 2 * 8192, sbase_ + 4 * 8192);                                                     \
    glds16(wb_ + 3 * 8192, sbase_ + 5 * 8192);                                                     \
    if (na == 3) glds16(wb_ + 4 * 8192, sbase_ + 6 * 8192);                                        \
    ++uc;                                                                                          \
  }
  // ---- fragment read offsets ----------------------------------------------------------------------------------------
  // patch rows of transform row xi: B^T = (1,0,-1,0), (0,1,1,0), (0,-1,1,0), (0,1,0,-1)
  const int r1 = (xi == 0) ? 0 : 1, r2 = (xi == 3) ? 3 : 2;
  const float sigma = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (xi == 1) ? 1.f : -1.f)));
  const int trow = li >> 4, tcol = li & 15;
  int poff[2][4];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int j = 0; j < 4; ++j) poff[rr][j] = a_off((4 * tg + 2 * trow + (rr ? r2 : r1)) * HWP + 2 * tcol + j, 2 * half);
  const int fw = A_BYTES + (xi * 4 * 4 + half) * 512 + li * 16;      // + (nu * 4 + plane * 2) * 512

  if (tid < 64) {       // y = act((acc / 2^11 + bias) * scale) = act(acc * ms + bs)
    const float sc_ = (tid < 32 * ntn) ? a.scale[tid] : 1.f, bi_ = (tid < 32 * ntn) ? a.bias[tid] : 0.f;
    reinterpret_cast<float*>(lds + TAB_OFF)[tid] = bi_ * sc_;
    reinterpret_cast<float*>(lds + TAB_OFF)[64 + tid] = sc_ * UNSPLIT;
  }

#if defined(WINO_PROF)
  unsigned long long pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pw_t0 = __builtin_readcyclecounter();
#endif
  int u = blockIdx.x;
  if (u >= nunits) return;
  WINO_SETUP_UNIT(u)
  WINO_ISSUE_CHUNK(0)
  int g = 0;
  const float slope = a.act == 1 ? 0.f : a.act == 2 ? 0.2f : 1.f;
  const float alo_ = (slope == 0.f) ? -3.0e38f : -INFINITY;

  while (true) {
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    const int eb = ub, ey0 = uy0, ex0 = ux0, ent = unt;
    const int un = u + gridDim.x;

    for (int c = 0; c < nchunk; ++c, ++g) {
      const int stg = g & 1;
#if defined(WINO_PROF)
      const unsigned long long q0 = __builtin_readcyclecounter();
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(WINO_PROF)
      const unsigned long long q1 = __builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_barrier();
#if defined(WINO_PROF)
      const unsigned long long q2 = __builtin_readcyclecounter();
#endif
      if (c + 1 == nchunk) {               // the cursor moves on to the next unit (or idles on the zero page)
        if (un < nunits) WINO_SETUP_UNIT(un)
        else { upix0 = upix1 = upix2 = -1; uc = 0; unt = 0; }
      }
      if (stg) WINO_ISSUE_CHUNK(0) else WINO_ISSUE_CHUNK(1)
#if defined(WINO_PROF)
      const unsigned long long q3 = __builtin_readcyclecounter();
      pw[0] += q1 - q0; pw[1] += q2 - q1; pw[5] += q3 - q2;
#endif
      const char* const sb = lds + (stg ? S1_OFF : S0_OFF);
      // raw patch pixels: 2 rows x 4 columns x 8 channels
      f32x4 d[2][4][2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          d[rr][j][0] = *reinterpret_cast<const f32x4*>(sb + poff[rr][j]);
          d[rr][j][1] = *reinterpret_cast<const f32x4*>(sb + (poff[rr][j] ^ 16));
        }
      f16x8 vh[4], vl[4];
      {
        float t[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 8; ++k) t[j][k] = __builtin_fmaf(sigma, d[1][j][k >> 2][k & 3], d[0][j][k >> 2][k & 3]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v[4] = {t[0][k] - t[2][k], t[1][k] + t[2][k], t[1][k] - t[2][k], t[1][k] - t[3][k]};
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const _Float16 h = (_Float16)v[p];
            vh[p][k] = h;
            vl[p][k] = (_Float16)(v[p] - (float)h);
          }
        }
      }
#if defined(WINO_PROF)
      asm volatile("" :: "v"(vh[0][0]), "v"(vl[3][7]));
      pw[6] += __builtin_readcyclecounter() - q3;
#endif
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const f16x8 w1 = *reinterpret_cast<const f16x8*>(sb + fw + (p * 4) * 512);
        const f16x8 w2 = *reinterpret_cast<const f16x8*>(sb + fw + (p * 4 + 2) * 512);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, vh[p], acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, vh[p], acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, vl[p], acc[p], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue of unit (eb, ey0, ex0, ent) ---------------------------------------------------------------------------
#if defined(WINO_PROF)
    const unsigned long long qe0 = __builtin_readcyclecounter();
#endif
    {
      float chk = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) chk = fmaf(acc[p][r], 0.f, chk);
      if (__any(chk != chk)) {
        if (lane == 0) atomicOr(a.ovf, 1);
      }
    }
#if defined(WINO_ABL) && (WINO_ABL & 2)
    if (acc[0][0] == 123.456f) a.out[0] = acc[1][1] + acc[2][2] + acc[3][3];
    u = un;
    if (u >= nunits) break;
    continue;
#endif
    // R[b] = sum_nu M[xi][nu] A[nu][b]:  R0 = M0 + M1 + M2,  R1 = M1 - M2 - M3
    // exchange buffer (free stage + X): [tg][xi][b][q][lane] 16-byte pieces
    char* const xb = lds + ((g & 1) ? S0_OFF : X_OFF);       // stage g & 1 is receiving the next chunk; the other one (+ X) is free
    __builtin_amdgcn_s_barrier();                             // every wave is done reading the free stage
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 r0, r1_;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e;
        r0[e] = acc[0][r] + acc[1][r] + acc[2][r];
        r1_[e] = acc[1][r] - acc[2][r] - acc[3][r];
      }
      *reinterpret_cast<f32x4*>(xb + ((((tg * 4 + xi) * 2 + 0) * 4 + q) * 64 + lane) * 16) = r0;
      *reinterpret_cast<f32x4*>(xb + ((((tg * 4 + xi) * 2 + 1) * 4 + q) * 64 + lane) * 16) = r1_;
    }
    __builtin_amdgcn_s_barrier();
    // wave (a, qp) = (xi >> 1, xi & 1): output row 2 trow + a of the patch, channel quarters q = 2 qp, 2 qp + 1, both columns b
    {
      const int oa = xi >> 1, qp = xi & 1;
      const int y = ey0 + 4 * tg + 2 * trow + oa;
      const int cb = ent * 32 + 4 * half;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int x = ex0 + 2 * tcol + b;
        const bool ok = y < H && x < W;
        const size_t pix = (size_t)((size_t)eb * H + (y < H ? y : H - 1)) * W + (x < W ? x : W - 1);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = 2 * qp + qq;
          auto R = [&](int x_) { return *reinterpret_cast<const f32x4*>(xb + ((((tg * 4 + x_) * 2 + b) * 4 + q) * 64 + lane) * 16); };
          const f32x4 ra = R(oa ? 1 : 0), rb = R(oa ? 2 : 1), rc = R(oa ? 3 : 2);
          const f32x4 bs = *reinterpret_cast<const f32x4*>(lds + TAB_OFF + (cb + 8 * q) * 4);
          const f32x4 ms = *reinterpret_cast<const f32x4*>(lds + TAB_OFF + 256 + (cb + 8 * q) * 4);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float yv = oa ? (ra[e] - rb[e] - rc[e]) : (ra[e] + rb[e] + rc[e]);     // Y1 = R1 - R2 - R3, Y0 = R0 + R1 + R2
            float t = fmaf(yv, ms[e], bs[e]);
            t = act1(t, slope, alo_);
            v[e] = t;
          }
          if (RES >= 1) {
            const f32x4 r1v = *reinterpret_cast<const f32x4*>(a.res1 + pix * a.res1_cs + a.res1_c0 + cb + 8 * q);
            v = v * a.rs1 + r1v;
          }
          if (RES == 2) {
            const f32x4 r2v = *reinterpret_cast<const f32x4*>(a.res2 + pix * a.res2_cs + a.res2_c0 + cb + 8 * q);
            v = v * a.rs2 + r2v;
          }
          if (ok && cb + 8 * q < a.cout) *reinterpret_cast<f32x4*>(a.out + pix * a.out_cs + a.out_c0 + cb + 8 * q) = v;
        }
      }
    }
#if defined(WINO_PROF)
    pw[3] += __builtin_readcyclecounter() - qe0;
#endif
    u = un;
    if (u >= nunits) break;
  }
#undef WINO_SETUP_UNIT
#undef WINO_ISSUE_CHUNK
#undef WINO_A_SRC
#if defined(WINO_PROF)
  if (a.dbg && lane == 0 && (blockIdx.x & 31) == 17) {
    atomicAdd(a.dbg + 0, pw[0]); atomicAdd(a.dbg + 1, pw[1]); atomicAdd(a.dbg + 2, __builtin_readcyclecounter() - pw_t0);
    atomicAdd(a.dbg + 3, pw[3]); atomicAdd(a.dbg + 4, 1ull); atomicAdd(a.dbg + 5, pw[5]); atomicAdd(a.dbg + 6, pw[6]);
  }
#endif
}


static inline int launch_v1(const Args& a, int ncu, hipStream_t st) {
  if (a.nsrc < 1 || a.nsrc > 3 || !a.wpack || !a.bias || !a.scale || !a.ovf || !a.zeros || !a.out || a.nchunk < 1) return -1;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long long nunits = (long long)a.B * tiles_x * tiles_y * a.ntile_n;
  const unsigned grid = (unsigned)(nunits < ncu ? nunits : ncu);
  const int res = a.res2 ? 2 : a.res1 ? 1 : 0;
  auto go = [&](auto fn) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -2;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), LDS_BYTES, st, a, (int)nunits);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  };
  if (res == 0) return go(conv_wino_kernel<0>);
  if (res == 1) return go(conv_wino_kernel<1>);
  return go(conv_wino_kernel<2>);
}
#endif  // __HIPCC__
}  // namespace wino
}  // namespace hcf
