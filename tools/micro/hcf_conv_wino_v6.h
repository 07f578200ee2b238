// tools/micro only: version 6 of the 64-output-channel Winograd kernel (round 5) -- v4 (hcf_conv_wino.h: conv_wino4_kernel) with the
// row phase software-pipelined under the position loop. Bit-identical to v4 on every harness check and NOT faster (profiles/r05_notes.md
// section 1: 938-947 us against 941-953 us on RDB conv5 at 16 x 320^2; the time the row phase gives up reappears inside the position
// loop), so the product keeps v4. Same packs, same LDS layout, same epilogue (wino64_epilogue).
#pragma once
#include "hcf_conv_wino.h"

namespace hcf {
namespace wino {
#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------------------------------
// v6 (round 5): v4 with the ROW PHASE software-pipelined under the position loop. v4's anatomy (profiles/r03_notes.md section 9):
// position loop 44.7 % of a wave's life with the matrix pipe ~100 % busy, then a row phase of 22.6 % in which all eight waves
// read their patch rows at once (128 KB of ds_read_b128 per chunk and CU behind one barrier) and nothing multiplies. Here the
// rows of chunk c + 1 are read and row-transformed INSIDE chunk c's position loop (one patch column per position block, its
// 8 FMAs one block later), so a chunk is: wait + barrier -> position loop. What makes that fit:
//   * the image cursor runs TWO chunks ahead of the MFMAs (the image of chunk c + 2 is requested right after chunk c's barrier),
//     the weight cursor one chunk ahead as before. Two image buffers still suffice: image c was consumed into registers during
//     chunk c - 1, so its buffer is free once every wave has passed chunk c's barrier.
//   * registers: t of the current chunk dies column by column while the next chunk's fills (positions run in the order
//     0, 3, 1, 2: column 0 is dead after the first block, column 3 after the second); the row phase's 64 raw registers are gone.
// Same packs, same arithmetic, same epilogue as v4: bit-identical results.
template <int RES, int VAR>
__global__ __launch_bounds__(512, 1) void conv_wino6_kernel(const Args a, const int nunits) {
  using namespace v4;
  using v2::ROWB;
  using v2::a2_off;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave & 3, tg = wave >> 2;
  const int H = a.H, W = a.W, nchunk = a.nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH4 - 1) / TH4;

  const bool whi = (wave >= 6);
  const int padpix = a.B * H * W;
  int upix[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) upix[j] = padpix;
  uint32_t partpk = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int pa_ = (8 * j + wave) * 64 + lane, hy_ = (pa_ * 241) >> 15, q_ = pa_ - hy_ * 136, m_ = q_ >> 4;
    partpk |= (uint32_t)(((q_ & 15) ^ (m_ & 7)) & 3) << (4 + 2 * j);
  }
  const int k0 = __builtin_amdgcn_readfirstlane(a.src[0].n >> 4);
  const int k1 = k0 + __builtin_amdgcn_readfirstlane(a.nsrc > 1 ? (a.src[1].n >> 4) : 0);
  const long long npx = (long long)a.B * H * W;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[0].p, 0, (int)(npx * a.src[0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 1 ? 1 : 0].p, 0, (int)(npx * a.src[a.nsrc > 1 ? 1 : 0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 2 ? 2 : 0].p, 0, (int)(npx * a.src[a.nsrc > 2 ? 2 : 0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, (nchunk + 1) * W4_BYTES, 0x00020000);
  const int csb0 = __builtin_amdgcn_readfirstlane(a.src[0].cs) * 4, csb1 = __builtin_amdgcn_readfirstlane(a.src[1].cs) * 4,
            csb2 = __builtin_amdgcn_readfirstlane(a.src[2].cs) * 4;
  const int cb0 = __builtin_amdgcn_readfirstlane(a.src[0].c0) * 4, cb1 = __builtin_amdgcn_readfirstlane(a.src[1].c0) * 4,
            cb2 = __builtin_amdgcn_readfirstlane(a.src[2].c0) * 4;
  const int wvo = lane * 16;

  // image cursor: unit iu, chunk ic of it (two chunks ahead of the MFMAs); weight cursor: chunk wc (one ahead; unit-independent)
  int iu = blockIdx.x, ic = 0, wc = 0;
#define W6_SETUP_UNIT(U)                                                                           \
  {                                                                                                \
    const int v_ = xcd_remap((U), nunits);                                                         \
    const int ux0_ = __builtin_amdgcn_readfirstlane((v_ % tiles_x) * TW);                          \
    const int uy0_ = __builtin_amdgcn_readfirstlane(((v_ / tiles_x) % tiles_y) * TH4);             \
    const int ub_ = __builtin_amdgcn_readfirstlane(v_ / (tiles_x * tiles_y));                      \
    int ln_ = lane;                                                                                \
    asm volatile("" : "+v"(ln_));                                                                  \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                \
      const int pa_ = (8 * j + wave) * 64 + ln_;                                                   \
      const int hy_ = (pa_ * 241) >> 15;                                                           \
      const int q_ = pa_ - hy_ * 136, m_ = q_ >> 4;                                                \
      const int hx_ = m_ * 4 + (((q_ & 15) ^ (m_ & 7)) >> 2);                                      \
      const int y = uy0_ + hy_ - 1, x = ux0_ + hx_ - 1;                                            \
      upix[j] = (hy_ < HH4 && y >= 0 && y < H && x >= 0 && x < W) ? (ub_ * H + y) * W + x : padpix; \
    }                                                                                              \
  }
#define W6_DMA(RS, VOFF, SOFF, DST) __builtin_amdgcn_raw_ptr_buffer_load_lds((RS), (lptr)(DST), 16, (VOFF), (SOFF), 0, 0)
  int csb_ = 0, so_ = 0, ws_ = 0;
  uint32_t pp_ = partpk;
  __amdgpu_buffer_rsrc_t rsa_ = rs0;
#define W6_IMG_SCALARS()                                                                           \
  {                                                                                                \
    const int sidx_ = (ic < k0) ? 0 : (ic < k1) ? 1 : 2;                                           \
    csb_ = sidx_ == 0 ? csb0 : sidx_ == 1 ? csb1 : csb2;                                           \
    rsa_ = sidx_ == 0 ? rs0 : sidx_ == 1 ? rs1 : rs2;                                              \
    so_ = (sidx_ == 0 ? cb0 + ic * 64 : sidx_ == 1 ? cb1 + (ic - k0) * 64 : cb2 + (ic - k1) * 64); \
    pp_ = partpk;                                                                                  \
    asm volatile("" : "+v"(pp_));                                                                  \
  }
#define W6_A_SLOT(J, IB)                                                                           \
  {                                                                                                \
    const int p16_ = (int)((pp_ >> (2 * (J))) & 0x30u);                                            \
    const int vo_ = (int)__umul24((unsigned)upix[J], (unsigned)csb_) + p16_;                       \
    W6_DMA(rsa_, vo_, so_, lds + (IB) * A4_BYTES + (8 * (J) + wave) * 1024);                       \
  }
  // the image cursor's chunk into image buffer IB, then the cursor moves on (next chunk, or chunk 0 of this block's next unit)
#define W6_ISSUE_A(IB)                                                                             \
  {                                                                                                \
    W6_IMG_SCALARS()                                                                               \
    W6_A_SLOT(0, IB) W6_A_SLOT(1, IB)                                                              \
    if (!whi) W6_A_SLOT(2, IB)                                                                     \
    if (++ic == nchunk) {                                                                          \
      ic = 0;                                                                                      \
      iu += gridDim.x;                                                                             \
      if (iu < nunits) W6_SETUP_UNIT(iu)                                                           \
      else {                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) upix[j] = padpix;                            \
      }                                                                                            \
    }                                                                                              \
  }
#define W6_ISSUE_W(J0, J1, WB)                                                                     \
  {                                                                                                \
    char* const wb_ = lds + ((WB) ? W4_OFF1 : W4_OFF0);                                            \
    _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) {                                       \
      const int l_ = 8 * j_ + wave - 22;                                                           \
      const int x_ = l_ / 12, r_ = l_ - 12 * x_, n_ = r_ / 3;                                      \
      const int g_ = x_ * 16 + n_ * 4 + (r_ - 3 * n_);                                             \
      if (j_ == 2) { if (whi) W6_DMA(rsw, wvo, ws_ + g_ * 1024, wb_ + l_ * 1024); }                \
      else if (j_ < 8) W6_DMA(rsw, wvo, ws_ + g_ * 1024, wb_ + l_ * 1024);                         \
      else if (!whi) W6_DMA(rsw, wvo, ws_ + g_ * 1024, wb_ + l_ * 1024);                           \
    }                                                                                              \
  }
  const gcptr w11p = uniform_ptr(a.wpack + (size_t)(xi * 16 + 3) * 1024);
#define W6_LOAD_W11(NU) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(w11[NU]) : "v"(wvo), "s"(w11p + ws_ + (NU) * 4096) : "memory");

  const int trow = li >> 4, tcol = li & 15;
  const int rA = (xi == 0) ? 0 : 1, rB = (xi == 3) ? 3 : 2;
  const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (xi == 1) ? 1.f : -1.f)));
  int poff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) poff[j] = a2_off(4 * tg + 2 * trow, 2 * tcol + j, 2 * half);
  const int offA = rA * ROWB, offB = rB * ROWB;
  const int fw = lane * 16 + xi * (12 * 1024);

  constexpr bool F1 = (RES == 3);
  if (tid < 64) {
    reinterpret_cast<float*>(lds + TAB4_OFF)[tid] = a.bias[tid] * a.scale[tid];
    reinterpret_cast<float*>(lds + TAB4_OFF)[64 + tid] = a.scale[tid] * UNSPLIT;
    if (F1) {
      reinterpret_cast<float*>(lds + TAB4_OFF)[128 + tid] = a.f_bias[tid] * a.f_scale[tid];
      reinterpret_cast<float*>(lds + TAB4_OFF)[192 + tid] = a.f_scale[tid] * UNSPLIT;
    }
  }
  int u = blockIdx.x;
  if (u >= nunits) return;
#if defined(WINO_PROF)
  unsigned long long pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pw_t0 = __builtin_readcyclecounter();
#endif
  // raw patch column J (rows rA, rB; both 16-byte parts) of image buffer IB -> 16 registers; its row transform -> T[J][0..7]
#define W6_READ_COL(RAW, IB, J)                                                                    \
  {                                                                                                \
    const char* const ib_ = lds + (IB) * A4_BYTES;                                                 \
    RAW[0] = *reinterpret_cast<const f32x4*>(ib_ + poff[J] + offA);                                \
    RAW[1] = *reinterpret_cast<const f32x4*>(ib_ + (poff[J] ^ 16) + offA);                         \
    RAW[2] = *reinterpret_cast<const f32x4*>(ib_ + poff[J] + offB);                                \
    RAW[3] = *reinterpret_cast<const f32x4*>(ib_ + (poff[J] ^ 16) + offB);                         \
  }
#define W6_ROW_T(T, J, RAW)                                                                        \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                  \
    T[J][k] = fmaf(sg, RAW[2][k], RAW[0][k]);                                                      \
    T[J][4 + k] = fmaf(sg, RAW[3][k], RAW[1][k]);                                                  \
  }
  f16x8 w11[4];
  float t_[4][8];
  // ---- prologue: image 0 -> buffer 0, weights 0 -> weight buffer 0 (+ the register quarter), image 1 -> buffer 1; t = rows(image 0)
  W6_SETUP_UNIT(iu)
  W6_ISSUE_A(0)
  ws_ = 0;
  W6_ISSUE_W(2, 9, 0)
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) W6_LOAD_W11(nu)
  wc = (nchunk > 1) ? 1 : 0;
  W6_ISSUE_A(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    f32x4 raw_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      W6_READ_COL(raw_, 0, j)
      W6_ROW_T(t_, j, raw_)
    }
  }
  int g = 0;
  const float slope = a.act == 1 ? 0.f : a.act == 2 ? 0.2f : 1.f;
  const float slope2 = a.act2 == 1 ? 0.f : a.act2 == 2 ? 0.2f : 1.f;
  const float slope_f = a.f_act == 1 ? 0.f : a.f_act == 2 ? 0.2f : 1.f;

  while (true) {
    f32x16 acc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][n][r] = 0.f;
    int eb, ey0, ex0;
    {
      const int v_ = xcd_remap(u, nunits);
      ex0 = __builtin_amdgcn_readfirstlane((v_ % tiles_x) * TW);
      ey0 = __builtin_amdgcn_readfirstlane(((v_ / tiles_x) % tiles_y) * TH4);
      eb = __builtin_amdgcn_readfirstlane(v_ / (tiles_x * tiles_y));
    }
    const int un = u + gridDim.x;

#if defined(WINO_PROF)
#define W6_TICK(X) const unsigned long long X = __builtin_readcyclecounter();
#define W6_ADD(I, A, B) pw[I] += (B) - (A);
#else
#define W6_TICK(X)
#define W6_ADD(I, A, B)
#endif
#define W6_MFMA(P, N, WW, VX) acc[P][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WW, __builtin_bit_cast(f16x8, VX), acc[P][N], 0, 0, 0);
    // Position block BI handles position NU with the transformed values V (8 per lane); unless LAST (the unit's last chunk) it reads
    // patch column COL of the NEXT chunk's image at its start and row-transforms it IN PLACE into t_[COL] at its end -- the block
    // order 0, 3, 1, 2 frees t_[0], t_[3], then t_[1] / t_[2] (both V1 = t1 + t2 and V2 = t1 - t2 are formed in the third block)
    // exactly when their successors arrive.
#define W6_BLOCK(LAST, BI, NU, V, COL)                                                             \
      {                                                                                            \
        if (!(LAST)) { W6_READ_COL(rw_, ibn, COL) }                                                \
        u32x4 vh_, vl_;                                                                            \
        split8(V, vh_, vl_);                                                                       \
        const f16x8 w00 = *reinterpret_cast<const f16x8*>(wb + ((NU) * 3 + 0) * 1024);             \
        const f16x8 w01 = *reinterpret_cast<const f16x8*>(wb + ((NU) * 3 + 1) * 1024);             \
        const f16x8 w10 = *reinterpret_cast<const f16x8*>(wb + ((NU) * 3 + 2) * 1024);             \
        W6_MFMA(NU, 0, w00, vh_)                                                                   \
        W6_MFMA(NU, 1, w10, vh_)                                                                   \
        W6_MFMA(NU, 0, w01, vh_)                                                                   \
        W6_MFMA(NU, 1, w11[NU], vh_)                                                               \
        W6_MFMA(NU, 0, w00, vl_)                                                                   \
        W6_MFMA(NU, 1, w10, vl_)                                                                   \
        if ((BI) == 0) { W6_ISSUE_W(2, 5, stg ^ 1) }                                               \
        else if ((BI) == 1) { W6_ISSUE_W(5, 7, stg ^ 1) }                                          \
        else if ((BI) == 2) { W6_ISSUE_W(7, 9, stg ^ 1) }                                          \
        W6_LOAD_W11(NU)                                                                            \
        if (!(LAST)) { W6_ROW_T(t_, COL, rw_) }                                                    \
        if (VAR & 1) __builtin_amdgcn_sched_barrier(0);                                            \
      }
    // One chunk: wait + barrier (this chunk's weights and the next chunk's image are complete and visible; every wave is through the
    // previous chunk: its weight buffer and the image it read are free), request the image two chunks ahead into the buffer whose
    // rows were read during the previous chunk, then the four position blocks.
#define W6_CHUNK(LAST)                                                                             \
    {                                                                                              \
      const int stg = g & 1;                                                                       \
      W6_TICK(q0)                                                                                  \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
      W6_TICK(q1)                                                                                  \
      __builtin_amdgcn_s_barrier();                                                                \
      W6_TICK(q2)                                                                                  \
      W6_ADD(0, q0, q1) W6_ADD(1, q1, q2)                                                          \
      _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) asm volatile("" : "+v"(w11[nu]));          \
      W6_ISSUE_A(g & 1)                                                                            \
      ws_ = wc * W4_BYTES;                                                                         \
      W6_TICK(q3)                                                                                  \
      W6_ADD(5, q2, q3)                                                                            \
      const char* const wb = lds + (stg ? W4_OFF1 : W4_OFF0) + fw;                                 \
      const int ibn = (g + 1) & 1;                                                                 \
      f32x4 rw_[4];                                                                                \
      float v_[8], v2_[8];                                                                         \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) v_[k] = t_[0][k] - t_[2][k];                   \
      W6_BLOCK(LAST, 0, 0, v_, 0)                                                                  \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) v_[k] = t_[1][k] - t_[3][k];                   \
      W6_BLOCK(LAST, 1, 3, v_, 3)                                                                  \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) { v_[k] = t_[1][k] + t_[2][k]; v2_[k] = t_[1][k] - t_[2][k]; } \
      W6_BLOCK(LAST, 2, 1, v_, 1)                                                                  \
      W6_BLOCK(LAST, 3, 2, v2_, 2)                                                                 \
      if (++wc == nchunk) wc = 0;                                                                  \
      W6_TICK(q4)                                                                                  \
      W6_ADD(6, q3, q4)                                                                            \
      ++g;                                                                                         \
    }
    for (int c = 0; c + 1 < nchunk; ++c) W6_CHUNK(false)
    // the unit's last chunk does NOT pre-transform the next unit's first rows: t would be live through the epilogue (32 registers
    // it does not have); the row phase of a unit's first chunk runs after the epilogue instead (once per unit, as every chunk in v4)
    W6_CHUNK(true)

#if defined(WINO_PROF)
    const unsigned long long qe0 = __builtin_readcyclecounter();
#endif
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    wino64_epilogue<RES>(a, lds, TAB4_OFF, lds + (((g - 1) & 1) ? W4_OFF0 + W4L_BYTES : W4_OFF0), acc, eb, ey0, ex0, H, W, slope, slope2, slope_f, wave, lane_e);
#if defined(WINO_PROF)
    pw[3] += __builtin_readcyclecounter() - qe0;
#endif
    u = un;
    if (u >= nunits) break;
    {
      // rows of the next unit's first chunk: its image (buffer g & 1) was complete at the last chunk's barrier
      W6_TICK(r0)
      f32x4 raw_[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        W6_READ_COL(raw_, g & 1, j)
        W6_ROW_T(t_, j, raw_)
      }
      W6_TICK(r1)
      W6_ADD(7, r0, r1)
    }
  }
#if defined(WINO_PROF)
  if (a.dbg && lane == 0 && (blockIdx.x & 31) == 17) {
    atomicAdd(a.dbg + 0, pw[0]); atomicAdd(a.dbg + 1, pw[1]); atomicAdd(a.dbg + 2, __builtin_readcyclecounter() - pw_t0);
    atomicAdd(a.dbg + 3, pw[3]); atomicAdd(a.dbg + 4, 1ull); atomicAdd(a.dbg + 5, pw[5]); atomicAdd(a.dbg + 6, pw[6]);
    atomicAdd(a.dbg + 7, pw[7]);
  }
#endif
#undef W6_CHUNK
#undef W6_BLOCK
#undef W6_MFMA
#undef W6_TICK
#undef W6_ADD
#undef W6_SETUP_UNIT
#undef W6_DMA
#undef W6_A_SLOT
#undef W6_ISSUE_A
#undef W6_ISSUE_W
#undef W6_LOAD_W11
#undef W6_IMG_SCALARS
#undef W6_READ_COL
#undef W6_ROW_T
}


// VAR: 0 = the compiler schedules each chunk freely, 1 = a scheduling barrier after every position block
static inline int launch_v6(const Args& a, int ncu, hipStream_t st, int var) {
  if (a.nsrc < 1 || a.nsrc > 3 || !a.wpack || !a.bias || !a.scale || !a.ovf || !a.zeros || !a.out || a.nchunk < 2) return -1;
  if (a.ntile_n != 2 || a.cout > 64 || (a.cout & 3)) return -6;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + v4::TH4 - 1) / v4::TH4;
  const long long nunits = (long long)a.B * tiles_x * tiles_y;
  const unsigned grid = (unsigned)(nunits < ncu ? nunits : ncu);
  const int res = (a.pre || a.f_w) ? 3 : a.res2 ? 2 : a.res1 ? 1 : 0;
  auto go = [&](auto fn) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, v4::LDS4_BYTES) != hipSuccess) return -2;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), v4::LDS4_BYTES, st, a, (int)nunits);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  };
  if (var == 0) {
    if (res == 0) return go(conv_wino6_kernel<0, 0>);
    if (res == 1) return go(conv_wino6_kernel<1, 0>);
    if (res == 3) return go(conv_wino6_kernel<3, 0>);
    return go(conv_wino6_kernel<2, 0>);
  }
  if (res == 0) return go(conv_wino6_kernel<0, 1>);
  if (res == 1) return go(conv_wino6_kernel<1, 1>);
  if (res == 3) return go(conv_wino6_kernel<3, 1>);
  return go(conv_wino6_kernel<2, 1>);
}
#endif  // __HIPCC__
}  // namespace wino
}  // namespace hcf
