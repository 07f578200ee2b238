// Experiment record (round 3, profiles/r03_notes.md section 9): the 64-output-channel Winograd kernel as a two-group ping-pong.
// NOT used by the library; tools/micro/conv_wino.hip can time it (version 5). Include after hcf_conv_wino.h.
#pragma once
#include "hcf_conv_wino.h"
#if !defined(W4_ABL)
#define W4_ABL 0           // timing ablations (results invalid): 1 no weight DMA in the loop, 2 no image DMA, 4 no MFMAs
#endif
#if defined(__HIPCC__)
namespace hcf {
namespace wino {

// ---- v5: the 64-output-channel kernel as a two-group ping-pong (profiles/r03_notes.md section 9). Same unit (8 x 32 pixels x 64
// channels), same pack, same arithmetic per accumulator as v4 -- what changes is WHEN the two tile groups (waves 0-3 / 4-7, which
// share the SIMDs pairwise) do what. v4 runs both through "read the patch rows" and "position loop" in lockstep: the matrix pipe
// idles during every row-read phase and barrier, and the single-buffered halo image exposes the DMA latency once per chunk. Here
// the block's barriers delimit PHASES in which one group runs its position loop (24 MFMAs per wave) while the other reads its
// patch rows for its next chunk and issues all DMA:
//   phase p (0 .. 2 n), group g, q = p - g:   q even -> DMA duty + rows of chunk q / 2;   q odd -> position loop of chunk q / 2
// Halo images are per group (rows 4 g .. 4 g + 5 of the unit's 10) and rotate through THREE buffers: the image read in phase p
// was requested in phase p - 2 by its own group and waited for at the end of that group's position loop (phase p - 1), so an HBM
// round trip has more than a full phase. The weights of a chunk are consumed in phases 2 c + 1 (group 0) and 2 c + 2 (group 1)
// and double-buffered; to make room the lo plane of channel tile 1 (a quarter of the pack) does not go through LDS: every wave
// fetches its own 4 fragments per chunk straight from L2 into registers during its row phase.
namespace v5 {
constexpr int IMG5 = 13 * 1024;                   // 6 x 34 pixels x 4 parts = 816 pieces of 16 bytes (+ 16 dead)
constexpr int W5_BYTES = 48 * 1024;               // per chunk in LDS: [xi 4][nu 4][(tile 0, hi), (tile 0, lo), (tile 1, hi)] x 1 KB
constexpr int W5_OFF0 = 3 * IMG5;                 // 39 936
constexpr int X5_SPARE = 24 * 1024;               // between the weight buffers: either one + the spare = 72 KB for the epilogue's exchange
constexpr int W5_OFF1 = W5_OFF0 + W5_BYTES + X5_SPARE;
constexpr int TAB5_OFF = W5_OFF1 + W5_BYTES;      // 162 816
constexpr int LDS5_BYTES = TAB5_OFF + 1024;       // 163 840 = all of the CU's LDS
}  // namespace v5

template <int RES>
__global__ __launch_bounds__(512, 1) void conv_wino5_kernel(const Args a, const int nunits) {
  using namespace v5;
  using v2::ROWB;
  using v2::a2_off;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave & 3, tg = wave >> 2;       // transform row xi of tile group tg (output rows 4 tg .. 4 tg + 3)
  const int H = a.H, W = a.W, nchunk = a.nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + v4::TH4 - 1) / v4::TH4;
  constexpr bool F1 = (RES == 3);

  // DMA duty of a wave: pieces xi + 4 j (j = 0..3, < 13) of its OWN group's next image, 6 of the 24 weight pieces of its group's half
  const int padpix = a.B * H * W;                // out-of-range pixel index: the DMA writes zeros (conv padding / dead pieces)
  int upix[4];
  uint32_t partpk = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    upix[j] = padpix;
    const int pa_ = (xi + 4 * j) * 64 + lane, hy_ = (pa_ * 241) >> 15, q_ = pa_ - hy_ * 136, m_ = q_ >> 4;
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

  int ub = 0, uy0 = 0, ux0 = 0;                  // unit of the image cursor
#define W5_SETUP_UNIT(U)                                                                           \
  {                                                                                                \
    const int v_ = xcd_remap((U), nunits);                                                         \
    ux0 = __builtin_amdgcn_readfirstlane((v_ % tiles_x) * TW);                                     \
    uy0 = __builtin_amdgcn_readfirstlane(((v_ / tiles_x) % tiles_y) * v4::TH4);                    \
    ub = __builtin_amdgcn_readfirstlane(v_ / (tiles_x * tiles_y));                                 \
    int ln_ = lane;                                                                                \
    asm volatile("" : "+v"(ln_));                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
      const int pa_ = (xi + 4 * j) * 64 + ln_;                                                     \
      const int hy_ = (pa_ * 241) >> 15;                                                           \
      const int q_ = pa_ - hy_ * 136, m_ = q_ >> 4;                                                \
      const int hx_ = m_ * 4 + (((q_ & 15) ^ (m_ & 7)) >> 2);                                      \
      const int y = uy0 + 4 * tg + hy_ - 1, x = ux0 + hx_ - 1;                                     \
      upix[j] = (hy_ < 6 && y >= 0 && y < H && x >= 0 && x < W) ? (ub * H + y) * W + x : padpix;   \
    }                                                                                              \
  }
#define W5_DMA(RS, VOFF, SOFF, DST) __builtin_amdgcn_raw_ptr_buffer_load_lds((RS), (lptr)(DST), 16, (VOFF), (SOFF), 0, 0)
  // this group's image of chunk CC (of the cursor unit) into image buffer IB
#define W5_ISSUE_IMG(CC, IB)                                                                       \
  {                                                                                                \
    const int cc_ = (CC);                                                                          \
    const int sidx_ = (cc_ < k0) ? 0 : (cc_ < k1) ? 1 : 2;                                         \
    const int csb_ = sidx_ == 0 ? csb0 : sidx_ == 1 ? csb1 : csb2;                                 \
    const __amdgpu_buffer_rsrc_t rsa_ = sidx_ == 0 ? rs0 : sidx_ == 1 ? rs1 : rs2;                 \
    const int so_ = (sidx_ == 0 ? cb0 + cc_ * 64 : sidx_ == 1 ? cb1 + (cc_ - k0) * 64 : cb2 + (cc_ - k1) * 64); \
    uint32_t pp_ = partpk;                                                                         \
    asm volatile("" : "+v"(pp_));                                                                  \
    char* const ib_ = lds + (IB) * IMG5;                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
      if ((j < 3 || xi == 0) && !(W4_ABL & 2)) {                                                   \
        const int p16_ = (int)((pp_ >> (2 * j)) & 0x30u);                                          \
        const int vo_ = (int)__umul24((unsigned)upix[j], (unsigned)csb_) + p16_;                   \
        W5_DMA(rsa_, vo_, so_, ib_ + (xi + 4 * j) * 1024);                                         \
      }                                                                                            \
    }                                                                                              \
  }
  // pieces K0 .. K1 - 1 (of 6) of this wave's share of half tg of chunk CC's weights (LDS pieces 24 tg + 6 xi + k) into weight buffer WB
#define W5_ISSUE_W(CC, WB, K0, K1)                                                                 \
  {                                                                                                \
    char* const wb_ = lds + ((WB) ? W5_OFF1 : W5_OFF0);                                            \
    const int ws_ = (CC) * W4_BYTES;                                                               \
    _Pragma("unroll") for (int k = (K0); k < (K1); ++k) {                                          \
      const int l_ = 24 * tg + 6 * xi + k;              /* LDS piece: xi' * 12 + nu * 3 + s */      \
      const int x_ = l_ / 12, r_ = l_ - 12 * x_, n_ = r_ / 3, s_ = r_ - 3 * n_;                    \
      const int g_ = x_ * 16 + n_ * 4 + s_;             /* pack piece: xi' * 16 + (nu * 2 + tile) * 2 + plane */ \
      if (!(W4_ABL & 1)) W5_DMA(rsw, wvo, ws_ + g_ * 1024, wb_ + l_ * 1024);                       \
    }                                                                                              \
  }

  // patch reads (rows relative to the group's image): transform row xi uses patch rows (rA, rB) = (0,2) (1,2) (1,2) (1,3)
  const int trow = li >> 4, tcol = li & 15;
  const int rA = (xi == 0) ? 0 : 1, rB = (xi == 3) ? 3 : 2;
  const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (xi == 1) ? 1.f : -1.f)));
  int poff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) poff[j] = a2_off(2 * trow, 2 * tcol + j, 2 * half);
  const int offA = rA * ROWB, offB = rB * ROWB;
  const int fw = lane * 16 + xi * (12 * 1024);   // + (nu * 3 + s) * 1024
  const char* const w11p = a.wpack + (size_t)(xi * 16 + 3) * 1024;                 // (wave-uniform) + chunk * W4_BYTES + nu * 4096 + lane * 16

  if (tid < 64) {
    reinterpret_cast<float*>(lds + TAB5_OFF)[tid] = a.bias[tid] * a.scale[tid];
    reinterpret_cast<float*>(lds + TAB5_OFF)[64 + tid] = a.scale[tid] * UNSPLIT;
    if (F1) {
      reinterpret_cast<float*>(lds + TAB5_OFF)[128 + tid] = a.f_bias[tid] * a.f_scale[tid];
      reinterpret_cast<float*>(lds + TAB5_OFF)[192 + tid] = a.f_scale[tid] * UNSPLIT;
    }
  }
  int u = blockIdx.x;
  if (u >= nunits) return;
  const float slope = a.act == 1 ? 0.f : a.act == 2 ? 0.2f : 1.f;
  const float slope2 = a.act2 == 1 ? 0.f : a.act2 == 2 ? 0.2f : 1.f;
  const float slope_f = a.f_act == 1 ? 0.f : a.f_act == 2 ? 0.2f : 1.f;

  // prologue: this group's image of the first unit's chunk 0 (buffer tg), all of chunk 0's weights (both halves by every group's
  // waves would be twice the pieces: each group loads its half, as in the steady state)
  int rbuf = tg;                                  // image buffer this group reads next; its following image goes to (rbuf + 2) % 3
  W5_SETUP_UNIT(u)
  W5_ISSUE_IMG(0, rbuf)
  W5_ISSUE_W(0, 0, 0, 6)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int gbase = 0;                                  // global chunk counter at the start of the unit (weight buffer = (gbase + c) & 1)
#if defined(WINO_PROF)
  unsigned long long pw5[5] = {0, 0, 0, 0, 0};
#endif

  while (true) {
    f32x16 acc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][n][r] = 0.f;
    const int eb = ub, ey0 = uy0, ex0 = ux0;
    const int un = u + gridDim.x;
    float t_[4][8];                               // row-transformed patch values of the group's coming position loop
    f16x8 w11[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) { t_[j][k] = 0.f; w11[j][k] = (_Float16)0.f; }
    bool img = false;                             // this group's last row phase requested an image

    // ---- position loop of chunk C_: the V values and their f16 split in the shadow of the 24 matrix instructions
#define W5_MFMA(P, N, WW, VX) if (!(W4_ABL & 4)) acc[P][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WW, __builtin_bit_cast(f16x8, VX), acc[P][N], 0, 0, 0); else { acc[P][N][0] += (float)(WW)[0] * (float)__builtin_bit_cast(f16x8, VX)[0]; }
#define W5_VSPLIT(NU, VH, VL)                                                                      \
  {                                                                                                \
    float v_[8];                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 8; ++k)                                                  \
      v_[k] = ((NU) == 0) ? t_[0][k] - t_[2][k] : ((NU) == 1) ? t_[1][k] + t_[2][k] : ((NU) == 2) ? t_[1][k] - t_[2][k] : t_[1][k] - t_[3][k]; \
    split8(v_, VH, VL);                                                                            \
  }
#define W5_POS(C_) {                                                                                                                                \
      const int cp_ = (C_);                                                                                                      \
      const char* const wb = lds + (((gbase + cp_) & 1) ? W5_OFF1 : W5_OFF0) + fw;                                               \
      const int cwn_ = (cp_ + 1 == nchunk) ? 0 : cp_ + 1;                                                                        \
      /* the lo-plane fragments fetched in the row phase: everything issued after them is the image */                           \
      if (!img) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
      else if (xi == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                         \
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                                      \
      u32x4 vh_[2], vl_[2];                                                                                                      \
      W5_VSPLIT(0, vh_[0], vl_[0])                                                                                               \
      _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) {                                                                         \
        /* a lone wave has to interleave by hand: the next position's V values and split ride between this position's MFMAs */   \
        if (nu < 3) W5_VSPLIT(nu + 1, vh_[(nu + 1) & 1], vl_[(nu + 1) & 1])                                                      \
        const f16x8 w00 = *reinterpret_cast<const f16x8*>(wb + (nu * 3 + 0) * 1024);                                             \
        const f16x8 w01 = *reinterpret_cast<const f16x8*>(wb + (nu * 3 + 1) * 1024);                                             \
        const f16x8 w10 = *reinterpret_cast<const f16x8*>(wb + (nu * 3 + 2) * 1024);                                             \
        W5_MFMA(nu, 0, w00, vh_[nu & 1])                                                                                         \
        W5_MFMA(nu, 1, w10, vh_[nu & 1])                                                                                         \
        W5_MFMA(nu, 0, w01, vh_[nu & 1])                                                                                         \
        W5_MFMA(nu, 1, w11[nu], vh_[nu & 1])                                                                                     \
        W5_MFMA(nu, 0, w00, vl_[nu & 1])                                                                                         \
        W5_MFMA(nu, 1, w10, vl_[nu & 1])                                                                                         \
        /* group 0 requests its half of the next chunk's weights from inside its matrix phase: they have until the end of */     \
        /* its next row phase; group 1 requests its half in its row phase and has until the end of this phase */                 \
        if (tg == 0 && nu < 3) W5_ISSUE_W(cwn_, (gbase + cp_ + 1) & 1, 2 * nu, 2 * nu + 2)                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                                                       \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                     \
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                                     \
        }                                                                                                                        \
      }                                                                                                                          \
      /* the image requested in the row phase before has landed (group 0: its 6 weight pieces may stay in flight) */             \
      if (tg == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                              \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                      \
    }
    // ---- row phase: the lo-plane fragments of tile 1 for the position loop of chunk C_ (< 0: no rows in this phase), group 1's
    // half of the next chunk's weights, the group's image of chunk C_ + 1, then the patch rows of C_ and their row transform
#define W5_ROWS(C_) {                                                                                                                                                                \
      const int c_ = (C_);                                                                                                                                       \
      const bool rows = c_ >= 0;                                                                                                                                 \
      if (rows) {                                                                                                                                                \
        const gcptr wp_ = uniform_ptr(w11p + (size_t)c_ * W4_BYTES);                                                                                             \
        _Pragma("unroll") for (int nu = 0; nu < 4; ++nu)      /* (inline asm: the compiler's own vmcnt bookkeeping would wait for ALL DMA at the first use) */   \
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(w11[nu]) : "v"(wvo), "s"(wp_ + nu * 4096) : "memory");                                           \
      }                                                                                                                                                          \
      if (tg == 1 && rows) {                                                                                                                                     \
        const int cw = (c_ + 1 == nchunk) ? 0 : c_ + 1;                                                                                                          \
        W5_ISSUE_W(cw, (gbase + c_ + 1) & 1, 0, 6)                                                                                                               \
      }                                                                                                                                                          \
      img = false;                                                                                                                                               \
      if (rows) {                                                                                                                                                \
        int ci = c_ + 1;                                                                                                                                         \
        img = true;                                                                                                                                              \
        if (ci == nchunk) {                                                                                                                                      \
          ci = 0;                                                                                                                                                \
          if (un < nunits) W5_SETUP_UNIT(un)                                                                                                                     \
          else img = false;                                                                                                                                      \
        }                                                                                                                                                        \
        if (img) W5_ISSUE_IMG(ci, rbuf >= 1 ? rbuf - 1 : 2)                                                                                                      \
      }                                                                                                                                                          \
      if (rows) {                                                                                                                                                \
        const char* const ib = lds + rbuf * IMG5;                                                                                                                \
        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {     /* four channels at a time: the register file is full */                                          \
          f32x4 xa_[4], xb_[4];                                                                                                                                  \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                                                        \
            xa_[j] = *reinterpret_cast<const f32x4*>(ib + (poff[j] ^ (16 * hh)) + offA);                                                                         \
            xb_[j] = *reinterpret_cast<const f32x4*>(ib + (poff[j] ^ (16 * hh)) + offB);                                                                         \
          }                                                                                                                                                      \
          _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                                                          \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) t_[j][4 * hh + k] = fmaf(sg, xb_[j][k], xa_[j][k]);                                                    \
          __builtin_amdgcn_sched_barrier(0);                                                                                                                     \
        }                                                                                                                                                        \
        rbuf = rbuf >= 1 ? rbuf - 1 : 2;          /* (rbuf + 2) % 3 */                                                                                           \
      }                                                                                                                                                          \
      /* the patch rows are in registers. Group 0: the weight pieces of its last matrix phase are older than the fragments and  */                               \
      /* the image pieces just requested; group 1 waits for nothing here (its next matrix phase ends with a full wait)          */                               \
      if (tg == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                                            \
      else if (!rows) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                                                \
      else if (!img) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                                                                 \
      else if (xi == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                                                              \
      else asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");                                                                                           \
    }
    // phases p = 0 .. 2 n, one barrier each:  group 0: rows(0) | pos(0) | rows(1) | pos(1) | ... | pos(n-1) | duty
    //                                         group 1:    -    | rows(0) | pos(0) | rows(1) | ...          | pos(n-1)
#if defined(WINO_PROF)
#define W5_T(I) { const unsigned long long t__ = __builtin_readcyclecounter(); pw5[I] += t__ - tp5; tp5 = t__; }
    unsigned long long tp5 = __builtin_readcyclecounter();
#else
#define W5_T(I)
#endif
    if (tg == 0) {
      W5_ROWS(0)
      W5_T(0)
      __builtin_amdgcn_s_barrier();
      W5_T(1)
      for (int c = 0; c < nchunk; ++c) {
        W5_POS(c)
        W5_T(2)
        __builtin_amdgcn_s_barrier();
        W5_T(3)
        W5_ROWS(c + 1 < nchunk ? c + 1 : -1)
        W5_T(0)
        __builtin_amdgcn_s_barrier();
        W5_T(1)
      }
    } else {
      __builtin_amdgcn_s_barrier();
      W5_T(1)
      for (int c = 0; c < nchunk; ++c) {
        W5_ROWS(c)
        W5_T(0)
        __builtin_amdgcn_s_barrier();
        W5_T(1)
        W5_POS(c)
        W5_T(2)
        __builtin_amdgcn_s_barrier();
        W5_T(3)
      }
    }
    gbase += nchunk;
    int lane_e = lane;                            // (opaque: the epilogue's lane-constant addresses are recomputed per unit, not kept
    asm volatile("" : "+v"(lane_e));              //  in registers through the chunk loop, where there are none to spare)
    wino64_epilogue<RES>(a, lds, TAB5_OFF, lds + (((gbase - 1) & 1) ? W5_OFF0 + W5_BYTES : W5_OFF0), acc, eb, ey0, ex0, H, W, slope, slope2, slope_f, wave, lane_e);
    W5_T(4)
    u = un;
    if (u >= nunits) break;                       // (the exchange area is refilled in phase 1 at the earliest: after phase 0's barrier)
  }
#if defined(WINO_PROF)
  // per group (waves 0 / 4 of a few blocks): [0] row phases [1] barrier after them [2] position loops [3] barrier after them [4] epilogue [5] samples
  if (a.dbg && lane == 0 && xi == 0 && (blockIdx.x & 31) == 17) {
    for (int i = 0; i < 5; ++i) atomicAdd(a.dbg + 8 * tg + i, pw5[i]);
    atomicAdd(a.dbg + 8 * tg + 5, 1ull);
  }
#endif
#undef W5_T
#undef W5_SETUP_UNIT
#undef W5_DMA
#undef W5_ISSUE_IMG
#undef W5_ISSUE_W
#undef W5_MFMA
#undef W5_VSPLIT
#undef W5_POS
#undef W5_ROWS
}


static inline int launch_v5(const Args& a, int ncu, hipStream_t st) {
  if (a.nsrc < 1 || a.nsrc > 3 || !a.wpack || !a.bias || !a.scale || !a.ovf || !a.out || a.nchunk < 2 || a.ntile_n != 2 || a.cout != 64) return -1;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + v4::TH4 - 1) / v4::TH4;
  const long long nunits = (long long)a.B * tiles_x * tiles_y;
  const unsigned grid = (unsigned)(nunits < ncu ? nunits : ncu);
  const int res = a.f_w ? 3 : a.res2 ? 2 : a.res1 ? 1 : 0;
  auto go = [&](auto fn) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, v5::LDS5_BYTES) != hipSuccess) return -2;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), v5::LDS5_BYTES, st, a, (int)nunits);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  };
  if (res == 0) return go(conv_wino5_kernel<0>);
  if (res == 1) return go(conv_wino5_kernel<1>);
  if (res == 3) return go(conv_wino5_kernel<3>);
  return go(conv_wino5_kernel<2>);
}

}  // namespace wino
}  // namespace hcf
#endif
