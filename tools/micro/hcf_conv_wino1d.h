// The 1-D Winograd form of the 64-output-channel convolution: F(2,3) along x, direct along y (round 5 experiment, NOT part of the
// product: correct on every check of tools/micro/conv_wino.hip, 8-10 % slower than conv_wino4_kernel; profiles/r05_notes.md section 8).
// Harness versions 8 (8 waves, one 16 x 32 block per CU) and 9 (4 waves, two independent 8 x 32 blocks per CU).
#pragma once
#include "hcf_conv_wino.h"

namespace hcf {
namespace wino {

// 1-D form (round 5, conv_wino1d_kernel): Winograd F(2,3) along x only, direct along y. Per 16-channel chunk THREE phases (one per
// kernel row dy) of 16 pieces of 1 KB: piece ((nu * 2 + ntile) * 2 + plane) = [k-half 2][32 oc][8 halves], u[nu] = sum_dx G[nu][dx]
// g[dy][dx] in double (position 2 negated: the kernel forms V_2 = d1 - d2), planes as everywhere (f16(u) 2^11, f16((u - f16(u)) 2^11)).
constexpr int W1D_PHASE_BYTES = 16384;
static inline bool pack_weights_wino1d(const float* w, int cin, int cout, std::vector<uint16_t>& pk) {
#pragma clang fp contract(off)
  static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  if (cout != 64 || (cin & 15)) return false;
  const int nchunk = cin / 16;
  pk.assign((size_t)nchunk * 3 * (W1D_PHASE_BYTES / 2), 0);
  for (int oc = 0; oc < cout; ++oc)
    for (int ic = 0; ic < cin; ++ic) {
      const float* g = w + ((size_t)oc * cin + ic) * 9;
      const int nt = oc >> 5, n = oc & 31, c = ic >> 4, h = (ic >> 3) & 1, e = ic & 7;
      for (int dy = 0; dy < 3; ++dy)
        for (int nu = 0; nu < 4; ++nu) {
          const double u = (G[nu][0] * g[dy * 3 + 0] + G[nu][1] * g[dy * 3 + 1] + G[nu][2] * g[dy * 3 + 2]) * ((nu == 2) ? -1.0 : 1.0);
          const float x = (float)u;
          if (!(fabsf(x) * 2048.f < 60000.f)) return false;
          const _Float16 hi = (_Float16)x;
          const _Float16 p0 = (_Float16)((float)hi * 2048.f), p1 = (_Float16)((float)((u - (double)(float)hi) * 2048.0));
          const size_t o = (size_t)(c * 3 + dy) * (W1D_PHASE_BYTES / 2) + (size_t)(((nu * 2 + nt) * 2) * 512) + (size_t)h * 256 + (size_t)n * 8 + e;
          memcpy(&pk[o], &p0, 2);
          memcpy(&pk[o + 512], &p1, 2);
        }
    }
  return true;
}


// ---------------------------------------------------------------------------------------------------------------------
// v8 (round 5): the 1-D form -- Winograd F(2,3) along x, direct along y -- for 64 output channels. Why (profiles/r05_notes.md):
// v4's matrix pipe is ~1/3 busy; its time is the non-matrix instruction stream, above all the staging: 102 KB per 16-channel chunk
// and 256-pixel unit through 1-KB LDS-DMA pieces of 60-185 issue cycles each, and 152 VALU per 24 MFMAs for the 2-D transform and
// split. The 1-D form multiplies 1.5x as often (12 instead of 8 position products per 2 x 2 outputs) but
//   * a wave owns ALL four positions of its patches: 8 accumulators cover 32 patches x 2 pixels (not 32 x 4) -> the unit is 16 x 32
//     pixels per block and the weights (12 (dy, nu) x 2 tiles x 2 planes = 48 KB per chunk instead of 64) serve twice the pixels:
//     30 LDS-DMA pieces per 192 MFMAs instead of 102;
//   * the input transform is 4 subtractions per channel and kernel row (no row transform): 80 VALU per 24 MFMAs instead of 152;
//   * the output transform y0 = M0 + M1 + M2, y1 = M1 - M2 - M3 stays inside the wave: no cross-wave exchange, one block barrier
//     per unit in the epilogue (the transpose for 16-byte-lane stores is wave-private).
// tools/micro/wino1d_tile.hip: the phase loop with its DMA pieces runs at 0.72-0.73 of 833 (2-D loop LDS-resident: 0.94, real v4 0.55).
// Structure: 8 waves, wave w = output rows 2 w, 2 w + 1 of a 16 x 32 unit (lane = patch (row li >> 4, column li & 15), k-half), both
// channel tiles. A PHASE = (chunk, kernel row dy): barrier, request the next phase's 16 KB of weights (2 pieces per wave) and a third of
// the next chunk's 39 KB halo image (v3's image layout), read the 4 patch columns of image row (output row + dy), V[nu], split, 24 MFMAs.
#if !defined(W8_ABL)
#define W8_ABL 0       // tools/micro timing ablations only (results are wrong with any bit set): 1 epilogue, 2 image DMA, 4 weight DMA, 8 barrier, 16 transform
#endif
#if !defined(W8_REGSTAGE)
#define W8_REGSTAGE 0  // 1: the loop's image pieces go through registers (buffer_load_dwordx4, ds_write_b128 at the next phase's top) instead of LDS-DMA
#endif
// NW = waves per block: 8 (one block of 16 x 32 pixels per CU) or 4 (8 x 32 pixels, 80 KB of LDS: TWO independent blocks per CU, whose
// staging, transform and epilogue segments overlap each other's matrix segments without any choreography between them).
namespace v8 {
template <int NW> struct Geo {
  static constexpr int TH = 2 * NW, HH = TH + 2;                         // output rows of a unit, halo rows
  static constexpr int IMG_PIECES = (HH * v2::ROWB + 1023) / 1024;       // 39 / 22 LDS-DMA instructions per chunk image (v3's layout)
  static constexpr int IMG_BYTES = IMG_PIECES * 1024;
  static constexpr int NSLOT = (IMG_PIECES + NW - 1) / NW;               // image instructions per wave and chunk: 5 / 6
  static constexpr int WPW = 16 / NW;                                    // weight pieces per wave and phase: 2 / 4
  static constexpr int W_OFF = 2 * IMG_BYTES;                            // two weight slots of one phase (16 KB) each
  static constexpr int X_OFF = W_OFF + 2 * W1D_PHASE_BYTES;              // NW = 8: spare for the epilogue transposes of waves 4..7
  static constexpr int LDS_BYTES = (NW == 8) ? 160 * 1024 : 77 * 1024;   // (NW = 4: 2 x 22 + 2 x 16 KB + tables; two blocks per CU)
  static constexpr int TAB_OFF = LDS_BYTES - 1024;
  static_assert(NW == 8 ? X_OFF + 4 * 8192 <= TAB_OFF : X_OFF <= TAB_OFF, "LDS budget");
};
}  // namespace v8

template <int RES, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void conv_wino1d_kernel(const Args a, const int nunits) {
  using G = v8::Geo<NW>;
  using v2::ROWB;
  using v2::a2_off;
  constexpr int TH2 = G::TH, HH2 = G::HH, IMG_BYTES = G::IMG_BYTES, W8_OFF = G::W_OFF, X8_OFF = G::X_OFF, TAB8_OFF = G::TAB_OFF, NSLOT = G::NSLOT;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, nchunk = a.nchunk, nphase = 3 * nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH2 - 1) / TH2;

  // image DMA slots (v3's): instruction I = NW j + wave, j = 0..NSLOT-1, I < IMG_PIECES
  const int padpix = a.B * H * W;
  int upix[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) upix[j] = padpix;
  uint32_t partpk = 0;
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int pa_ = (NW * j + wave) * 64 + lane, hy_ = (pa_ * 241) >> 15, q_ = pa_ - hy_ * 136, m_ = q_ >> 4;
    partpk |= (uint32_t)(((q_ & 15) ^ (m_ & 7)) & 3) << (4 + 2 * j);
  }
  const int k0 = __builtin_amdgcn_readfirstlane(a.src[0].n >> 4);
  const int k1 = k0 + __builtin_amdgcn_readfirstlane(a.nsrc > 1 ? (a.src[1].n >> 4) : 0);
  const long long npx = (long long)a.B * H * W;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[0].p, 0, (int)(npx * a.src[0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 1 ? 1 : 0].p, 0, (int)(npx * a.src[a.nsrc > 1 ? 1 : 0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 2 ? 2 : 0].p, 0, (int)(npx * a.src[a.nsrc > 2 ? 2 : 0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, nphase * W1D_PHASE_BYTES, 0x00020000);
  const int csb0 = __builtin_amdgcn_readfirstlane(a.src[0].cs) * 4, csb1 = __builtin_amdgcn_readfirstlane(a.src[1].cs) * 4,
            csb2 = __builtin_amdgcn_readfirstlane(a.src[2].cs) * 4;
  const int cb0 = __builtin_amdgcn_readfirstlane(a.src[0].c0) * 4, cb1 = __builtin_amdgcn_readfirstlane(a.src[1].c0) * 4,
            cb2 = __builtin_amdgcn_readfirstlane(a.src[2].c0) * 4;
  const int wvo = lane * 16;

  // cursors: the image of (unit iu, chunk ic) is the next one to request; wp = the next weight phase (chunk * 3 + dy, unit-independent)
  int iu = blockIdx.x, ic = 0, wp = 0;
#define W8_SETUP_UNIT(U)                                                                           \
  {                                                                                                \
    const int v_ = (W8_ABL & 32) ? (xcd_remap((U), nunits) & 7) : xcd_remap((U), nunits);          \
    const int ux0_ = __builtin_amdgcn_readfirstlane((v_ % tiles_x) * TW);                          \
    const int uy0_ = __builtin_amdgcn_readfirstlane(((v_ / tiles_x) % tiles_y) * TH2);             \
    const int ub_ = __builtin_amdgcn_readfirstlane(v_ / (tiles_x * tiles_y));                      \
    int ln_ = lane;                                                                                \
    asm volatile("" : "+v"(ln_));                                                                  \
    _Pragma("unroll") for (int j = 0; j < NSLOT; ++j) {                                            \
      const int pa_ = (NW * j + wave) * 64 + ln_;                                                   \
      const int hy_ = (pa_ * 241) >> 15;                                                           \
      const int q_ = pa_ - hy_ * 136, m_ = q_ >> 4;                                                \
      const int hx_ = m_ * 4 + (((q_ & 15) ^ (m_ & 7)) >> 2);                                      \
      const int y = uy0_ + hy_ - 1, x = ux0_ + hx_ - 1;                                            \
      upix[j] = (hy_ < HH2 && y >= 0 && y < H && x >= 0 && x < W) ? (ub_ * H + y) * W + x : padpix; \
    }                                                                                              \
  }
#define W8_DMA(RS, VOFF, SOFF, DST) __builtin_amdgcn_raw_ptr_buffer_load_lds((RS), (lptr)(DST), 16, (VOFF), (SOFF), 0, 0)
  int csb_ = 0, so_ = 0;
  uint32_t pp_ = partpk;
  __amdgpu_buffer_rsrc_t rsa_ = rs0;
#define W8_IMG_SCALARS()                                                                           \
  {                                                                                                \
    const int sidx_ = (ic < k0) ? 0 : (ic < k1) ? 1 : 2;                                           \
    csb_ = sidx_ == 0 ? csb0 : sidx_ == 1 ? csb1 : csb2;                                           \
    rsa_ = sidx_ == 0 ? rs0 : sidx_ == 1 ? rs1 : rs2;                                              \
    so_ = (sidx_ == 0 ? cb0 + ic * 64 : sidx_ == 1 ? cb1 + (ic - k0) * 64 : cb2 + (ic - k1) * 64); \
    if (W8_ABL & 64) { so_ = (sidx_ == 0 ? ic : sidx_ == 1 ? ic - k0 : ic - k1) * padpix * 64; csb_ = 64; } \
    pp_ = partpk;                                                                                  \
    asm volatile("" : "+v"(pp_));                                                                  \
  }
#define W8_A_SLOT(J, IB)                                                                           \
  {                                                                                                \
    const int p16_ = (int)((pp_ >> (2 * (J))) & 0x30u);                                            \
    const int vo_ = (int)__umul24((unsigned)upix[J], (unsigned)csb_) + p16_;                       \
    W8_DMA(rsa_, vo_, so_, lds + (IB) * IMG_BYTES + (NW * (J) + wave) * 1024);                     \
  }
  u32x4 ireg[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) ireg[j] = u32x4{0, 0, 0, 0};
#define W8_R_SLOT(J)                                                                               \
  {                                                                                                \
    const int p16_ = (int)((pp_ >> (2 * (J))) & 0x30u);                                            \
    const int vo_ = (int)__umul24((unsigned)upix[J], (unsigned)csb_) + p16_;                       \
    ireg[J] = __builtin_amdgcn_raw_buffer_load_b128(rsa_, vo_, so_, 0);                        \
  }
#define W8_LOAD_IMG(J0, J1)                                                                        \
  {                                                                                                \
    _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) {                                       \
      if (NW * j_ + NW - 1 < G::IMG_PIECES) W8_R_SLOT(j_)                                          \
      else if (NW * j_ + wave < G::IMG_PIECES) W8_R_SLOT(j_)                                       \
    }                                                                                              \
  }
#define W8_WRITE_IMG(J0, J1, IB)                                                                   \
  {                                                                                                \
    _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) {                                       \
      if (NW * j_ + NW - 1 < G::IMG_PIECES || NW * j_ + wave < G::IMG_PIECES)                      \
        *reinterpret_cast<u32x4*>(lds + (IB) * IMG_BYTES + (NW * j_ + wave) * 1024 + lane * 16) = ireg[j_]; \
    }                                                                                              \
  }
  // image slots [J0, J1) of the cursor's chunk into image buffer IB
#define W8_ISSUE_IMG(J0, J1, IB)                                                                   \
  {                                                                                                \
    _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) {                                       \
      if (NW * j_ + NW - 1 < G::IMG_PIECES) W8_A_SLOT(j_, IB)                                      \
      else if (NW * j_ + wave < G::IMG_PIECES) W8_A_SLOT(j_, IB)                                   \
    }                                                                                              \
  }
  // the image cursor moves on: next chunk, or chunk 0 of this block's next unit
#define W8_ADVANCE_IMG()                                                                           \
  {                                                                                                \
    if (++ic == nchunk) {                                                                          \
      ic = 0;                                                                                      \
      iu += gridDim.x;                                                                             \
      if (iu < nunits) W8_SETUP_UNIT(iu)                                                           \
      else {                                                                                       \
        _Pragma("unroll") for (int j = 0; j < NSLOT; ++j) upix[j] = padpix;                        \
      }                                                                                            \
    }                                                                                              \
  }
  // the weight cursor's phase (two 1-KB pieces per wave) into weight slot WB, then the cursor moves on
#define W8_ISSUE_W(WB)                                                                             \
  {                                                                                                \
    char* const wb_ = lds + W8_OFF + (WB) * W1D_PHASE_BYTES + wave * (G::WPW * 1024);              \
    const int ws_ = wp * W1D_PHASE_BYTES + wave * (G::WPW * 1024);                                 \
    _Pragma("unroll") for (int k_ = 0; k_ < G::WPW; ++k_) W8_DMA(rsw, wvo, ws_ + k_ * 1024, wb_ + k_ * 1024); \
    if (++wp == nphase) wp = 0;                                                                    \
  }

  // patch reads: lane = patch (row prow of the wave's two output rows, column pcol), k-half: 4 columns x 2 parts of ONE halo row per
  // phase: halo row = 2 wave + prow + dy (output row r of the unit <-> halo rows r .. r + 2)
  const int prow = li >> 4, pcol = li & 15;
  int poff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) poff[j] = a2_off(2 * wave + prow, 2 * pcol + j, 2 * half);
  const int fw = lane * 16;                        // + ((nu * 2 + tile) * 2 + plane) * 1024 inside a weight slot

  if (tid < 64) {
    reinterpret_cast<float*>(lds + TAB8_OFF)[tid] = a.bias[tid] * a.scale[tid];
    reinterpret_cast<float*>(lds + TAB8_OFF)[64 + tid] = a.scale[tid] * UNSPLIT;
  }
  int u = blockIdx.x;
  if (u >= nunits) return;
#if defined(WINO_PROF)
  unsigned long long pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pw_t0 = __builtin_readcyclecounter();
#define W8_T(V) const unsigned long long V = __builtin_readcyclecounter();
#define W8_ACC(I, A, B) pw[I] += (B) - (A);
#else
#define W8_T(V)
#define W8_ACC(I, A, B)
#endif
  // ---- prologue: image of chunk 0 -> buffer 0, weights of phase 0 -> slot 0
  W8_SETUP_UNIT(iu)
  W8_IMG_SCALARS()
  W8_ISSUE_IMG(0, NSLOT, 0)
  W8_ADVANCE_IMG()
  W8_ISSUE_W(0)
  int g = 0, cg = 0;                                // global phase / chunk counters (slot and buffer parities)
  const float slope = a.act == 1 ? 0.f : a.act == 2 ? 0.2f : 1.f;
  const float slope2 = a.act2 == 1 ? 0.f : a.act2 == 2 ? 0.2f : 1.f;

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
      ey0 = __builtin_amdgcn_readfirstlane(((v_ / tiles_x) % tiles_y) * TH2);
      eb = __builtin_amdgcn_readfirstlane(v_ / (tiles_x * tiles_y));
    }
    const int un = u + gridDim.x;

    for (int c = 0; c < nchunk; ++c, ++cg) {
      const char* const ib = lds + (cg & 1) * IMG_BYTES;
      const int ibn = (cg + 1) & 1;
#define W8_MFMA(P, N, WW, VX) acc[P][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WW, __builtin_bit_cast(f16x8, VX), acc[P][N], 0, 0, 0);
#define W8_PHASE(DY)                                                                               \
      {                                                                                            \
        W8_T(q0_)                                                                                  \
        if (W8_REGSTAGE) {                                                                         \
          if ((DY) == 0) { if (g > 0) W8_WRITE_IMG(4, NSLOT, cg & 1) }                             \
          else if ((DY) == 1) { W8_WRITE_IMG(0, 2, ibn) }                                          \
          else { W8_WRITE_IMG(2, 4, ibn) }                                                         \
        }                                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                \
        W8_T(q1_)                                                                                  \
        if (!(W8_ABL & 8)) __builtin_amdgcn_s_barrier();   /* this phase's weights (and, DY 0, this chunk's image) are complete and visible; every wave \
                                           is through the previous phase: its weight slot (DY 0: and the previous image) are free */ \
        W8_T(q2_)                                                                                  \
        W8_ACC(0, q0_, q1_) W8_ACC(1, q1_, q2_)                                                    \
        const char* const wb = lds + W8_OFF + (g & 1) * W1D_PHASE_BYTES + fw;                      \
        if (!(W8_ABL & 4)) W8_ISSUE_W((g + 1) & 1)                                                 \
        if (!(W8_ABL & 2)) {                                                                       \
        if (W8_REGSTAGE) {                                                                         \
          if ((DY) == 0) { W8_IMG_SCALARS() W8_LOAD_IMG(0, 2) }                                    \
          else if ((DY) == 1) { W8_LOAD_IMG(2, 4) }                                                \
          else { W8_LOAD_IMG(4, NSLOT) W8_ADVANCE_IMG() }                                          \
        } else                                                                                     \
        if ((DY) == 0) { W8_IMG_SCALARS() W8_ISSUE_IMG(0, 2, ibn) }                                \
        else if ((DY) == 1) { W8_ISSUE_IMG(2, 4, ibn) }                                            \
        else { W8_ISSUE_IMG(4, NSLOT, ibn) W8_ADVANCE_IMG() }                                      \
        }                                                                                          \
        W8_T(q3_)                                                                                  \
        W8_ACC(5, q2_, q3_)                                                                        \
        float d_[4][8];                                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
          const f32x4 x0_ = *reinterpret_cast<const f32x4*>(ib + poff[j] + (DY) * ROWB);           \
          const f32x4 x1_ = *reinterpret_cast<const f32x4*>(ib + (poff[j] ^ 16) + (DY) * ROWB);    \
          _Pragma("unroll") for (int k = 0; k < 4; ++k) { d_[j][k] = x0_[k]; d_[j][4 + k] = x1_[k]; } \
        }                                                                                          \
        _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) {                                         \
          float v_[8];                                                                             \
          _Pragma("unroll") for (int k = 0; k < 8; ++k)                                            \
            v_[k] = (W8_ABL & 16) ? d_[nu][k] : (nu == 0) ? d_[0][k] - d_[2][k] : (nu == 1) ? d_[1][k] + d_[2][k] : (nu == 2) ? d_[1][k] - d_[2][k] : d_[1][k] - d_[3][k]; \
          u32x4 vh_, vl_;                                                                          \
          split8(v_, vh_, vl_);                                                                    \
          const f16x8 w00 = *reinterpret_cast<const f16x8*>(wb + (nu * 4 + 0) * 1024);             \
          const f16x8 w01 = *reinterpret_cast<const f16x8*>(wb + (nu * 4 + 1) * 1024);             \
          const f16x8 w10 = *reinterpret_cast<const f16x8*>(wb + (nu * 4 + 2) * 1024);             \
          const f16x8 w11 = *reinterpret_cast<const f16x8*>(wb + (nu * 4 + 3) * 1024);             \
          W8_MFMA(nu, 0, w00, vh_)                                                                 \
          W8_MFMA(nu, 1, w10, vh_)                                                                 \
          W8_MFMA(nu, 0, w01, vh_)                                                                 \
          W8_MFMA(nu, 1, w11, vh_)                                                                 \
          W8_MFMA(nu, 0, w00, vl_)                                                                 \
          W8_MFMA(nu, 1, w10, vl_)                                                                 \
        }                                                                                          \
        ++g;                                                                                       \
        W8_T(q4_)                                                                                  \
        W8_ACC(6, q3_, q4_)                                                                        \
      }
      W8_PHASE(0)
      W8_PHASE(1)
      W8_PHASE(2)
#undef W8_PHASE
#undef W8_MFMA
    }

    // ---- epilogue: y0 = M0 + M1 + M2, y1 = M1 - M2 - M3 in registers (lane = 16 channels of ONE patch per tile); one block barrier,
    // then a WAVE-PRIVATE transpose through LDS (the image buffer just consumed for waves 0..3, the spare for waves 4..7): pixel
    // records of 128 bytes (32 channels), 16-byte slots XOR-swizzled with the patch index; after it lane (pixel p of 8, channel quad
    // l & 7) reads 4 channels of pixel 8 k + p, so 8 consecutive lanes cover the 128 contiguous bytes of a pixel's 32 channels.
    {
      W8_T(qe0)
      float chk = 0.f;
      // (NW = 4: waves 0, 1 in the image buffer just consumed, waves 2, 3 in the weight slot just consumed)
      char* const xw = (wave < NW / 2) ? lds + ((cg - 1) & 1) * IMG_BYTES + wave * 8192
                       : (NW == 8)     ? lds + X8_OFF + (wave - 4) * 8192
                                       : lds + W8_OFF + ((g - 1) & 1) * W1D_PHASE_BYTES + (wave - 2) * 8192;
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int hf = lane_e >> 5, pr = (lane_e & 31) >> 4, pc = lane_e & 15;
      const int rp8 = lane_e >> 3, rsl = lane_e & 7;
      __builtin_amdgcn_s_barrier();                 // every wave is done reading the last phase's image rows
      if (W8_ABL & 1) {
        float sacc = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[p][n][r];
        if (sacc == 12345.678f) a.out[lane] = sacc;
      } else
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 y0, y1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            y0[e] = (acc[0][nt][r] + acc[1][nt][r]) + acc[2][nt][r];
            y1[e] = (acc[1][nt][r] - acc[2][nt][r]) - acc[3][nt][r];
          }
          const int P0 = pr * 32 + 2 * pc;          // this lane's first pixel among the wave's 64 (2 rows x 32 columns)
          const int key = (P0 >> 1) & 7;
          *reinterpret_cast<f32x4*>(xw + P0 * 128 + (((2 * q + hf) ^ key) << 4)) = y0;
          *reinterpret_cast<f32x4*>(xw + (P0 + 1) * 128 + (((2 * q + hf) ^ key) << 4)) = y1;
        }
        const int cb = nt * 32 + 4 * rsl;
        const f32x4 bs = *reinterpret_cast<const f32x4*>(lds + TAB8_OFF + cb * 4);
        const f32x4 ms = *reinterpret_cast<const f32x4*>(lds + TAB8_OFF + 256 + cb * 4);
        const bool split_t = (nt == 1) && a.out2 != nullptr;
        const float slope_t = split_t ? slope2 : slope;
#pragma unroll
        for (int hk = 0; hk < 2; ++hk) {            // two halves of 4 pixel groups: bounds the registers the residuals hold
          f32x4 rv1[4], rv2[4];
          size_t pixs[4];
          bool oks[4];
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int P = 8 * (4 * hk + k4) + rp8;
            const int yy = ey0 + 2 * wave + (P >> 5), xx = ex0 + (P & 31);
            oks[k4] = yy < H && xx < W;
            pixs[k4] = (size_t)((size_t)eb * H + (yy < H ? yy : H - 1)) * W + (xx < W ? xx : W - 1);
            if (RES == 1 || RES == 2) rv1[k4] = *reinterpret_cast<const f32x4*>(a.res1 + pixs[k4] * a.res1_cs + a.res1_c0 + cb);
            if (RES == 2) rv2[k4] = *reinterpret_cast<const f32x4*>(a.res2 + pixs[k4] * a.res2_cs + a.res2_c0 + cb);
          }
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int P = 8 * (4 * hk + k4) + rp8;
            const f32x4 yv = *reinterpret_cast<const f32x4*>(xw + P * 128 + ((rsl ^ ((P >> 1) & 7)) << 4));
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              chk = fmaf(yv[e], 0.f, chk);
              const float z = fmaf(yv[e], ms[e], bs[e]);
              v[e] = fmaxf(z, slope_t * z);
              if (RES == 1 || RES == 2) v[e] = fmaf(v[e], a.rs1, rv1[k4][e]);
              if (RES == 2) v[e] = fmaf(v[e], a.rs2, rv2[k4][e]);
            }
            if (oks[k4] && cb < a.cout) {
              if (split_t) *reinterpret_cast<f32x4*>(a.out2 + pixs[k4] * a.out2_cs + a.out2_c0 + (cb - 32)) = v;
              else *reinterpret_cast<f32x4*>(a.out + pixs[k4] * a.out_cs + a.out_c0 + cb) = v;
            }
          }
        }
      }
      if (__any(chk != chk)) {
        if (lane == 0) atomicOr(a.ovf, 1 | (2 << (eb % 30)));     // bit 0 + the unit's sample slot (see hcf_conv_f16x3.hip)
      }
      W8_T(qe1)
      W8_ACC(3, qe0, qe1)
    }
    u = un;
    if (u >= nunits) break;
  }
#if defined(WINO_PROF)
  if (a.dbg && lane == 0 && (blockIdx.x & 31) == 17) {
    atomicAdd(a.dbg + 0, pw[0]); atomicAdd(a.dbg + 1, pw[1]); atomicAdd(a.dbg + 2, __builtin_readcyclecounter() - pw_t0);
    atomicAdd(a.dbg + 3, pw[3]); atomicAdd(a.dbg + 4, 1ull); atomicAdd(a.dbg + 5, pw[5]); atomicAdd(a.dbg + 6, pw[6]);
  }
#endif
#undef W8_T
#undef W8_ACC
#undef W8_SETUP_UNIT
#undef W8_DMA
#undef W8_A_SLOT
#undef W8_R_SLOT
#undef W8_LOAD_IMG
#undef W8_WRITE_IMG
#undef W8_ISSUE_IMG
#undef W8_ADVANCE_IMG
#undef W8_ISSUE_W
#undef W8_IMG_SCALARS
}



static inline int launch_1d(const Args& a, int ncu, hipStream_t st, int version) {
  if (a.nsrc < 1 || a.nsrc > 3 || !a.wpack || !a.bias || !a.scale || !a.ovf || !a.out || a.nchunk < 1) return -1;
  if (a.ntile_n != 2 || a.cout != 64 || a.f_w || a.pre) return -6;
  const int v9 = version == 9;
  const int th = v9 ? 8 : 16;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + th - 1) / th;
  const long long nunits = (long long)a.B * tiles_x * tiles_y;
  const long long nres = v9 ? 2LL * ncu : ncu;              // resident blocks (v9: two 4-wave blocks per CU)
  const unsigned grid = (unsigned)(nunits < nres ? nunits : nres);
  const int res = a.res2 ? 2 : a.res1 ? 1 : 0;
  const int ldsb = v9 ? v8::Geo<4>::LDS_BYTES : v8::Geo<8>::LDS_BYTES;
  auto go8 = [&](auto fn) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) != hipSuccess) return -2;
#if defined(WINO_OCC_PRINT)
    { int nb_ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, fn, v9 ? 256 : 512, ldsb); printf("    [occupancy: %d blocks per CU, %d B LDS]\n", nb_, ldsb); }
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(v9 ? 256 : 512), ldsb, st, a, (int)nunits);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  };
  if (v9) return res == 0 ? go8(conv_wino1d_kernel<0, 4>) : res == 1 ? go8(conv_wino1d_kernel<1, 4>) : go8(conv_wino1d_kernel<2, 4>);
  return res == 0 ? go8(conv_wino1d_kernel<0, 8>) : res == 1 ? go8(conv_wino1d_kernel<1, 8>) : go8(conv_wino1d_kernel<2, 8>);
}

}  // namespace wino
}  // namespace hcf
