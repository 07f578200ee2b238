// fp32-equivalent 3x3 convolution in Winograd F(4x4, 3x3) form on the f16 matrix cores (gfx950), 64 output channels.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 6x6 input patch d -> 4x4 outputs: 36 transformed positions instead of 4 x 16 for
//   the same 16 outputs = 1.78x fewer matrix instructions than hcf_conv_wino.h's F(2x2, 3x3). Every product is the f16x3 split
//   (a_hi w_hi + a_hi w_lo + a_lo w_hi, fp32 accumulate) of the transformed operands: U = G g G^T formed in fp64 on the host and
//   pre-split (both planes x 2^11), V = B^T d B formed in fp32 registers and split there. Full-depth deviation from an fp64
//   evaluation = plain fp32's (tools/winograd_precision_check.py, profiles/r06_winograd_tiles.txt: the round-5 claim that the
//   larger tile is numerically out of reach did not survive the measurement).
//   Matrices: interpolation points 0, +-1, +-2, inf (Lavin & Gray); rows 0, 1, 2, 5 of B^T scaled by 1/4 and of G by 4 (rows 3, 4
//   as published: their pair then costs four instead of five operations per value): |V| <= 36 max|d| (F(2x2): 4), |U| <= 16 max|g|.
//
// Structure (not the F(2x2) kernels' one): a unit of 16 x 32 output pixels = 4 x 8 patches = the 32 columns of the matrix
// instruction; its 36 positions x 2 channel tiles are 72 accumulators of 16 registers = 57 % of the CU's register file, so
//   * 256 threads = 4 waves, ONE per SIMD, 512 registers each; wave (rb, cb) owns the 3 x 3 block of positions
//     xi in {3 rb ..}, nu in {3 cb ..} for BOTH channel tiles: 18 accumulators;
//   * a lane = (patch, k-half) forms its block's V values itself from 5 x 5 of the patch's 6 x 6 pixels (row pass, then column
//     pass, 8 channels per lane = the B fragment of v_mfma_f32_32x32x16_f16): no exchange of V between waves;
//   * with only 32 patches per unit a weight fragment has exactly one consumer in the CU: weights go L2 -> registers
//     (buffer loads, 1 KB per instruction, three positions ahead), never through LDS;
//   * LDS holds the fp32 halo image of a 16-channel chunk (18 x 34 pixels, two buffers, filled by buffer_load ... lds one chunk
//     ahead; 16-byte slots XOR-swizzled so that the patch reads of a 16-lane group hit 16 distinct slots), the accumulators of
//     two of a wave's nine positions (the register file holds 16 of the 18), and, in the epilogue, the exchange of M through
//     which every lane collects the positions of 4 channels of one patch: four rounds of 72 KB (channel tile x transform rows
//     0..2 / 3..5 -- the output transform is linear, the second round adds to the first one's partial outputs); also the
//     transpose that makes 8 consecutive lanes store 128 contiguous bytes of one pixel.
#pragma once
#include "hcf_conv_wino.h"

namespace hcf {
namespace wino6 {

using wino::Args;
using wino::Src;

constexpr int TW = 32, TH = 16, HH = TH + 2;
constexpr int ROWB = 2048;                       // main block: halo columns 0..31, 64 bytes per pixel
constexpr int MAIN_BYTES = HH * ROWB;            // 36 864 = 36 DMA instructions
constexpr int SIDE_OFF = MAIN_BYTES;             // side block: halo columns 32..35 (34, 35 dead), 256 bytes per halo row
constexpr int SIDE_BYTES = 5 * 1024;             // 20 rows (18, 19 dead) = 5 DMA instructions
constexpr int IMG_BYTES = MAIN_BYTES + SIDE_BYTES;   // 41 984
constexpr int NRES = 7;                          // positions whose accumulators stay in registers: 14 x 16 = 224 of the 256 AGPRs; the
                                                 // other two positions' accumulators live in LDS during the chunk loop and pass
                                                 // through the remaining 32 (with all 18 resident the allocator spills one to scratch)
constexpr int ACC8_OFF = 2 * IMG_BYTES;          // 83 968: [position 2][tile 2][reg quad 4][256 threads] x 16 bytes = 64 KB
constexpr int X_OFF = ACC8_OFF;                  // the epilogue's exchange buffer (the accumulator slots are read back first):
constexpr int X_BYTES = 18 * 4096;               // 73 728: HALF a channel tile's M (transform rows 0..2 or 3..5), [pos][q][half][32 patches]
                                                 // x 16 bytes -- outside the image buffers, so the next unit's first image streams in
                                                 // during the last chunk and the epilogue as in steady state
constexpr int TAB_OFF = X_OFF + X_BYTES;         // 157 696: bias * scale, scale * 2^-11
constexpr int PIX_OFF = TAB_OFF + 512;           // 158 208: [5][256 threads] pixel indices of the DMA cursor's unit (kept out of registers:
                                                 // carried through the epilogue they are spilled to scratch, and a scratch reload in
                                                 // front of the DMA issue drains vmcnt -- all prefetched weights -- in every chunk)
constexpr int LDS_BYTES = PIX_OFF + 5 * 1024;    // 163 328 (of 163 840)
constexpr int WPOS_BYTES = 4096;                 // per position: [tile 2][plane 2] fragments of 1 KB ([k-half 2][32 oc][8 halves])
constexpr int WCHUNK_BYTES = 36 * WPOS_BYTES;    // 147 456 per 16-channel chunk
constexpr float UNSPLIT = 1.f / 2048.f;

// w: PyTorch [64][cin][3][3], cin % 16 == 0. U = G g G^T in double; plane 0 = f16(U) * 2^11, plane 1 = f16((U - f16(U)) * 2^11).
static inline bool pack_weights_wino6(const float* w, int cin, int cout, std::vector<uint16_t>& pk) {
#pragma clang fp contract(off)          /* bit-identical to the device-side rebuild */
  static const double G[6][3] = {{1, 0, 0}, {-2.0 / 3, -2.0 / 3, -2.0 / 3}, {-2.0 / 3, 2.0 / 3, -2.0 / 3},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 4}};
  if (cout != 64 || cin < 16 || (cin & 15)) return false;
  const int nchunk = cin / 16;
  pk.assign(((size_t)nchunk + 1) * (WCHUNK_BYTES / 2), 0);
  for (int oc = 0; oc < cout; ++oc)
    for (int ic = 0; ic < cin; ++ic) {
      const float* g = w + ((size_t)oc * cin + ic) * 9;
      double t[6][3], U[6][6];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) U[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
      const int nt = oc >> 5, n = oc & 31, c = ic >> 4, h = (ic >> 3) & 1, e = ic & 7;
      for (int xi = 0; xi < 6; ++xi)
        for (int nu = 0; nu < 6; ++nu) {
          const double u = U[xi][nu];
          const float x = (float)u;
          if (!(fabsf(x) * 2048.f < 60000.f)) return false;
          const _Float16 hi = (_Float16)x;
          const _Float16 p0 = (_Float16)((float)hi * 2048.f), p1 = (_Float16)((float)((u - (double)(float)hi) * 2048.0));
          const size_t o = (size_t)c * (WCHUNK_BYTES / 2) + (size_t)(((xi * 6 + nu) * 2 + nt) * 2) * 512 + (size_t)h * 256 + (size_t)n * 8 + e;
          memcpy(&pk[o], &p0, 2);
          memcpy(&pk[o + 512], &p1, 2);
        }
    }
  return true;
}

#if defined(__HIPCC__)
#if defined(W6_PROF)        // tools/micro only: shader-clock stamps per phase, accumulated per wave, written through Args::dbg
#define W6_T(I) { const unsigned long long t_ = __builtin_readcyclecounter(); pw[I] += t_ - pt; pt = t_; }
#else
#define W6_T(I)
#endif
#ifndef W6_X
#define W6_X 0
#endif
// Guard in front of every position's first matrix instruction. Without it one of this kernel's builds (no residual loads, i.e. a
// slightly different schedule) returned wrong products for the first MFMA after each VALU-only stretch -- the instruction sat
// directly behind the split's run of v_fma_mixhi_f16 ... op_sel writes. An empty asm volatile with a memory clobber (a
// scheduling constraint) is already enough to make it disappear, as is any s_nop; interleaved MFMAs (positions 1, 2 of a row)
// never showed it. Same family as profiles/r02_fault_rootcause.md (op_sel VALU beside f16 MFMAs). tools/micro checks 23 / 24 /
// 36 / 38 / 39 are the regression set.
#ifndef W6_NOPS
#define W6_NOPS 1
#endif
#define W6_STR2(X) #X
#define W6_STR(X) W6_STR2(X)
#if W6_NOPS >= 8
#define W6_GUARD() asm volatile("s_nop 7\n\ts_nop " W6_STR(W6_NOPS - 8) ::: "memory");
#elif W6_NOPS >= 0
#define W6_GUARD() asm volatile("s_nop " W6_STR(W6_NOPS) ::: "memory");
#elif W6_NOPS == -2
#define W6_GUARD() asm volatile("" ::: "memory");
#else
#define W6_GUARD()
#endif
#ifndef W6_ABL
#define W6_ABL 0        // timing-only ablation builds of tools/micro (results wrong by construction): 1 no weight loads in the loop,
#endif                  // 2 no image DMA in the loop, 4 no MFMAs, 8 no patch reads, 16 no output transform / stores, 32 no LDS accumulators
using wino::f32x16;
using wino::f32x4;
using wino::f16x8;
using wino::u32x4;
using wino::lptr;
using wino::split8;
using wino::xcd_remap;

// One half of B^T (rows 0, 1, 2, 5 scaled by 1/4) applied to five consecutive samples s0..s4 = d[BLK .. BLK + 4] of a 6-vector:
//   single: BLK 0: xi 0 = d0 - 1.25 d2 + 0.25 d4;  BLK 1: xi 5 = d1 - 1.25 d3 + 0.25 d5      (s0, s2, s4)
//   pair:   BLK 0: xi 1, 2 = (0.25 d4 - d2) +- (0.25 d3 - d1);  BLK 1: xi 3, 4 = (d4 - d2) +- 2 (d3 - d1)    (d1..d4)
__device__ __forceinline__ float bt_single(float s0, float s2, float s4) { return fmaf(0.25f, s4, fmaf(-1.25f, s2, s0)); }
template <int BLK>
__device__ __forceinline__ void bt_pair(float d1, float d2, float d3, float d4, float& o0, float& o1) {
  if (BLK == 0) {
    const float u = fmaf(0.25f, d4, -d2), v = fmaf(0.25f, d3, -d1);
    o0 = u + v; o1 = u - v;
  } else {
    const float q = d4 - d2, e = d3 - d1;
    o0 = fmaf(2.f, e, q); o1 = fmaf(-2.f, e, q);
  }
}
// f16 hi / lo split of 8 fp32 values: wino::split8 on values made opaque first. Left to the compiler, a value that is itself an fma
// result gets its hi half from a second, mixed-precision copy of that fma (2 instead of 1.5 instructions per value); the
// conversion written as inline asm (v_cvt_pk_f16_f32) instead gave wrong results in one of four builds of this kernel -- an
// instruction the hazard recognizer cannot see next to the matrix instructions -- so the conversion stays the compiler's.
__device__ __forceinline__ void split8p(float (&v)[8], u32x4& h, u32x4& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm("" : "+v"(v[i]));
#if defined(W6_SPLITNOP)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const wino::f16x2 hh = {(_Float16)v[2 * i], (_Float16)v[2 * i + 1]};
    h[i] = __builtin_bit_cast(uint32_t, hh);
    uint32_t lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(h[i]), "v"(v[2 * i]), "v"(v[2 * i + 1]));
    l[i] = lo;
  }
#else
  split8(v, h, l);
#endif
}
// Block barrier that (i) waits for this wave's outstanding LDS operations first -- a ds_read issued before a bare s_barrier may still
// be in flight when another wave, released by the barrier, overwrites its source (the epilogue's exchange rounds) -- and (ii) is a
// compiler-level memory fence (the s_barrier builtin is not: loads may be moved across it).
#define W6_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// local index (0 = the single, 1 / 2 = the pair) -> transform index
__host__ __device__ constexpr int tidx(int blk, int l) { return blk == 0 ? l : (l == 0 ? 5 : 2 + l); }

template <int RES, int RB, int CB>
__device__ __forceinline__ void wave_body(const Args& a, const int nunits, char* const lds, const int wave, const int lane) {
  const int H = a.H, W = a.W, nchunk = a.nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int padpix = a.B * H * W;                // out-of-range pixel index: the DMA writes zeros (conv padding / dead pieces)

  const int k0 = __builtin_amdgcn_readfirstlane(a.src[0].n >> 4);
  const int k1 = k0 + __builtin_amdgcn_readfirstlane(a.nsrc > 1 ? (a.src[1].n >> 4) : 0);
  const long long npx = (long long)a.B * H * W;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[0].p, 0, (int)(npx * a.src[0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 1 ? 1 : 0].p, 0, (int)(npx * a.src[a.nsrc > 1 ? 1 : 0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 2 ? 2 : 0].p, 0, (int)(npx * a.src[a.nsrc > 2 ? 2 : 0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, (nchunk + 1) * WCHUNK_BYTES, 0x00020000);
  const int csb0 = __builtin_amdgcn_readfirstlane(a.src[0].cs) * 4, csb1 = __builtin_amdgcn_readfirstlane(a.src[1].cs) * 4,
            csb2 = __builtin_amdgcn_readfirstlane(a.src[2].cs) * 4;
  const int cb0 = __builtin_amdgcn_readfirstlane(a.src[0].c0) * 4, cb1 = __builtin_amdgcn_readfirstlane(a.src[1].c0) * 4,
            cb2 = __builtin_amdgcn_readfirstlane(a.src[2].c0) * 4;
  const int wvo = lane * 16;
  int* const pixp = reinterpret_cast<int*>(lds + PIX_OFF) + wave * 64 + lane;      // + k * 256

  // ---- DMA roles. Main block: instruction I = wave + 4 k (k = 0..8) = halo row I >> 1 = 2 k + (wave >> 1), column block I & 1 =
  // wave & 1 (16 pixels); lane -> 256-byte row r4 = lane >> 4 (pixels 4 xq .. 4 xq + 3, xq = 4 (wave & 1) + r4), physical slot
  // s = lane & 15 = logical slot ((x & 3) * 4 + part) ^ key, key = (((y >> 2) & 1) << 3) | (xq & 7). The row's key bit is
  // (k >> 1) & 1: static per unrolled k, so a lane needs two column variants (x, x ^ 2) and the row goes through the scalar offset.
  // Side block: instruction t = halo rows 4 t .. 4 t + 3 (wave t; wave 0 also t = 4), lane -> row 4 t + r4, logical slot s ^ ((t & 1) << 3).
  // The five pixel indices of a lane (main row 0 / rows >= 1 in both key variants, two side pieces) are computed per unit from an
  // opaque copy of the lane id and parked in LDS; the 16-byte part of each piece rides in bits 24..25 of the index.
  int ub = 0, uy0 = 0, ux0 = 0, uc = 0;          // DMA cursor (runs one chunk ahead of the matrix loop)
  int pixr0, pix0, pix1, pixs0, pixs1;
#define W6_SETUP_UNIT(U)                                                                           \
  {                                                                                                \
    const int v_ = a.rev ? nunits - 1 - xcd_remap((U), nunits) : xcd_remap((U), nunits);           \
    ux0 = __builtin_amdgcn_readfirstlane((v_ % tiles_x) * TW);                                     \
    uy0 = __builtin_amdgcn_readfirstlane(((v_ / tiles_x) % tiles_y) * TH);                         \
    ub = __builtin_amdgcn_readfirstlane(v_ / (tiles_x * tiles_y));                                 \
    uc = 0;                                                                                        \
    int ln_ = lane;                                                                                \
    asm volatile("" : "+v"(ln_));                                                                  \
    const int r4_ = ln_ >> 4, sl_ = ln_ & 15, xq_ = 4 * (wave & 1) + r4_, lg0_ = sl_ ^ (xq_ & 7);   \
    const int hx0_ = 4 * xq_ + (lg0_ >> 2), hx1_ = 4 * xq_ + ((lg0_ >> 2) ^ 2), part_ = (lg0_ & 3) << 24; \
    const int rowpix_ = (ub * H + uy0) * W;              /* halo row 1 (always inside the image) */ \
    const int gx0_ = ux0 - 1 + hx0_, gx1_ = ux0 - 1 + hx1_;                                        \
    const bool ok0_ = gx0_ >= 0 && gx0_ < W, ok1_ = gx1_ >= 0 && gx1_ < W;                         \
    pix0 = (ok0_ ? rowpix_ + gx0_ : padpix) | part_;                                               \
    pix1 = (ok1_ ? rowpix_ + gx1_ : padpix) | part_;                                               \
    pixr0 = ((ok0_ && uy0 > 0) ? rowpix_ - W + gx0_ : padpix) | part_;                             \
    {                                                                                              \
      const int lgs_ = sl_ ^ ((wave & 1) << 3), sx_ = 32 + (lgs_ >> 2), sy_ = 4 * wave + r4_;      \
      const int gy_ = uy0 - 1 + sy_, gx_ = ux0 - 1 + sx_;                                          \
      pixs0 = ((sx_ < 34 && gy_ >= 0 && gy_ < H && gx_ < W) ? (ub * H + gy_) * W + gx_ : padpix) | ((lgs_ & 3) << 24); \
    }                                                                                              \
    {                                                                                              \
      const int sx_ = 32 + (sl_ >> 2), sy_ = 16 + r4_;                                             \
      const int gy_ = uy0 - 1 + sy_, gx_ = ux0 - 1 + sx_;                                          \
      pixs1 = ((sx_ < 34 && sy_ < HH && gy_ < H && gx_ < W) ? (ub * H + gy_) * W + gx_ : padpix) | ((sl_ & 3) << 24); \
    }                                                                                              \
    pixp[0] = pixr0; pixp[256] = pix0; pixp[512] = pix1; pixp[768] = pixs0; pixp[1024] = pixs1;    \
  }
#define W6_DMA(RS, VOFF, SOFF, DST) __builtin_amdgcn_raw_ptr_buffer_load_lds((RS), (lptr)(DST), 16, (VOFF), (SOFF), 0, 0)
#define W6_VOFF(PIX, CSB) ((int)__umul24((unsigned)(PIX), (unsigned)(CSB)) + (int)(((unsigned)(PIX) >> 24) << 4))
  // image of the cursor's chunk into image buffer IB: 9 main + 1 (wave 0: 2) side instructions per wave
#define W6_ISSUE_A(IB)                                                                             \
  {                                                                                                \
    const int sidx_ = (uc < k0) ? 0 : (uc < k1) ? 1 : 2;                                           \
    const int csb_ = sidx_ == 0 ? csb0 : sidx_ == 1 ? csb1 : csb2;                                 \
    const __amdgpu_buffer_rsrc_t rsa_ = sidx_ == 0 ? rs0 : sidx_ == 1 ? rs1 : rs2;                 \
    const int so_ = (sidx_ == 0 ? cb0 + uc * 64 : sidx_ == 1 ? cb1 + (uc - k0) * 64 : cb2 + (uc - k1) * 64); \
    const int wcsb_ = W * csb_;                                                                    \
    const int vpad_ = (int)__umul24((unsigned)padpix, (unsigned)csb_);                             \
    const int v0_ = W6_VOFF(pix0, csb_), v1_ = W6_VOFF(pix1, csb_), vr_ = W6_VOFF(pixr0, csb_);    \
    char* const ib_ = lds + (IB) * IMG_BYTES;                                                      \
    _Pragma("unroll") for (int k_ = 0; k_ < 9; ++k_) {                                             \
      const int yrel_ = 2 * k_ + (wave >> 1);                                                      \
      const bool yok_ = uy0 - 1 + yrel_ < H;                                                       \
      const int vsel_ = ((k_ >> 1) & 1) ? v1_ : v0_;                                               \
      if (k_ == 0) {                                                                               \
        if (wave < 2) W6_DMA(rsa_, vr_, so_, ib_ + wave * 1024);                                   \
        else W6_DMA(rsa_, yok_ ? v0_ : vpad_, so_, ib_ + wave * 1024);                             \
      } else {                                                                                     \
        W6_DMA(rsa_, yok_ ? vsel_ : vpad_, yok_ ? so_ + (yrel_ - 1) * wcsb_ : so_, ib_ + (wave + 4 * k_) * 1024); \
      }                                                                                            \
    }                                                                                              \
    W6_DMA(rsa_, W6_VOFF(pixs0, csb_), so_, ib_ + SIDE_OFF + wave * 1024);                         \
    if (wave == 0) W6_DMA(rsa_, W6_VOFF(pixs1, csb_), so_, ib_ + SIDE_OFF + 4 * 1024);             \
  }

  // ---- patch reads: pixel (row i, column j) of this lane's patch, the k-half's two 16-byte parts (second part: address ^ 16).
  // Columns CB .. CB + 4; a column in the side block (x >= 32: patch column 7, j >= 4) has a 256-byte row stride.
  // Per chunk: cur[jj] = ca[jj] + the image buffer's LDS address, made opaque so that the address variants (^ 16, ^ 128, + row) are
  // formed next to their reads instead of being hoisted out of the chunk loop into ~40 loop-invariant registers (which spill).
  // (columns j <= 3 are never in the side block: their row stride is the constant ROWB and folds into the instruction's offset)
#define W6_LOAD_PX(D, JJ, I)                                                                       \
  {                                                                                                \
    const int ad_ = (cur[JJ] ^ (((I) >= 4) ? 128 : 0)) + ((CB + (JJ) <= 3) ? (I) * ROWB : ((I) << shs_)); \
    const f32x4 x0_ = *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)(unsigned)ad_;   \
    const f32x4 x1_ = *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)(unsigned)(ad_ ^ 16); \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) { D[k] = x0_[k]; D[4 + k] = x1_[k]; }            \
  }

  // ---- weights: position (xi, nu) of chunk c at byte (c * 36 + xi * 6 + nu) * 4096 of the pack, four fragments of 1 KB
  u32x4 wf[3][4];
#define W6_LOAD_W(SLOT, CHUNK, LP)                                                                 \
  {                                                                                                \
    const int pos_ = tidx(RB, (LP) / 3) * 6 + tidx(CB, (LP) % 3);                                  \
    const int so_ = ((CHUNK) * 36 + pos_) * WPOS_BYTES;                                            \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)                                               \
      wf[SLOT][f_] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wvo, so_ + f_ * 1024, 0);          \
  }

  if (threadIdx.x < 64) {
    reinterpret_cast<float*>(lds + TAB_OFF)[threadIdx.x] = a.bias[threadIdx.x] * a.scale[threadIdx.x];
    reinterpret_cast<float*>(lds + TAB_OFF)[64 + threadIdx.x] = a.scale[threadIdx.x] * UNSPLIT;
  }
  char* const acc8p = lds + ACC8_OFF + threadIdx.x * 16;     // + (position * 8 + tile * 4 + reg quad) * 4096
  {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q_ = 0; q_ < 8 * (9 - NRES); ++q_) *reinterpret_cast<f32x4*>(acc8p + q_ * 4096) = z4;
  }
  int u = blockIdx.x;
  if (u >= nunits) return;
  W6_SETUP_UNIT(u)
  W6_ISSUE_A(0)
  ++uc;
  int g = 0;
#if defined(W6_PROF)
  unsigned long long pw[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pt0 = __builtin_readcyclecounter();
  unsigned long long pt = pt0;
#endif

  while (true) {
    f32x16 acc[NRES][2];
#pragma unroll
    for (int p = 0; p < NRES; ++p)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][n][r] = 0.f;
    const int eb = ub, ey0 = uy0, ex0 = ux0;
    const int un = u + gridDim.x;
    // patch-read constants of this lane, recomputed per unit from an opaque copy of the lane id: their live range is the chunk loop,
    // not the kernel (carried across the epilogue they are spilled to scratch and reloaded -- behind vmcnt(0) -- in every chunk)
    int ca[5], shs;
    {
      int ln_ = lane;
      asm volatile("" : "+v"(ln_));
      const int half_ = ln_ >> 5, pr_ = (ln_ >> 3) & 3, pc_ = ln_ & 7;
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) {
        const int x = 4 * pc_ + CB + jj;
        const bool side = x >= 32;
        const int slot = (((x & 3) * 4 + 2 * half_) ^ (((pr_ & 1) << 3) | ((x >> 2) & 7))) << 4;
        ca[jj] = (side ? SIDE_OFF + 4 * pr_ * 256 : 4 * pr_ * ROWB + (x >> 2) * 256) + slot;
      }
      shs = (pc_ == 7) ? 8 : 11;                 // row stride (as a shift) of the columns j >= 4: 256 bytes in the side block
    }

    for (int c = 0; c < nchunk; ++c, ++g) {
      W6_T(11)
      const bool last = (c + 1 == nchunk);
      // the chunk's first three positions (ring slots 0..2, free since the previous chunk's last row): requested here, first
      // used after the single row pass; nothing is in flight across a unit's epilogue
      if (!(W6_ABL & 1) || g == 0) { W6_LOAD_W(0, c, 0) W6_LOAD_W(1, c, 1) W6_LOAD_W(2, c, 2) }
      // this chunk's image was requested a chunk ago, before every weight load of that chunk; of all loads only the 12 just
      // issued may still be in flight: loads return in order. (The first chunk of a LATER unit: requested during the previous
      // unit's last chunk and waited for in that unit's epilogue, before its first store.)
      if (c > 0 || g == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      W6_T(0)
      if (!last) { pixr0 = pixp[0]; pix0 = pixp[256]; pix1 = pixp[512]; pixs0 = pixp[768]; pixs1 = pixp[1024]; }
      W6_BARRIER();                                 // image g complete and visible; every wave is through chunk g - 1: buffer (g + 1) & 1 is free
      W6_T(1)
      if (last) {
        if (un < nunits) W6_SETUP_UNIT(un)
        else { uc = 0; pixr0 = pix0 = pix1 = pixs0 = pixs1 = padpix; }
      }
      if ((!last || un < nunits) && !(W6_ABL & 2)) { W6_ISSUE_A((g + 1) & 1) ++uc; }
      W6_T(2)
      int cur[5], shs_ = shs;
      {
        const int ibo = (g & 1) * IMG_BYTES + (int)(uintptr_t)(lptr)lds;     // (LDS byte address: the array's base is added once per column)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) { cur[jj] = ca[jj] + ibo; asm volatile("" : "+v"(cur[jj])); }
        asm volatile("" : "+v"(shs_));
      }

#if (W6_ABL & 4)
#define W6_MFMA(ACC, WW, VX) asm volatile("" :: "v"(WW), "v"(VX));
#else
#define W6_MFMA(ACC, WW, VX) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, WW), __builtin_bit_cast(f16x8, VX), ACC, 0, 0, 0);
#endif
      // the three positions of local row LA: column pass of T (five columns), split, six MFMAs each, weight prefetch three positions ahead
#define W6_ROW(LA, T)                                                                              \
  {                                                                                                \
    float v_[3][8];                                                                                \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                \
      v_[0][k] = bt_single(T[0][k], T[2][k], T[4][k]);                                             \
      bt_pair<CB>(T[1 - CB][k], T[2 - CB][k], T[3 - CB][k], T[4 - CB][k], v_[1][k], v_[2][k]);     \
    }                                                                                              \
    _Pragma("unroll") for (int lb = 0; lb < 3; ++lb) {                                             \
      constexpr int lp_ = (LA) * 3;                                                                \
      u32x4 vh_, vl_;                                                                              \
      split8p(v_[lb], vh_, vl_);                                                                   \
      W6_GUARD()                                                                                   \
      if ((lp_ + lb < NRES || (W6_ABL & 32)) && (W6_X & 1)) {                                      \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][1], wf[lb][2], vh_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][0], wf[lb][0], vh_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][1], wf[lb][3], vh_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][0], wf[lb][1], vh_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][1], wf[lb][2], vl_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][0], wf[lb][0], vl_)                           \
      } else if (lp_ + lb < NRES || (W6_ABL & 32)) {                                               \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][0], wf[lb][0], vh_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][1], wf[lb][2], vh_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][0], wf[lb][1], vh_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][1], wf[lb][3], vh_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][0], wf[lb][0], vl_)                           \
        W6_MFMA(acc[lp_ + lb < NRES ? lp_ + lb : 0][1], wf[lb][2], vl_)                           \
      } else {                                                                                     \
        f32x16 a8_[2];                                                                             \
        char* const ap_ = acc8p + (lp_ + lb - NRES) * 32768;                                       \
        _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_)                                           \
          _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                       \
            const f32x4 x_ = *reinterpret_cast<const f32x4*>(ap_ + (n_ * 4 + q_) * 4096);          \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) a8_[n_][4 * q_ + e_] = x_[e_];        \
          }                                                                                        \
        W6_MFMA(a8_[0], wf[lb][0], vh_)                                                            \
        W6_MFMA(a8_[1], wf[lb][2], vh_)                                                            \
        W6_MFMA(a8_[0], wf[lb][1], vh_)                                                            \
        W6_MFMA(a8_[1], wf[lb][3], vh_)                                                            \
        W6_MFMA(a8_[0], wf[lb][0], vl_)                                                            \
        W6_MFMA(a8_[1], wf[lb][2], vl_)                                                            \
        _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_)                                           \
          _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                       \
            f32x4 x_;                                                                              \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) x_[e_] = a8_[n_][4 * q_ + e_];        \
            *reinterpret_cast<f32x4*>(ap_ + (n_ * 4 + q_) * 4096) = x_;                            \
          }                                                                                        \
      }                                                                                            \
      if (!(W6_ABL & 1) && (LA) < 2) { W6_LOAD_W(lb, c, lp_ + lb + 3) }                            \
    }                                                                                              \
  }
      // T columns: index jj = column CB + jj. bt_single uses columns CB, CB + 2, CB + 4 = jj 0, 2, 4; bt_pair columns 1..4 = jj 1 - CB ..
      {
        float ts[5][8];                             // the single row: pixels rows RB, RB + 2, RB + 4
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
          float d0[8], d2[8], d4[8];
          W6_LOAD_PX(d0, jj, RB)
          W6_LOAD_PX(d2, jj, RB + 2)
          W6_LOAD_PX(d4, jj, RB + 4)
#pragma unroll
          for (int k = 0; k < 8; ++k) ts[jj][k] = bt_single(d0[k], d2[k], d4[k]);
        }
        W6_T(3)
        W6_ROW(0, ts)
        W6_T(4)
      }
#if !defined(W6_NOFENCE)
      asm volatile("" ::: "memory");                // (the pair's pixel reads stay behind the single's positions: 80, not 120, live t registers)
#endif
      {
        float tp[2][5][8];                          // the pair: pixel rows 1..4
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
          float d1[8], d2[8], d3[8], d4[8];
          W6_LOAD_PX(d1, jj, 1)
          W6_LOAD_PX(d2, jj, 2)
          W6_LOAD_PX(d3, jj, 3)
          W6_LOAD_PX(d4, jj, 4)
#pragma unroll
          for (int k = 0; k < 8; ++k) bt_pair<RB>(d1[k], d2[k], d3[k], d4[k], tp[0][jj][k], tp[1][jj][k]);
        }
        W6_T(5)
        W6_ROW(1, tp[0])
        W6_T(6)
        W6_ROW(2, tp[1])
        W6_T(7)
      }
#undef W6_ROW
#undef W6_MFMA
    }

    // ---- epilogue --------------------------------------------------------------------------------------------------------
    // Exchange of M in four rounds (channel tile t) x (transform rows 3 h .. 3 h + 2): the two waves that own those rows write
    // their 9 positions of the tile, slot (pos, q, half) x 32 patches (patch slots rotated by 8 q + 4 half: conflict-free
    // 128-bit writes and reads); reader lane (ps, q, hd) of EVERY wave w collects the 18 positions of channels
    // 32 t + 8 q + 4 hd .. + 3 of patch 8 w + ps, forms R[xi][b] = sum_nu M[xi][nu] A[nu][b] and adds A^T[a][xi] R[xi][b] to its
    // 4 x 4 outputs; after h = 1: bias / scale / activation / residuals and 16 stores, 8 consecutive lanes = 128 contiguous bytes.
    float chk = 0.f;
    int lane_e = lane;                              // (opaque: the epilogue's lane constants are recomputed per unit, not carried
    asm volatile("" : "+v"(lane_e));                //  through the chunk loop)
    const int ps = lane_e >> 3, qd = (lane_e >> 1) & 3, hd = lane_e & 1;
    const int patch = 8 * wave + ps, prow = patch >> 3, pcol = patch & 7;
    const int rbase = (qd * 2 + hd) * 512 + ((patch + 8 * qd + 4 * hd) & 31) * 16;
    const int half_e = lane_e >> 5, li_e = lane_e & 31;
    const float slope = a.act == 1 ? 0.f : a.act == 2 ? 0.2f : 1.f;
    const float slope2 = a.act2 == 1 ? 0.f : a.act2 == 2 ? 0.2f : 1.f;
    f32x16 a8[9 - NRES][2];                         // the last positions' accumulators come back from LDS (this thread's own slots)
#pragma unroll
    for (int p_ = 0; p_ < 9 - NRES; ++p_)
#pragma unroll
      for (int n_ = 0; n_ < 2; ++n_)
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_) {
          const f32x4 x_ = *reinterpret_cast<const f32x4*>(acc8p + p_ * 32768 + (n_ * 4 + q_) * 4096);
#pragma unroll
          for (int e_ = 0; e_ < 4; ++e_) a8[p_][n_][4 * q_ + e_] = x_[e_];
        }
    W6_T(11)
#pragma unroll
    for (int nt_ = 0; nt_ < 2; ++nt_) {
#if defined(W6_SWAPT)
      const int nt = 1 - nt_;
#else
      const int nt = nt_;
#endif
      const int cbq = nt * 32 + 8 * qd + 4 * hd;
      // this tile's residuals: requested now, used after the two rounds (the pixel addresses are recomputed at the stores)
#define W6_PIX(OA, OB) ((size_t)((size_t)eb * H + (ey0 + 4 * prow + (OA) < H ? ey0 + 4 * prow + (OA) : H - 1)) * W + (ex0 + 4 * pcol + (OB) < W ? ex0 + 4 * pcol + (OB) : W - 1))
      f32x4 rv1[4][4], rv2[4][4];
#pragma unroll
      for (int oa = 0; oa < 4; ++oa)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
          if (RES >= 1) rv1[oa][ob] = *reinterpret_cast<const f32x4*>(a.res1 + W6_PIX(oa, ob) * a.res1_cs + a.res1_c0 + cbq);
          if (RES == 2) rv2[oa][ob] = *reinterpret_cast<const f32x4*>(a.res2 + W6_PIX(oa, ob) * a.res2_cs + a.res2_c0 + cbq);
        }
      f32x4 Y[4][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        W6_BARRIER();                               // the previous round has been read (h = 0, nt = 0: every wave is done with the
                                                    // last chunk's image and has its own accumulator slots back)
        if (RB == h) {
#pragma unroll
          for (int lp = 0; lp < 9; ++lp) {
            const int p18 = (tidx(RB, lp / 3) - 3 * RB) * 6 + tidx(CB, lp % 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (lp < NRES) ? acc[lp < NRES ? lp : 0][nt][4 * q + e] : a8[lp < NRES ? 0 : lp - NRES][nt][4 * q + e];
              *reinterpret_cast<f32x4*>(lds + X_OFF + p18 * 4096 + q * 1024 + half_e * 512 + ((li_e + 8 * q + 4 * half_e) & 31) * 16) = v;
            }
          }
        }
        W6_BARRIER();
        f32x4 R[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          f32x4 mm[6];
#pragma unroll
          for (int nu = 0; nu < 6; ++nu) mm[nu] = *reinterpret_cast<const f32x4*>(lds + X_OFF + (r * 6 + nu) * 4096 + rbase);
          const f32x4 s1 = mm[1] + mm[2], d1 = mm[1] - mm[2], s2 = mm[3] + mm[4], d2 = mm[3] - mm[4];
          R[r][0] = mm[0] + s1 + s2;
          R[r][1] = d1 + 2.f * d2;
          R[r][2] = s1 + 4.f * s2;
          R[r][3] = d1 + 8.f * d2 + mm[5];
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
          if (h == 0) {                             // xi = 0, 1, 2: A^T columns (1, 0, 0, 0), (1, 1, 1, 1), (1, -1, 1, -1)
            const f32x4 s = R[1][ob] + R[2][ob], d = R[1][ob] - R[2][ob];
            Y[0][ob] = R[0][ob] + s; Y[1][ob] = d; Y[2][ob] = s; Y[3][ob] = d;
          } else {                                  // xi = 3, 4, 5: (1, 2, 4, 8), (1, -2, 4, -8), (0, 0, 0, 1)
            const f32x4 s = R[0][ob] + R[1][ob], d = R[0][ob] - R[1][ob];
            Y[0][ob] += s; Y[1][ob] += 2.f * d; Y[2][ob] += 4.f * s; Y[3][ob] += 8.f * d + R[2][ob];
          }
        }
      }
      W6_T(8)
      const f32x4 bs = *reinterpret_cast<const f32x4*>(lds + TAB_OFF + cbq * 4);
      const f32x4 ms = *reinterpret_cast<const f32x4*>(lds + TAB_OFF + 256 + cbq * 4);
      const bool split_t = (nt == 1) && a.out2 != nullptr;        // second tile routed to its own tensor / activation (fat launches)
      const float slope_t = split_t ? slope2 : slope;
      if (!(W6_ABL & 16)) {
#pragma unroll
        for (int oa = 0; oa < 4; ++oa)
#pragma unroll
          for (int ob = 0; ob < 4; ++ob) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float yv = Y[oa][ob][e];
              chk = fmaf(yv, 0.f, chk);
              const float z = fmaf(yv, ms[e], bs[e]);
              v[e] = fmaxf(z, slope_t * z);
              if (RES >= 1) v[e] = fmaf(v[e], a.rs1, rv1[oa][ob][e]);
              if (RES == 2) v[e] = fmaf(v[e], a.rs2, rv2[oa][ob][e]);
            }
            // every load of this wave -- the next unit's first image (requested during the last chunk) and first weights included --
            // has landed BEFORE the unit's first store is issued: vmcnt counts stores too, a wait at the next chunk's top would sit
            // out the store acknowledgements
            if (nt_ == 0 && oa == 0 && ob == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ey0 + 4 * prow + oa < H && ex0 + 4 * pcol + ob < W && cbq < a.cout) {
              if (split_t) *reinterpret_cast<f32x4*>(a.out2 + W6_PIX(oa, ob) * a.out2_cs + a.out2_c0 + (cbq - 32)) = v;
              else *reinterpret_cast<f32x4*>(a.out + W6_PIX(oa, ob) * a.out_cs + a.out_c0 + cbq) = v;
            }
          }
      } else {
        float s_ = 0.f;
#pragma unroll
        for (int i_ = 0; i_ < 16; ++i_) s_ += Y[i_ >> 2][i_ & 3][0];
        chk += s_ * 0.f;
        if (nt_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      W6_T(9 + nt_)
    }
    W6_BARRIER();                                   // the last round has been read: the accumulator slots are this thread's again
    {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q_ = 0; q_ < 8 * (9 - NRES); ++q_) *reinterpret_cast<f32x4*>(acc8p + q_ * 4096) = z4;
    }
    if (__any(chk != chk)) {
      if (lane == 0) atomicOr(a.ovf, 1 | (2 << (eb % 30)));       // bit 0 + the unit's sample slot (see hcf_conv_f16x3.hip)
    }
    u = un;
    if (u >= nunits) break;
  }
#if defined(W6_PROF)
  if (a.dbg && lane == 0 && (blockIdx.x & 31) == 17) {
    for (int i = 0; i < 12; ++i) atomicAdd(a.dbg + i, pw[i]);
    atomicAdd(a.dbg + 12, __builtin_readcyclecounter() - pt0);
    atomicAdd(a.dbg + 13, 1ull);
  }
#endif
#undef W6_PIX
#undef W6_SETUP_UNIT
#undef W6_DMA
#undef W6_VOFF
#undef W6_ISSUE_A
#undef W6_LOAD_PX
#undef W6_LOAD_W
}

template <int RES>
__global__ __launch_bounds__(256, 1) void conv_wino6_kernel(const Args a, const int nunits) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (four specialisations of one body: the halves of B^T a wave applies differ in form, not just in coefficients)
  if (wave == 0) wave_body<RES, 0, 0>(a, nunits, lds, 0, lane);
  else if (wave == 1) wave_body<RES, 0, 1>(a, nunits, lds, 1, lane);
  else if (wave == 2) wave_body<RES, 1, 0>(a, nunits, lds, 2, lane);
  else wave_body<RES, 1, 1>(a, nunits, lds, 3, lane);
}

static inline int launch(const Args& a, int ncu, hipStream_t st) {
  if (a.nsrc < 1 || a.nsrc > 3 || !a.wpack || !a.bias || !a.scale || !a.ovf || !a.out || a.nchunk < 1) return -1;
  if (a.ntile_n != 2 || a.cout > 64 || (a.cout & 3) || a.pre || a.f_w) return -6;
  int kt = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    if (!a.src[i].p || (a.src[i].n & 15) || (a.src[i].cs & 3) || (a.src[i].c0 & 3) || (reinterpret_cast<uintptr_t>(a.src[i].p) & 15)) return -6;
    if ((long long)a.B * a.H * a.W * a.src[i].cs * 4 >= 0x7f000000LL) return -6;       // 31-bit byte offsets incl. the row term of the scalar offset
    kt += a.src[i].n >> 4;
  }
  if (kt != a.nchunk) return -1;
  if (((a.out_cs | a.out_c0) & 3) || (reinterpret_cast<uintptr_t>(a.out) & 15)) return -6;
  if (a.res1 && (((a.res1_cs | a.res1_c0) & 3) || (reinterpret_cast<uintptr_t>(a.res1) & 15))) return -6;
  if (a.res2 && (((a.res2_cs | a.res2_c0) & 3) || (reinterpret_cast<uintptr_t>(a.res2) & 15))) return -6;
  if (a.res2 && !a.res1) return -1;
  if (a.out2 && (a.res1 || ((a.out2_cs | a.out2_c0) & 3) || (reinterpret_cast<uintptr_t>(a.out2) & 15))) return -6;
  if ((long long)a.B * a.H * a.W >= (1LL << 24)) return -6;                            // 24-bit pixel index (mul24)
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long long nunits = (long long)a.B * tiles_x * tiles_y;
  if (nunits < 1 || nunits > 0x7fffffffLL) return -1;
  const unsigned grid = (unsigned)(nunits < ncu ? nunits : ncu);
  static bool attr_dev[64][3] = {};
  int dev_ = 0;
  if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return -2;
  const int res = a.res2 ? 2 : a.res1 ? 1 : 0;
  auto go = [&](auto fn) {
    if (!attr_dev[dev_][res]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -2;
      attr_dev[dev_][res] = true;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), LDS_BYTES, st, a, (int)nunits);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  };
  if (res == 0) return go(conv_wino6_kernel<0>);
  if (res == 1) return go(conv_wino6_kernel<1>);
  return go(conv_wino6_kernel<2>);
}
#endif  // __HIPCC__

}  // namespace wino6
}  // namespace hcf
