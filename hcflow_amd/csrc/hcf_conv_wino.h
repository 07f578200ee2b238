// fp32-equivalent 3x3 convolution in Winograd F(2x2, 3x3) form on the f16 matrix cores (gfx950).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 4x4 input patch d -> 2x2 outputs, summed over input channels in the
//   transformed domain: 16 independent [32 oc x 16 ic] x [16 ic x 32 patches] products per chunk -> 2.25x fewer MFMAs than the
//   direct form. Every product is the f16x3 split of hcf_conv_f16x3.hip (a_hi w_hi + a_hi w_lo + a_lo w_hi, fp32 accumulate),
//   applied to the TRANSFORMED operands: U = G g G^T is formed in fp64 on the host and pre-split, V = B^T d B is formed in
//   fp32 registers (entries 0, +-1: three additions) and split there. On the full-depth nets the deviation from an fp64
//   evaluation equals plain fp32's (tools/winograd_precision_check.py: 3.1e-6 / 3.7e-6 / 4.4e-6 for x4 / x8 / rescaling).
//
// Why: profiles/r02_notes.md -- under dense random-data f16 MFMA this part delivers ~1.5 PFLOP/s whatever the tiling, i.e. the
// direct form is capped at ~0.60 of the nominal 833 TFLOP/s-equivalent before any memory traffic; the only way past that is
// fewer MFMAs per output. The price is VALU work (transform + split: ~3 instructions per transformed value) and SIMD issue
// slots, which is what bounds this kernel (tools/micro/wino_tile.hip).
//
// Structure: 512 threads = 8 waves, ONE persistent block per CU; unit = 8 output rows x 32 columns x 32 output channels.
//   wave (xi, tg): transform row xi = 0..3 of tile group tg = 0 / 1 (4 output rows x 32 columns = 32 patches = the MFMA N);
//   it reads the 2 patch rows x 4 columns its row of B^T touches, forms V[xi][0..3] for 8 channels per lane (the MFMA B
//   fragment layout: lane = (patch, k-half)), and accumulates M[xi][nu] in 4 accumulators. Operand roles are swapped as in
//   tools/micro/hcf_conv_s16.h (A = weights): a lane ends up with 16 output channels of ONE patch.
//   LDS: two stages of {10 x 34 halo pixels x 16 channels fp32 (16-byte slots XOR-swizzled inside each 256-byte bank row:
//   conflict-free patch reads), 32 KB of transformed weights [pos][plane][k-half][32 oc][8]}, filled by global_load_lds
//   one chunk ahead (the next unit's first chunk during the last chunk of the current one); one barrier per chunk.
//   Epilogue: R[xi][b] = sum_nu M[xi][nu] A[nu][b] in registers, one exchange of R through LDS (64 KB), then wave
//   (a, q-pair) = (xi >> 1, xi & 1) forms Y[a][b] = sum_xi A^T[a][xi] R[xi][b] for 16 channels x 2 pixels per lane and stores
//   16-byte vectors (bias / scale / activation / residuals as in the other conv kernels).
// Sign folding: the wave computes t_j = d[r1][j] + sigma d[r2][j] and V_2 = t_1 - t_2, i.e. rows / columns 2 negated; the pack
// negates U[2][.] and U[.][2] accordingly, so M is the true product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include <vector>

namespace hcf {
namespace wino {

constexpr int TW = 32, TH = 8, HWP = TW + 2, HHP = TH + 2;
constexpr int A_REAL_PIECES = HHP * HWP * 4;      // 1 360 16-byte pieces (340 pixels x 4 parts)
constexpr int A_BYTES = 22 * 1024;                // 22 DMA instructions (48 dead pieces)
constexpr int W_BYTES = 16 * 2 * 2 * 32 * 16;     // 32 768: [pos][plane][k-half][32 oc][8 halves]
constexpr int STAGE = A_BYTES + W_BYTES;          // 55 296 = 54 DMA instructions
constexpr int X_BYTES = 10240;                    // with the free stage: the 64 KB exchange buffer of the epilogue
constexpr int S0_OFF = 0, X_OFF = STAGE, S1_OFF = STAGE + X_BYTES, TAB_OFF = S1_OFF + STAGE;
constexpr int LDS_BYTES = TAB_OFF + 512;          // 121 344
constexpr float UNSPLIT = 1.f / 2048.f;

struct Src { const float* p; int cs, c0, n; };   // fp32 NHWC window [c0, c0 + n), n % 16 == 0, cs % 4 == 0, c0 % 4 == 0
struct Args {
  Src src[3];
  int nsrc;
  int B, H, W;
  const char* wpack;       // [ntile_n][nchunk][W_BYTES] (+ one zero chunk), pack_weights_wino
  int nchunk, ntile_n;
  const float* bias;       // [32 * ntile_n]
  const float* scale;
  int act;                 // 0 none, 1 relu, 2 leaky relu 0.2
  float* out; int out_cs, out_c0, cout;
  const float* res1; int res1_cs, res1_c0; float rs1;      // y = res2 + rs2 * (res1 + rs1 * act((acc + bias) * scale))
  const float* res2; int res2_cs, res2_c0; float rs2;
  // "fat" dense-block launches (hcf_engine.hip run_rdb): the 64-channel kernel may route its second 32-channel tile to another
  // tensor with its own activation (conv k's outputs + the old-input part of conv k+1 as ONE launch), and the 32-channel kernel
  // may add a stored partial sum BEFORE bias / activation (conv k+1's completion): pre != null -> res1 / res2 are unused.
  float* out2; int out2_cs, out2_c0, act2;                  // 64-channel kernel: the second 32-channel tile goes here with activation act2
                                                            // (null: both tiles go to `out`). 32-channel kernel (round 4, 16-channel dense
                                                            // blocks): channels [out2_split, 32) of the tile go here, out2_split = 16
  int out2_split;
  const float* pre; int pre_cs, pre_c0;                     // 32-channel kernel only
  // fused 1x1 second layer (FCN conv1 -> conv2 of the conditional coupling nets, Basic.py:441-447; hcf_engine.hip
  // run_coupling_net): 64-channel kernel only. The first layer's tile (after bias / scale / activation) is split and parked in
  // LDS as the B operand of a 64 x 64 product: out = act_f((W_f h1 + f_bias) * f_scale); res1 / res2 / out2 are unused.
  const char* f_w;                                          // pack_weights_1x1_frag: 16 KB
  const float* f_bias; const float* f_scale; int f_act;
  // Data-gradient form (training: hcf_engine_train.inc bwd_conv; the SC template variants): the input is a gradient tensor, multiplied
  // by a power of two derived from *in_max (its max |x|, device) inside the f16 split -- gradients of 1e-8 would otherwise vanish
  // in the f16 planes -- and the sums are scaled back in the epilogue (as hcf_conv_f16x3.hip's SCALED variants).
  const float* in_max;
  // 32-channel kernel, SC, no residuals: fused epilogue backward of the conv whose dL/dy this launch completes (ConvArgs::fb_y):
  // y = that conv's forward output; stores dL/dpre = dL/dy * act'(y), leaves one row of per-channel sums per BLOCK in fb_part
  // ([gridDim.x][2][cout]: row 0 = sum dL/dpre, row 1 = zeros) and the maxima |dL/dpre| in fb_max / max(that, *in_max) in fb_max2.
  const float* fb_y; int fb_y_cs, fb_y_c0;
  int fb_act;              // 0 none, 1 relu, 2 leaky relu 0.2 (of the conv whose output y is)
  float* fb_part; float* fb_max; float* fb_max2;
  int* ovf;
  const char* zeros;       // >= 64 bytes of zeros
  int rev;                 // walk the units in reverse order (the engine alternates per launch: the consumer starts on what the producer touched last, which the 256 MB MALL still holds)
  int top_wait;            // A/B knob (HCF_WINO_TOP_WAIT=1): the 64-channel kernel waits at the top of every unit's first chunk as before round 5
  unsigned long long* dbg; // WINO_PROF builds: [0] vmcnt wait [1] barrier wait [2] life [3] epilogue [4] samples [5] setup+issue [6] loads+transform
  unsigned long long* clk; // in-kernel clock probe of the 64-channel kernel (hcf_debug_clock_probe): block 0 adds its life in shader
                           // cycles (s_memtime) to clk[0] and in 100 MHz ticks (s_memrealtime) to clk[1] ([2], [3]: its start stamps); null: off
};

// w: PyTorch [cout][cin][3][3] (cin = sum of the source widths, each a multiple of 16). U = G g G^T in double, rows and
// columns 2 negated, plane 0 = f16(U) * 2^11, plane 1 = f16((U - f16(U)) * 2^11).
static inline bool pack_weights_wino(const float* w, int cin, int cout, std::vector<uint16_t>& pk) {
#pragma clang fp contract(off)          /* bit-identical to the device-side rebuild (hcf_conv_wino.hip: repack_wino_kernel) */
  static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  const int nchunk = cin / 16, ntn = (cout + 31) / 32;
  pk.assign(((size_t)ntn * nchunk + 1) * (W_BYTES / 2), 0);
  for (int oc = 0; oc < cout; ++oc)
    for (int ic = 0; ic < cin; ++ic) {
      const float* g = w + ((size_t)oc * cin + ic) * 9;
      double t[4][3], U[4][4];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) U[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
      const int nt = oc >> 5, n = oc & 31, c = ic >> 4, h = (ic >> 3) & 1, e = ic & 7;
      for (int xi = 0; xi < 4; ++xi)
        for (int nu = 0; nu < 4; ++nu) {
          const double u = U[xi][nu] * ((xi == 2) ? -1.0 : 1.0) * ((nu == 2) ? -1.0 : 1.0);
          const float x = (float)u;
          if (!(fabsf(x) * 2048.f < 60000.f)) return false;
          const _Float16 hi = (_Float16)x;
          const _Float16 p0 = (_Float16)((float)hi * 2048.f), p1 = (_Float16)((float)((u - (double)(float)hi) * 2048.0));
          const size_t o = ((size_t)(nt * nchunk + c) * W_BYTES) / 2 + (size_t)(((xi * 4 + nu) * 2 + 0) * 2 + h) * 256 + (size_t)n * 8 + e;
          memcpy(&pk[o], &p0, 2);
          memcpy(&pk[o + 512], &p1, 2);
        }
    }
  return true;
}

// 64-output-channel layout of the v4 kernel: per 16-channel chunk 64 pieces of 1 KB, piece ((pos * 2 + ntile) * 2 + plane) =
// [k-half 2][32 oc][8 halves]; same arithmetic as pack_weights_wino.
constexpr int W4_BYTES = 65536;
static inline bool pack_weights_wino64(const float* w, int cin, int cout, std::vector<uint16_t>& pk) {
#pragma clang fp contract(off)
  static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  if (cout != 64 || (cin & 15)) return false;
  const int nchunk = cin / 16;
  pk.assign(((size_t)nchunk + 1) * (W4_BYTES / 2), 0);
  for (int oc = 0; oc < cout; ++oc)
    for (int ic = 0; ic < cin; ++ic) {
      const float* g = w + ((size_t)oc * cin + ic) * 9;
      double t[4][3], U[4][4];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) U[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
      const int nt = oc >> 5, n = oc & 31, c = ic >> 4, h = (ic >> 3) & 1, e = ic & 7;
      for (int xi = 0; xi < 4; ++xi)
        for (int nu = 0; nu < 4; ++nu) {
          const double u = U[xi][nu] * ((xi == 2) ? -1.0 : 1.0) * ((nu == 2) ? -1.0 : 1.0);
          const float x = (float)u;
          if (!(fabsf(x) * 2048.f < 60000.f)) return false;
          const _Float16 hi = (_Float16)x;
          const _Float16 p0 = (_Float16)((float)hi * 2048.f), p1 = (_Float16)((float)((u - (double)(float)hi) * 2048.0));
          const size_t o = (size_t)c * (W4_BYTES / 2) + (size_t)(((xi * 4 + nu) * 2 + nt) * 2) * 512 + (size_t)h * 256 + (size_t)n * 8 + e;
          memcpy(&pk[o], &p0, 2);
          memcpy(&pk[o + 512], &p1, 2);
        }
    }
  return true;
}

// 1x1 64 -> 64 layer fused into the 64-channel kernel's epilogue: the A fragments of v_mfma_f32_32x32x16_f16 in lane order,
// piece ((m-tile * 4 + k-step) * 2 + plane) = [64 lanes][8 halves], lane (half, li) = output channel 32 mt + li, input channels
// 16 s + 8 half .. + 7; planes as everywhere: f16(w) * 2^11 and f16((w - f16(w)) * 2^11). w: [64][64] (PyTorch [cout][cin][1][1]).
constexpr int F1_BYTES = 16384;
static inline bool pack_weights_1x1_frag(const float* w, std::vector<uint16_t>& pk) {
  pk.assign(F1_BYTES / 2, 0);
  for (int mt = 0; mt < 2; ++mt)
    for (int ks = 0; ks < 4; ++ks)
      for (int ln = 0; ln < 64; ++ln)
        for (int e = 0; e < 8; ++e) {
          const int oc = mt * 32 + (ln & 31), ic = 16 * ks + 8 * (ln >> 5) + e;
          const float x = w[oc * 64 + ic];
          if (!(fabsf(x) * 2048.f < 60000.f)) return false;
          const _Float16 hi = (_Float16)x;
          const _Float16 p0 = (_Float16)((float)hi * 2048.f), p1 = (_Float16)((x - (float)hi) * 2048.f);
          const size_t o = ((size_t)((mt * 4 + ks) * 2 + 0) * 64 + ln) * 8 + e;
          memcpy(&pk[o], &p0, 2);
          memcpy(&pk[o + 512], &p1, 2);
        }
  return true;
}

#if defined(__HIPCC__)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef const char __attribute__((address_space(1)))* gcptr;
typedef __attribute__((address_space(3))) void* lptr;

__device__ __forceinline__ int xcd_remap(int orig, int n) {
  const int xcd = orig & 7, q = n >> 3, r = n & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}
__device__ __forceinline__ gcptr uniform_ptr(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gcptr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void glds16(gcptr g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lptr)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ float act1(float v, float slope, float lo) { return !(v <= 0.f) ? v : slope * fmaxf(v, lo); }
// byte offset of (halo pixel px, 16-byte part) inside the activation region
__device__ __forceinline__ int a_off(int px, int part) {
  const int R = px >> 2, sl = (px & 3) * 4 + part;
  return R * 256 + ((sl ^ ((R & 7) << 1)) << 4);
}


// ---------------------------------------------------------------------------------------------------------------------
// v3: 16 x 32-pixel unit, 8 waves = 4 tile groups x 2 transform-row pairs, TWO waves per SIMD (256 registers per lane: the 8
// position accumulators of a wave sit in the AGPR half). What the one-wave-per-SIMD experiment (v2, profiles/r02_notes.md)
// showed: every instruction of a lone wave costs ~5 cycles of issue, an LDS-DMA piece 60-100, and nothing overlaps unless it
// is hand-interleaved; with a partner wave on the SIMD the hardware interleaves the two streams by itself. Versus v1: the
// 32 KB weight chunk is shared by 512 pixels (not 256), conflict-free 128-bit LDS reads (the XOR key covers all four slot
// bits: a 16-lane group of a ds_read_b128 sees 16 distinct slots), the f16 hi / lo split costs 1.5 instructions per value
// (v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16) instead of 3.5, no packed-fp32 VALU (slow beside MFMAs), staging by
// buffer_load ... lds with an SGPR resource and 32-bit offsets (conv zero padding = an out-of-range offset: the hardware
// writes zeros; no per-lane 64-bit address arithmetic).
namespace v2 {
constexpr int TH2 = 16, HH2 = TH2 + 2;
constexpr int A2_REAL = HH2 * HWP * 4;            // 2 448 pieces
constexpr int A2_BYTES = 39 * 1024;               // 39 DMA instructions (48 dead pieces)
constexpr int STAGE2 = A2_BYTES + W_BYTES;        // 72 704 = 71 instructions
constexpr int TAB2_OFF = 2 * STAGE2;
constexpr int LDS2_BYTES = TAB2_OFF + 512;        // 145 920
constexpr int FB2_OFF = LDS2_BYTES;               // SC variants: per-thread channel-quad sums of the fused epilogue backward (8 KB) + the block's max
constexpr int LDS2S_BYTES = FB2_OFF + 512 * 16 + 64;      // 154 176
constexpr int ROWB = HWP * 64;                    // 2 176 bytes per halo row
// byte offset of (halo row, column x, 16-byte part): rows are affine (row * ROWB), the swizzle depends on the column only, so a
// lane needs ONE base register per patch column and reaches the patch rows through immediates / a scalar row offset
__host__ __device__ constexpr int a2_off(int row, int x, int part) {
  return row * ROWB + (x >> 2) * 256 + (((((x & 3) * 4 + part)) ^ ((x >> 2) & 7)) << 4);
}
}  // namespace v2

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// f16 hi / lo split of 8 fp32 values: hi = RNE f16 pairs (v_cvt_pk_f16_f32), lo = f16(v - hi) by the mixed-precision FMA (one
// instruction per value: reads hi as f16 and v as f32, writes the f16 result into one half of the destination register).
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& h, u32x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 hh = {(_Float16)v[2 * i], (_Float16)v[2 * i + 1]};
    h[i] = __builtin_bit_cast(uint32_t, hh);
    uint32_t lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h[i]), "v"(v[2 * i]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(h[i]), "v"(v[2 * i + 1]));
    l[i] = lo;
  }
}

// the same split of v * s, s a power of two (SC variants): hi = f16(v s) and lo = f16(v s - hi) by the mixed-precision FMA alone
// (2 instructions per value instead of 1.5: the scaling is free)
__device__ __forceinline__ void split8s(const float (&v)[8], const float s, u32x4& h, u32x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t hi, lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(hi) : "v"(v[2 * i]), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hi) : "v"(v[2 * i + 1]), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(v[2 * i]), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v[2 * i + 1]), "v"(s), "v"(hi));
    h[i] = hi;
    l[i] = lo;
  }
}
// x * in_s lands in [2^9, 2^10] at the tensor's max |x| (the transformed values reach 4x that: 2^12 << 65504)
__device__ __forceinline__ void input_scales(const float* in_max, float& in_s, float& out_s, int& in_max_bits) {
  in_s = 1.f; out_s = 1.f;
  in_max_bits = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *in_max));
  const float mx = __builtin_bit_cast(float, in_max_bits);
  if (mx > 0.f && mx < 3.0e38f) {
    int ex = 0;
    (void)frexpf(mx, &ex);                       // mx = m * 2^ex, m in [0.5, 1)
    in_s = ldexpf(1.f, 10 - ex);
    out_s = ldexpf(1.f, ex - 10);
  }
  in_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, in_s)));
  out_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, out_s)));
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const f32x4& v, u32x2& h, u32x2& l) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f16x2 hh = {(_Float16)v[2 * i], (_Float16)v[2 * i + 1]};
    h[i] = __builtin_bit_cast(uint32_t, hh);
    uint32_t lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h[i]), "v"(v[2 * i]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(h[i]), "v"(v[2 * i + 1]));
    l[i] = lo;
  }
}

template <int RES, bool SC = false>
__global__ __launch_bounds__(512, 1) void conv_wino2_kernel(const Args a, const int nunits) {
  using namespace v2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xp = wave & 1, tg = wave >> 1;       // transform rows 2 xp, 2 xp + 1 of tile group tg (output rows 4 tg .. 4 tg + 3)
  const int H = a.H, W = a.W, ntn = a.ntile_n, nchunk = a.nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH2 - 1) / TH2;

  // DMA slots: instruction I = 8 j + wave (j = 0..8, I < 71); I < 39 activation pieces, else weight piece I - 39.
  // j < 4: activations; j = 4: activations (wave 7: weight piece 0); j = 5..7: weights; j = 8: weights (wave 7: none).
  const bool w7 = (wave == 7);
  // global pixel index of activation slot j; padding / dead pieces: B H W, whose byte offset is exactly the end of the
  // tensor for every source -> out of range -> the DMA writes zeros (no select on the issue path)
  const int padpix = a.B * H * W;
  int upix[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) upix[j] = padpix;
  // piece (8 j + wave) * 64 + lane sits in 256-byte row R = piece >> 4 at physical slot lane & 15; its logical slot (pixel in
  // row, 16-byte part) = physical ^ key(R). The part bits of all j are packed into one register, at bits 4 + 2 j.
  uint32_t partpk = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int pa_ = (8 * j + wave) * 64 + lane, hy_ = (pa_ * 241) >> 15, q_ = pa_ - hy_ * 136, m_ = q_ >> 4;
    partpk |= (uint32_t)(((q_ & 15) ^ (m_ & 7)) & 3) << (4 + 2 * j);
  }
  const int k0 = __builtin_amdgcn_readfirstlane(a.src[0].n >> 4);
  const int k1 = k0 + __builtin_amdgcn_readfirstlane(a.nsrc > 1 ? (a.src[1].n >> 4) : 0);
  const long long npx = (long long)a.B * H * W;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[0].p, 0, (int)(npx * a.src[0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 1 ? 1 : 0].p, 0, (int)(npx * a.src[a.nsrc > 1 ? 1 : 0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[a.nsrc > 2 ? 2 : 0].p, 0, (int)(npx * a.src[a.nsrc > 2 ? 2 : 0].cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, (ntn * nchunk + 1) * W_BYTES, 0x00020000);
  const int csb0 = __builtin_amdgcn_readfirstlane(a.src[0].cs) * 4, csb1 = __builtin_amdgcn_readfirstlane(a.src[1].cs) * 4,
            csb2 = __builtin_amdgcn_readfirstlane(a.src[2].cs) * 4;
  const int cb0 = __builtin_amdgcn_readfirstlane(a.src[0].c0) * 4, cb1 = __builtin_amdgcn_readfirstlane(a.src[1].c0) * 4,
            cb2 = __builtin_amdgcn_readfirstlane(a.src[2].c0) * 4;
  const int wvo = lane * 16;                     // weight piece q covers bytes [q * 1024, + 1024) of the chunk

  int ub = 0, uy0 = 0, ux0 = 0, unt = 0, uc = 0;   // DMA cursor
  // (the lane index goes through an empty asm so that the per-slot row / column arithmetic is redone here, once per unit,
  //  instead of being hoisted into loop-invariant registers that then spill)
#define W2_SETUP_UNIT(U)                                                                           \
  {                                                                                                \
    const int v_ = a.rev ? nunits - 1 - xcd_remap((U), nunits) : xcd_remap((U), nunits);           \
    unt = __builtin_amdgcn_readfirstlane(v_ % ntn);                                                \
    const int t_ = v_ / ntn;                                                                       \
    ux0 = __builtin_amdgcn_readfirstlane((t_ % tiles_x) * TW);                                     \
    uy0 = __builtin_amdgcn_readfirstlane(((t_ / tiles_x) % tiles_y) * TH2);                        \
    ub = __builtin_amdgcn_readfirstlane(t_ / (tiles_x * tiles_y));                                 \
    uc = 0;                                                                                        \
    int ln_ = lane;                                                                                \
    asm volatile("" : "+v"(ln_));                                                                  \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                \
      const int pa_ = (8 * j + wave) * 64 + ln_;           /* piece = row * 136 + m * 16 + physical slot */ \
      const int hy_ = (pa_ * 241) >> 15;                   /* pa / 136 for pa < 2700 */             \
      const int q_ = pa_ - hy_ * 136, m_ = q_ >> 4;                                                \
      const int hx_ = m_ * 4 + (((q_ & 15) ^ (m_ & 7)) >> 2);                                      \
      const int y = uy0 + hy_ - 1, x = ux0 + hx_ - 1;                                              \
      upix[j] = (hy_ < HH2 && y >= 0 && y < H && x >= 0 && x < W) ? (ub * H + y) * W + x : padpix; \
    }                                                                                              \
  }
#define W2_DMA(RS, VOFF, SOFF, DST) __builtin_amdgcn_raw_ptr_buffer_load_lds((RS), (lptr)(DST), 16, (VOFF), (SOFF), 0, 0)
  // per-chunk scalars of the DMA cursor (source selected per chunk, wave-uniform; scalar selects: no control flow around a DMA)
  int csb_ = 0, so_ = 0, ws_ = 0;
  uint32_t pp_ = partpk;
  __amdgpu_buffer_rsrc_t rsa_ = rs0;
#define W2_CHUNK_SCALARS()                                                                         \
  {                                                                                                \
    const int sidx_ = (uc < k0) ? 0 : (uc < k1) ? 1 : 2;                                           \
    csb_ = sidx_ == 0 ? csb0 : sidx_ == 1 ? csb1 : csb2;                                           \
    rsa_ = sidx_ == 0 ? rs0 : sidx_ == 1 ? rs1 : rs2;                                              \
    so_ = (sidx_ == 0 ? cb0 + uc * 64 : sidx_ == 1 ? cb1 + (uc - k0) * 64 : cb2 + (uc - k1) * 64); \
    ws_ = (unt * nchunk + uc) * W_BYTES;                                                           \
    pp_ = partpk;                                                                                  \
    asm volatile("" : "+v"(pp_));           /* keeps the per-slot field extraction out of loop-invariant registers */ \
  }
#define W2_A_SLOT(J, DST)                                                                          \
  {                                                                                                \
    const int p16_ = (int)((pp_ >> (2 * (J))) & 0x30u);                                            \
    const int vo_ = (int)__umul24((unsigned)upix[J], (unsigned)csb_) + p16_;                       \
    W2_DMA(rsa_, vo_, so_, DST);                                                                   \
  }
  // slots [J0, J1) of the cursor's chunk -> stage STG
#define W2_ISSUE(J0, J1, STG)                                                                      \
  {                                                                                                \
    char* const sb_ = lds + (STG) * STAGE2 + wave * 1024;                                          \
    _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) {                                       \
      if (j_ < 4) W2_A_SLOT(j_ < 5 ? j_ : 0, sb_ + j_ * 8192)                                      \
      else if (j_ == 4) { if (!w7) W2_A_SLOT(4, sb_ + 4 * 8192) else W2_DMA(rsw, wvo, ws_, sb_ + 4 * 8192); } \
      else if (j_ < 8) W2_DMA(rsw, wvo, ws_ + (8 * j_ + wave - 39) * 1024, sb_ + j_ * 8192);       \
      else if (!w7) W2_DMA(rsw, wvo, ws_ + (8 * 8 + wave - 39) * 1024, sb_ + 8 * 8192);            \
    }                                                                                              \
  }

  // fragment read offsets: patch pixel (row i, column j) of this lane's patch, part 2 half (the other part: ^ 16)
  const int trow = li >> 4, tcol = li & 15;
  int poff[4];                                   // patch column j, patch row 0; row i: + i * ROWB
#pragma unroll
  for (int j = 0; j < 4; ++j) poff[j] = a2_off(4 * tg + 2 * trow, 2 * tcol + j, 2 * half);
  // transform rows of this wave: xp = 0: t = r0 - r2, then r1 + r2; xp = 1: t = r1 - r2 (sign folded into the weights), then r1 - r3
  const int rowP = xp ? ROWB : 0, rowS = xp ? 3 * ROWB : 2 * ROWB;     // first: row P - row 2; second: row 1 + sg * row S
  const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xp ? -1.f : 1.f)));
  const int fw = A2_BYTES + half * 512 + li * 16 + xp * (8 * 4 * 512);   // + ((local pos * 2 + plane) * 2) * 512

  float in_s = 1.f, out_s = 1.f;
  int in_max_bits = 0;
  if (SC) input_scales(a.in_max, in_s, out_s, in_max_bits);
  constexpr bool FBK = SC && RES == 0;           // the fused epilogue backward exists in this variant
  const bool fb = FBK && a.fb_y != nullptr;
  if (tid < 64) {
    const float sc_ = (tid < 32 * ntn) ? a.scale[tid] : 1.f, bi_ = (tid < 32 * ntn) ? a.bias[tid] : 0.f;
    reinterpret_cast<float*>(lds + TAB2_OFF)[tid] = bi_ * sc_;
    reinterpret_cast<float*>(lds + TAB2_OFF)[64 + tid] = sc_ * UNSPLIT * out_s;
  }
  if (FBK) {
    if (fb) {
      *reinterpret_cast<f32x4*>(lds + FB2_OFF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tid == 0) *reinterpret_cast<int*>(lds + FB2_OFF + 512 * 16) = 0;
    }
  }
  int u = blockIdx.x;
  if (u >= nunits) return;
#if defined(WINO_PROF)
  unsigned long long pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pw_t0 = __builtin_readcyclecounter();
#endif
  W2_SETUP_UNIT(u)
  W2_CHUNK_SCALARS()
  W2_ISSUE(0, 9, 0)
  ++uc;
  int g = 0;
  const float slope = a.act == 1 ? 0.f : a.act == 2 ? 0.2f : 1.f;

  while (true) {
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
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
      pw[0] += q1 - q0; pw[1] += q2 - q1;
#endif
      if (c + 1 == nchunk) {
        if (un < nunits) W2_SETUP_UNIT(un)
        else {
          uc = 0; unt = 0;
#pragma unroll
          for (int j = 0; j < 5; ++j) upix[j] = padpix;
        }
      }
      W2_CHUNK_SCALARS()
      const char* const sb = lds + stg * STAGE2;
      const int so = stg ^ 1;
#define W2_LOAD_ROW(D, OFF)                                                                        \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
    const f32x4 x0_ = *reinterpret_cast<const f32x4*>(sb + poff[j] + (OFF));                       \
    const f32x4 x1_ = *reinterpret_cast<const f32x4*>(sb + (poff[j] ^ 16) + (OFF));                \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) { D[j][k] = x0_[k]; D[j][4 + k] = x1_[k]; }      \
  }
#define W2_V(NU, T, K) (((NU) == 0) ? T[0][K] - T[2][K] : ((NU) == 1) ? T[1][K] + T[2][K] : ((NU) == 2) ? T[1][K] - T[2][K] : T[1][K] - T[3][K])
#define W2_MFMA(P, WW, VX) acc[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WW, __builtin_bit_cast(f16x8, VX), acc[P], 0, 0, 0);
      // the four positions of local transform row I (t in T)
#define W2_XI(I, T)                                                                                \
  _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) {                                               \
    float v_[8];                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) v_[k] = W2_V(nu, T, k);                          \
    u32x4 vh_, vl_;                                                                                \
    if (SC) split8s(v_, in_s, vh_, vl_); else split8(v_, vh_, vl_);                                \
    const f16x8 w1_ = *reinterpret_cast<const f16x8*>(sb + fw + (((I) * 4 + nu) * 4) * 512);       \
    const f16x8 w2_ = *reinterpret_cast<const f16x8*>(sb + fw + (((I) * 4 + nu) * 4 + 2) * 512);   \
    W2_MFMA((I) * 4 + nu, w1_, vh_)                                                                \
    W2_MFMA((I) * 4 + nu, w2_, vh_)                                                                \
    W2_MFMA((I) * 4 + nu, w1_, vl_)                                                                \
  }
      float t_[4][8];
      {
        float ra[4][8], rc[4][8];
        W2_LOAD_ROW(ra, rowP)
        W2_LOAD_ROW(rc, 2 * ROWB)
        W2_ISSUE(0, 3, so)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 8; ++k) t_[j][k] = ra[j][k] - rc[j][k];
      }
      W2_XI(0, t_)
      W2_ISSUE(3, 6, so)
      {
        float ra[4][8], rc[4][8];
        W2_LOAD_ROW(ra, ROWB)
        W2_LOAD_ROW(rc, rowS)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 8; ++k) t_[j][k] = fmaf(sg, rc[j][k], ra[j][k]);
      }
      W2_ISSUE(6, 9, so)
      ++uc;
      W2_XI(1, t_)
#if defined(WINO_PROF)
      pw[6] += __builtin_readcyclecounter() - q2;
#endif
#undef W2_XI
#undef W2_MFMA
#undef W2_V
#undef W2_LOAD_ROW
    }

    // ---- epilogue -----------------------------------------------------------------------------------------------------------
    // In-wave: R[i][b] = sum_nu M[i][nu] A[nu][b]; the pair's partial outputs: xp 0: y0 = R0 + R1, y1 = R1; xp 1 (rows 2, 3):
    // y0 = R0', y1 = -R0' - R1'. The two waves of a tile group meet in LDS (the stage just consumed), one output row (a) per
    // round, and the exchange is also a TRANSPOSE: after it lane (patch p of 8, register group q, half) holds the 4 channels
    // 8 q + 4 half .. + 3 of one patch from both waves, so 8 consecutive lanes cover the 128 contiguous bytes of a pixel's 32
    // channels (residual loads and stores touch 8 cache lines per instruction instead of 32). Wave xp finishes patches
    // 16 xp .. 16 xp + 15 of the tile group.
#if defined(WINO_PROF)
    const unsigned long long qe0 = __builtin_readcyclecounter();
#endif
    float chk = 0.f;                               // Inf / NaN anywhere in the accumulators reaches a partial output
    f32x4 fbs = {0.f, 0.f, 0.f, 0.f};              // fused epilogue backward: this unit's sums of the lane's channel quad, max |dL/dpre|
    float fbm = 0.f;
    const float fb_neg = a.fb_act == 1 ? 0.f : a.fb_act == 2 ? 0.2f : 1.f;
    {
      char* const xbuf = lds + ((g - 1) & 1) * STAGE2 + tg * 16384;
      const int p8 = lane >> 3, qd = (lane >> 1) & 3, hd = lane & 1;            // reader role
      const int wpos = (half * 32 + ((li + 4 * half) & 31)) * 16;               // writer slot for even q; odd q: + 8 patches (mod 32)
      const int wpos1 = (half * 32 + ((li + 4 * half + 8) & 31)) * 16;
      const int cb = ent * 32 + 8 * qd + 4 * hd;
      const f32x4 bs = *reinterpret_cast<const f32x4*>(lds + TAB2_OFF + cb * 4);
      const f32x4 ms = *reinterpret_cast<const f32x4*>(lds + TAB2_OFF + 256 + cb * 4);
      // fat launch of a 16-channel dense block: the upper half of the tile is the NEXT conv's partial sum, stored raw elsewhere
      const bool second = (RES == 0) && a.out2 != nullptr && cb >= a.out2_split;
      const float slope_l = second ? (a.act2 == 1 ? 0.f : a.act2 == 2 ? 0.2f : 1.f) : slope;
      __builtin_amdgcn_s_barrier();                // every wave is done reading the stage
#pragma unroll
      for (int oa = 0; oa < 2; ++oa) {
        // this round's residuals: issued now, used after the exchange
        f32x4 rv1[2][2], rv2[2][2];
        size_t pixs[2][2];
        bool oks[2][2];
        int rpos[2];
#pragma unroll
        for (int sgrp = 0; sgrp < 2; ++sgrp) {
          const int patch = 16 * xp + 8 * sgrp + p8, prow = patch >> 4, pcol = patch & 15;
          rpos[sgrp] = (hd * 32 + ((patch + 4 * hd + 8 * (qd & 1)) & 31)) * 16 + qd * 1024;
#pragma unroll
          for (int ob = 0; ob < 2; ++ob) {
            const int yy = ey0 + 4 * tg + 2 * prow + oa, xx = ex0 + 2 * pcol + ob;
            oks[sgrp][ob] = yy < H && xx < W;
            pixs[sgrp][ob] = (size_t)((size_t)eb * H + (yy < H ? yy : H - 1)) * W + (xx < W ? xx : W - 1);
            if (RES == 1 || RES == 2) rv1[sgrp][ob] = *reinterpret_cast<const f32x4*>(a.res1 + pixs[sgrp][ob] * a.res1_cs + a.res1_c0 + cb);
            if (RES == 2) rv2[sgrp][ob] = *reinterpret_cast<const f32x4*>(a.res2 + pixs[sgrp][ob] * a.res2_cs + a.res2_c0 + cb);
            if (RES == 3) rv1[sgrp][ob] = *reinterpret_cast<const f32x4*>(a.pre + pixs[sgrp][ob] * a.pre_cs + a.pre_c0 + cb);
            if (FBK) { if (fb) rv1[sgrp][ob] = *reinterpret_cast<const f32x4*>(a.fb_y + pixs[sgrp][ob] * a.fb_y_cs + a.fb_y_c0 + cb); }
          }
        }
        if (oa == 1) __builtin_amdgcn_s_barrier(); // round 0's buffer has been read
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 P[2];                              // this wave's contribution to y[oa][b], b = 0 / 1
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            float R[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              R[i][0] = acc[i * 4 + 0][r] + acc[i * 4 + 1][r] + acc[i * 4 + 2][r];
              R[i][1] = acc[i * 4 + 1][r] - acc[i * 4 + 2][r] - acc[i * 4 + 3][r];
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
              P[bb][e] = (oa == 0) ? (xp ? R[0][bb] : R[0][bb] + R[1][bb]) : (xp ? -R[0][bb] - R[1][bb] : R[1][bb]);
          }
          char* const wbq = xbuf + ((xp * 2) * 4 + q) * 1024 + ((q & 1) ? wpos1 : wpos);
          *reinterpret_cast<f32x4*>(wbq) = P[0];
          *reinterpret_cast<f32x4*>(wbq + 4096) = P[1];
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int sgrp = 0; sgrp < 2; ++sgrp)
#pragma unroll
          for (int ob = 0; ob < 2; ++ob) {
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(xbuf + ((0 * 2 + ob) * 4) * 1024 + rpos[sgrp]);
            const f32x4 y1 = *reinterpret_cast<const f32x4*>(xbuf + ((1 * 2 + ob) * 4) * 1024 + rpos[sgrp]);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float yv = y0[e] + y1[e];
              chk = fmaf(yv, 0.f, chk);
              float z = fmaf(yv, ms[e], bs[e]);
              if (RES == 3) z = fmaf(rv1[sgrp][ob][e], ms[e] * 2048.f, z);      // stored partial sum, scaled like the conv sum
              v[e] = fmaxf(z, slope_l * z);         // none / relu / leaky relu for slope 1 / 0 / 0.2 (a non-finite z trips chk)
              if (RES == 1 || RES == 2) v[e] = fmaf(v[e], a.rs1, rv1[sgrp][ob][e]);
              if (RES == 2) v[e] = fmaf(v[e], a.rs2, rv2[sgrp][ob][e]);
            }
            if (FBK) {
              if (fb && oks[sgrp][ob] && cb < a.cout) {          // dL/dy -> dL/dpre of the conv that produced y (as conv_epilogue_bwd_kernel)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float yv = rv1[sgrp][ob][e];
                  const bool pos = a.fb_act == 1 ? (yv > 0.f) : (yv >= 0.f);
                  const float dz = (a.fb_act == 0 || pos) ? v[e] : v[e] * fb_neg;
                  v[e] = dz;
                  fbs[e] += dz;
                  fbm = fmaxf(fbm, fabsf(dz));
                }
              }
            }
            if (oks[sgrp][ob] && cb < a.cout) {
              if (second) *reinterpret_cast<f32x4*>(a.out2 + pixs[sgrp][ob] * a.out2_cs + a.out2_c0 + (cb - a.out2_split)) = v;
              else *reinterpret_cast<f32x4*>(a.out + pixs[sgrp][ob] * a.out_cs + a.out_c0 + cb) = v;
            }
          }
      }
    }
    if (__any(chk != chk)) {
      if (lane == 0) atomicOr(a.ovf, 1 | (2 << (eb % 30)));     // bit 0 + the unit's sample slot (see hcf_conv_f16x3.hip)
    }
    if (FBK) {
      if (fb) {                                    // own slot: no other thread touches it (a fixed summation order per block)
        f32x4* const slot = reinterpret_cast<f32x4*>(lds + FB2_OFF + tid * 16);
        *slot = *slot + fbs;
        if (fbm > 0.f) atomicMax(reinterpret_cast<int*>(lds + FB2_OFF + 512 * 16), __builtin_bit_cast(int, fbm));
      }
    }
#if defined(WINO_PROF)
    pw[3] += __builtin_readcyclecounter() - qe0;
#endif
    u = un;
    if (u >= nunits) break;
  }
  if (FBK) {
    if (fb) {
      // the lane's channel quad is (lane & 7) in every unit: channel c = 4 (lane & 7) + e sums the 64 threads (8 waves x 8 patch
      // groups) that hold it, in a fixed order; one row of partials per block (reduced later by launch_sum_jobs)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 32) {
        const int q4 = tid >> 2, e = tid & 3;
        float t0 = 0.f;
        for (int j = 0; j < 64; ++j) t0 += reinterpret_cast<const float*>(lds + FB2_OFF)[(j * 8 + q4) * 4 + e];
        if (tid < a.cout) {
          a.fb_part[((size_t)blockIdx.x * 2 + 0) * a.cout + tid] = t0;
          a.fb_part[((size_t)blockIdx.x * 2 + 1) * a.cout + tid] = 0.f;
        }
      }
      if (tid == 0) {
        const int m = *reinterpret_cast<const int*>(lds + FB2_OFF + 512 * 16), m2 = max(m, in_max_bits);
        if (a.fb_max && m) atomicMax(reinterpret_cast<int*>(a.fb_max), m);
        if (a.fb_max2 && m2) atomicMax(reinterpret_cast<int*>(a.fb_max2), m2);
      }
    }
  }
#if defined(WINO_PROF)
  if (a.dbg && lane == 0 && (blockIdx.x & 31) == 17) {
    atomicAdd(a.dbg + 0, pw[0]); atomicAdd(a.dbg + 1, pw[1]); atomicAdd(a.dbg + 2, __builtin_readcyclecounter() - pw_t0);
    atomicAdd(a.dbg + 3, pw[3]); atomicAdd(a.dbg + 4, 1ull); atomicAdd(a.dbg + 5, pw[5]); atomicAdd(a.dbg + 6, pw[6]);
  }
#endif
#undef W2_SETUP_UNIT
#undef W2_DMA
#undef W2_A_SLOT
#undef W2_ISSUE
#undef W2_CHUNK_SCALARS
}

// ---------------------------------------------------------------------------------------------------------------------
// v4: the 64-output-channel form (RDB conv5). What bounds v3 is VALU issue (28 VALU per 3 MFMAs: every V value feeds one
// 32-channel tile); here a wave owns ONE transform row of a tile group and BOTH channel tiles, so each V feeds 6 MFMAs
// (152 VALU per 24 MFMAs per wave and chunk instead of 270). Unit = 8 x 32 pixels x 64 channels, 8 waves = 4 transform
// rows x 2 tile groups, 8 accumulators (4 positions x 2 channel tiles) per wave.
// LDS: the weights of a chunk are 64 KB (16 positions x 2 planes x 2 tiles x 1 KB), so W is double-buffered (128 KB) and the
// 22 KB halo image is SINGLE-buffered: every wave reads its two patch rows right after the chunk barrier, a second barrier
// ("rows are in registers") releases the image, and the next chunk's image DMA then has almost a whole chunk to land.
namespace v4 {
constexpr int TH4 = 8, HH4 = TH4 + 2;
constexpr int A4_BYTES = 22 * 1024;               // 1 360 real pieces (10 x 34 pixels x 4 parts) + 48 dead; TWO image buffers
constexpr int W4L_BYTES = 48 * 1024;              // a chunk's weights in LDS: [xi 4][nu 4][(tile 0, hi), (tile 0, lo), (tile 1, hi)] x 1 KB;
                                                  // the lo plane of tile 1 (a quarter of the pack) goes L2 -> registers, see the kernel
constexpr int W4_OFF0 = 2 * A4_BYTES;             // 45 056
constexpr int X4_SPARE = 19 * 1024;               // between the weight buffers: either one + the spare >= 64 KB for the epilogue's exchange
constexpr int W4_OFF1 = W4_OFF0 + W4L_BYTES + X4_SPARE;
constexpr int TAB4_OFF = W4_OFF1 + W4L_BYTES;     // 162 816
constexpr int LDS4_BYTES = TAB4_OFF + 1024;       // 163 840 = all of the CU's LDS (tables: bias, scale of the conv and of the fused 1x1 layer)
constexpr int NPIECE4 = 22 + 48;                  // DMA instructions per chunk
}  // namespace v4

// Output transform + epilogue of the 64-output-channel kernels (conv_wino4_kernel, conv_wino5_kernel): acc[nu][tile] holds this
// wave's transform row xi of its tile group tg; `xbase` = 64 KB of LDS nobody else uses until the block's next barrier after the
// call (the weight buffer just consumed), `tab_off` = the bias / scale tables. Starts with a block barrier.
template <int RES>
__device__ __forceinline__ void wino64_epilogue(const Args& a, char* const lds, const int tab_off, char* const xbase, f32x16 (&acc)[4][2],
                                                const int eb, const int ey0, const int ex0, const int H, const int W, const float slope,
                                                const float slope2, const float slope_f, const int wave, const int lane) {
  constexpr bool F1 = (RES == 3);
  const int half = lane >> 5, li = lane & 31, xi = wave & 3, tg = wave >> 2;
  (void)li;
  // ---- epilogue: R[b] = sum_nu M[nu] A[nu][b] per wave (its transform row). The four rows of a tile group meet in LDS
  // (the weight buffer just consumed), one channel tile per round, and the exchange is also a TRANSPOSE: after it lane
  // (patch p of 8, register group q, half) of wave xi holds, for patch 8 xi + p, the 4 channels 8 q + 4 half .. + 3 of all four
  // rows -> y0 = R0 + R1 + R2, y1 = R1 - R2 - R3, and 8 consecutive lanes cover 128 contiguous bytes of one pixel
  // (residual loads and stores touch 8 cache lines per instruction instead of 32).
  float chk = 0.f;
  char* const xbuf = xbase + tg * 32768;
  const int p8 = lane >> 3, qd = (lane >> 1) & 3, hd = lane & 1;              // reader role
  const int patch = 8 * xi + p8, prow = patch >> 4, pcol = patch & 15;
  const int wpos = (half * 32 + ((li + 4 * half) & 31)) * 16;                 // writer slot for even q; odd q: + 8 patches (mod 32)
  const int wpos1 = (half * 32 + ((li + 4 * half + 8) & 31)) * 16;
  const int rpos = (hd * 32 + ((patch + 4 * hd + 8 * (qd & 1)) & 31)) * 16 + qd * 1024;
  f32x4 hv[2][2][2];                              // F1: the first layer's values of this lane (2 rounds x 2 x 2 pixels x 4 channels)
  __builtin_amdgcn_s_barrier();                   // every wave is done reading the last chunk's weights
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int cb = nt * 32 + 8 * qd + 4 * hd;
    f32x4 rv1[2][2], rv2[2][2];                   // this round's residuals: issued now, used after the exchange
    size_t pixs[2][2];
    bool oks[2][2];
#pragma unroll
    for (int oa = 0; oa < 2; ++oa)
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        const int yy = ey0 + 4 * tg + 2 * prow + oa, xx = ex0 + 2 * pcol + ob;
        oks[oa][ob] = yy < H && xx < W;
        pixs[oa][ob] = (size_t)((size_t)eb * H + (yy < H ? yy : H - 1)) * W + (xx < W ? xx : W - 1);
        if (RES == 1 || RES == 2) rv1[oa][ob] = *reinterpret_cast<const f32x4*>(a.res1 + pixs[oa][ob] * a.res1_cs + a.res1_c0 + cb);
        if (RES == 2) rv2[oa][ob] = *reinterpret_cast<const f32x4*>(a.res2 + pixs[oa][ob] * a.res2_cs + a.res2_c0 + cb);
      }
    if (nt == 1) __builtin_amdgcn_s_barrier();    // round 0's buffer has been read
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 Rq[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e;
        Rq[0][e] = acc[0][nt][r] + acc[1][nt][r] + acc[2][nt][r];
        Rq[1][e] = acc[1][nt][r] - acc[2][nt][r] - acc[3][nt][r];
      }
      char* const wbq = xbuf + ((xi * 2) * 4 + q) * 1024 + ((q & 1) ? wpos1 : wpos);
      *reinterpret_cast<f32x4*>(wbq) = Rq[0];
      *reinterpret_cast<f32x4*>(wbq + 4096) = Rq[1];
    }
    __builtin_amdgcn_s_barrier();
    f32x4 R[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) R[x][bb] = *reinterpret_cast<const f32x4*>(xbuf + ((x * 2 + bb) * 4) * 1024 + rpos);
    const f32x4 bs = *reinterpret_cast<const f32x4*>(lds + tab_off + cb * 4);
    const f32x4 ms = *reinterpret_cast<const f32x4*>(lds + tab_off + 256 + cb * 4);
    const bool split_t = (nt == 1) && a.out2 != nullptr;         // second tile routed to its own tensor / activation
    const float slope_t = split_t ? slope2 : slope;
#pragma unroll
    for (int oa = 0; oa < 2; ++oa)
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float yv = oa ? (R[1][ob][e] - R[2][ob][e]) - R[3][ob][e] : (R[0][ob][e] + R[1][ob][e]) + R[2][ob][e];
          chk = fmaf(yv, 0.f, chk);
          const float z = fmaf(yv, ms[e], bs[e]);
          v[e] = fmaxf(z, slope_t * z);
          if (RES == 1 || RES == 2) v[e] = fmaf(v[e], a.rs1, rv1[oa][ob][e]);
          if (RES == 2) v[e] = fmaf(v[e], a.rs2, rv2[oa][ob][e]);
        }
        if (F1) hv[nt][oa][ob] = v;
        else {
          if (nt == 0 && oa == 0 && ob == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every load of this wave (the next
                                                    // unit's first chunk included) has landed BEFORE the unit's first store is issued:
                                                    // the next chunk's top does not wait again (conv_wino4_kernel)
        }
        if (!F1 && oks[oa][ob] && cb < a.cout) {
          if (split_t) *reinterpret_cast<f32x4*>(a.out2 + pixs[oa][ob] * a.out2_cs + a.out2_c0 + (cb - 32)) = v;
          else *reinterpret_cast<f32x4*>(a.out + pixs[oa][ob] * a.out_cs + a.out_c0 + cb) = v;
        }
      }
  }
  if (F1) {
    // ---- second layer: h2 = act_f((W_f h1 + bias_f) * scale_f) on the tile's 256 pixels. The exchange buffer becomes 256 pixel
    // records of 256 bytes: 16 slots of 8 halves, slot (plane * 8 + channel / 8) ^ (pixel & 15) (a 128-bit LDS read is served in
    // groups of 16 lanes, here 16 consecutive pixels -> 16 distinct slots). A wave multiplies 32 pixels (the N of
    // the matrix instruction) of two tile rows by one 32-channel half of W_f; its A fragments come from L2 (8 KB, lane order).
    const int mt = wave & 1, r0 = 2 * (wave >> 1);    // this wave: output channels 32 mt .. + 31 of tile rows r0, r0 + 1
    f16x8 wf[4][2];
    asm volatile("" ::: "memory");                // (keeps the fragments' registers out of the exchange rounds' live range)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        wf[ks][pl] = *reinterpret_cast<const f16x8*>(a.f_w + (size_t)(((mt * 4 + ks) * 2 + pl) * 64 + lane) * 16);
    char* const rec = xbase;
    __builtin_amdgcn_s_barrier();                 // round 1's exchange has been read by every wave
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int oa = 0; oa < 2; ++oa)
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
          const int pi_ = (4 * tg + 2 * prow + oa) * 32 + 2 * pcol + ob;
          u32x2 h_, l_;
          split4(hv[nt][oa][ob], h_, l_);
          char* const rp = rec + pi_ * 256 + hd * 8;
          *reinterpret_cast<u32x2*>(rp + (((nt * 4 + qd) ^ (pi_ & 15)) << 4)) = h_;
          *reinterpret_cast<u32x2*>(rp + (((8 + nt * 4 + qd) ^ (pi_ & 15)) << 4)) = l_;
        }
    __builtin_amdgcn_s_barrier();
    f32x16 c2[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int r = 0; r < 16; ++r) c2[rr][r] = 0.f;
    const char* const rq = rec + (r0 * 32 + li) * 256;
    const int key = li & 15;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const f16x8 vh_ = *reinterpret_cast<const f16x8*>(rq + rr * 8192 + (((2 * ks + half) ^ key) << 4));
        const f16x8 vl_ = *reinterpret_cast<const f16x8*>(rq + rr * 8192 + (((8 + 2 * ks + half) ^ key) << 4));
        c2[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][0], vh_, c2[rr], 0, 0, 0);
        c2[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][1], vh_, c2[rr], 0, 0, 0);
        c2[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][0], vl_, c2[rr], 0, 0, 0);
      }
    }
    const int xx = ex0 + li;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (as above: before the unit's first store)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int yy = ey0 + r0 + rr;
      const bool ok2 = yy < H && xx < W;
      float* const op = a.out + ((size_t)((size_t)eb * H + (yy < H ? yy : H - 1)) * W + (size_t)(xx < W ? xx : W - 1)) * a.out_cs + a.out_c0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cb2 = mt * 32 + 8 * q + 4 * half;
        const f32x4 bs2 = *reinterpret_cast<const f32x4*>(lds + tab_off + 512 + cb2 * 4);
        const f32x4 ms2 = *reinterpret_cast<const f32x4*>(lds + tab_off + 768 + cb2 * 4);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float yv = c2[rr][4 * q + e];
          chk = fmaf(yv, 0.f, chk);
          const float z = fmaf(yv, ms2[e], bs2[e]);
          v[e] = fmaxf(z, slope_f * z);
        }
        if (ok2) *reinterpret_cast<f32x4*>(op + cb2) = v;
      }
    }
  }
  if (__any(chk != chk)) {
    if (lane == 0) atomicOr(a.ovf, 1 | (2 << (eb % 30)));       // bit 0 + the unit's sample slot (see hcf_conv_f16x3.hip)
  }
}

template <int RES, bool SC = false>
__global__ __launch_bounds__(512, 1) void conv_wino4_kernel(const Args a, const int nunits) {
  using namespace v4;
  using v2::ROWB;
  using v2::a2_off;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave & 3, tg = wave >> 2;       // transform row xi of tile group tg (output rows 4 tg .. 4 tg + 3)
  const int H = a.H, W = a.W, nchunk = a.nchunk;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH4 - 1) / TH4;

  // DMA instruction I = 8 j + wave, j = 0..8: I < 22 halo pieces, 22 <= I < 70 weight piece (LDS order) I - 22, I >= 70 nothing.
  // j = 0, 1: halo; j = 2: halo for waves 0..5, weight pieces 0 / 1 for waves 6 / 7; j = 3..7: weights; j = 8: waves 0..5.
  const bool whi = (wave >= 6);
  const int padpix = a.B * H * W;                // out-of-range pixel index: the DMA writes zeros (conv padding / dead pieces)
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

  int ub = 0, uy0 = 0, ux0 = 0, uc = 0;          // DMA cursor
#define W4_SETUP_UNIT(U)                                                                           \
  {                                                                                                \
    const int v_ = a.rev ? nunits - 1 - xcd_remap((U), nunits) : xcd_remap((U), nunits);           \
    ux0 = __builtin_amdgcn_readfirstlane((v_ % tiles_x) * TW);                                     \
    uy0 = __builtin_amdgcn_readfirstlane(((v_ / tiles_x) % tiles_y) * TH4);                        \
    ub = __builtin_amdgcn_readfirstlane(v_ / (tiles_x * tiles_y));                                 \
    uc = 0;                                                                                        \
    int ln_ = lane;                                                                                \
    asm volatile("" : "+v"(ln_));                                                                  \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                \
      const int pa_ = (8 * j + wave) * 64 + ln_;                                                   \
      const int hy_ = (pa_ * 241) >> 15;                                                           \
      const int q_ = pa_ - hy_ * 136, m_ = q_ >> 4;                                                \
      const int hx_ = m_ * 4 + (((q_ & 15) ^ (m_ & 7)) >> 2);                                      \
      const int y = uy0 + hy_ - 1, x = ux0 + hx_ - 1;                                              \
      upix[j] = (hy_ < HH4 && y >= 0 && y < H && x >= 0 && x < W) ? (ub * H + y) * W + x : padpix; \
    }                                                                                              \
  }
#define W4_DMA(RS, VOFF, SOFF, DST) __builtin_amdgcn_raw_ptr_buffer_load_lds((RS), (lptr)(DST), 16, (VOFF), (SOFF), 0, 0)
  int csb_ = 0, so_ = 0, ws_ = 0;
  uint32_t pp_ = partpk;
  __amdgpu_buffer_rsrc_t rsa_ = rs0;
#define W4_CHUNK_SCALARS()                                                                         \
  {                                                                                                \
    const int sidx_ = (uc < k0) ? 0 : (uc < k1) ? 1 : 2;                                           \
    csb_ = sidx_ == 0 ? csb0 : sidx_ == 1 ? csb1 : csb2;                                           \
    rsa_ = sidx_ == 0 ? rs0 : sidx_ == 1 ? rs1 : rs2;                                              \
    so_ = (sidx_ == 0 ? cb0 + uc * 64 : sidx_ == 1 ? cb1 + (uc - k0) * 64 : cb2 + (uc - k1) * 64); \
    ws_ = uc * W4_BYTES;                                                                           \
    pp_ = partpk;                                                                                  \
    asm volatile("" : "+v"(pp_));                                                                  \
  }
#define W4_A_SLOT(J, IB)                                                                           \
  {                                                                                                \
    const int p16_ = (int)((pp_ >> (2 * (J))) & 0x30u);                                            \
    const int vo_ = (int)__umul24((unsigned)upix[J], (unsigned)csb_) + p16_;                       \
    W4_DMA(rsa_, vo_, so_, lds + (IB) * A4_BYTES + (8 * (J) + wave) * 1024);                       \
  }
  // halo pieces of the cursor's chunk into image buffer IB (the one whose rows were read before the last barrier)
#define W4_ISSUE_A(IB)                                                                             \
  {                                                                                                \
    W4_A_SLOT(0, IB) W4_A_SLOT(1, IB)                                                              \
    if (!whi) W4_A_SLOT(2, IB)                                                                     \
  }
  // weight pieces [J0, J1) (j index) of the cursor's chunk into weight buffer WB: LDS piece l = xi' * 12 + nu * 3 + s comes
  // from pack piece xi' * 16 + nu * 4 + s (s = 0, 1, 2: tile 0 hi, tile 0 lo, tile 1 hi)
#define W4_ISSUE_W(J0, J1, WB)                                                                     \
  {                                                                                                \
    char* const wb_ = lds + ((WB) ? W4_OFF1 : W4_OFF0);                                            \
    _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) {                                       \
      const int l_ = 8 * j_ + wave - 22;                                                           \
      const int x_ = l_ / 12, r_ = l_ - 12 * x_, n_ = r_ / 3;                                      \
      const int g_ = x_ * 16 + n_ * 4 + (r_ - 3 * n_);                                             \
      if (j_ == 2) { if (whi) W4_DMA(rsw, wvo, ws_ + g_ * 1024, wb_ + l_ * 1024); }                \
      else if (j_ < 8) W4_DMA(rsw, wvo, ws_ + g_ * 1024, wb_ + l_ * 1024);                         \
      else if (!whi) W4_DMA(rsw, wvo, ws_ + g_ * 1024, wb_ + l_ * 1024);                           \
    }                                                                                              \
  }
  // the lo plane of channel tile 1 does not go through LDS: this wave's fragment of position NU (pack piece xi * 16 + NU * 4 + 3)
  // of the cursor's chunk comes straight from L2 into the register the position loop just used for the current chunk's
  // (inline asm: the compiler's own vmcnt bookkeeping would wait for ALL DMA at the first use; the chunk's top wait covers it)
  const gcptr w11p = uniform_ptr(a.wpack + (size_t)(xi * 16 + 3) * 1024);
#define W4_LOAD_W11(NU) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(w11[NU]) : "v"(wvo), "s"(w11p + ws_ + (NU) * 4096) : "memory");

  // patch reads: transform row xi uses patch rows (rA, rB) = (0,2) (1,2) (1,2) (1,3): t = rA + sg rB, sg = -1, +1, -1, -1
  const int trow = li >> 4, tcol = li & 15;
  const int rA = (xi == 0) ? 0 : 1, rB = (xi == 3) ? 3 : 2;
  const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (xi == 1) ? 1.f : -1.f)));
  int poff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) poff[j] = a2_off(4 * tg + 2 * trow, 2 * tcol + j, 2 * half);
  const int offA = rA * ROWB, offB = rB * ROWB;
  const int fw = lane * 16 + xi * (12 * 1024);   // + (nu * 3 + s) * 1024

  constexpr bool F1 = (RES == 3);                // fused 1x1 second layer (Args::f_w)
  float in_s = 1.f, out_s = 1.f;
  int in_max_bits = 0;
  if (SC) input_scales(a.in_max, in_s, out_s, in_max_bits);
  (void)in_max_bits;
  if (tid < 64) {
    reinterpret_cast<float*>(lds + TAB4_OFF)[tid] = a.bias[tid] * a.scale[tid];
    reinterpret_cast<float*>(lds + TAB4_OFF)[64 + tid] = a.scale[tid] * UNSPLIT * out_s;
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
  if (a.clk != nullptr && blockIdx.x == 0 && tid == 0) {      // (start stamps parked in memory: nothing of the probe lives through the loop)
    a.clk[2] = __builtin_readcyclecounter();
    a.clk[3] = __builtin_amdgcn_s_memrealtime();
  }
  f16x8 w11[4];
  W4_SETUP_UNIT(u)
  W4_CHUNK_SCALARS()
  W4_ISSUE_A(0)
  W4_ISSUE_W(2, 9, 0)
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) W4_LOAD_W11(nu)
  ++uc;
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
    const int eb = ub, ey0 = uy0, ex0 = ux0;
    const int un = u + gridDim.x;

    for (int c = 0; c < nchunk; ++c, ++g) {
      const int stg = g & 1;
#if defined(WINO_PROF)
      const unsigned long long q0 = __builtin_readcyclecounter();
#endif
      // (the first chunk of a LATER unit: its image, weights and fragments were requested during the previous unit's last chunk and
      //  waited for inside that unit's epilogue, before its first global store -- vmcnt counts stores too, and a wait here would
      //  sit out the epilogue's store acknowledgements, ~1 500 cycles per unit)
      if (c > 0 || g == 0 || a.top_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(WINO_PROF)
      const unsigned long long q1 = __builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_barrier();                 // this chunk's image, weights and fragments are complete and visible; every
                                                    // wave is through the previous chunk: its image and weight buffers are free
#if defined(WINO_PROF)
      const unsigned long long q2 = __builtin_readcyclecounter();
      pw[0] += q1 - q0; pw[1] += q2 - q1;
#endif
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) asm volatile("" : "+v"(w11[nu]));      // (the fragments were written behind the compiler's back)
      if (c + 1 == nchunk) {
        if (un < nunits) W4_SETUP_UNIT(un)
        else {
          uc = 0;
#pragma unroll
          for (int j = 0; j < 3; ++j) upix[j] = padpix;
        }
      }
      W4_CHUNK_SCALARS()
      W4_ISSUE_A((g + 1) & 1)                       // the next chunk's image: a whole chunk ahead of its first read
      float t_[4][8];
      {
        const char* const ib = lds + (g & 1) * A4_BYTES;
        float ra[4][8], rb[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 x0_ = *reinterpret_cast<const f32x4*>(ib + poff[j] + offA);
          const f32x4 x1_ = *reinterpret_cast<const f32x4*>(ib + (poff[j] ^ 16) + offA);
          const f32x4 y0_ = *reinterpret_cast<const f32x4*>(ib + poff[j] + offB);
          const f32x4 y1_ = *reinterpret_cast<const f32x4*>(ib + (poff[j] ^ 16) + offB);
#pragma unroll
          for (int k = 0; k < 4; ++k) { ra[j][k] = x0_[k]; ra[j][4 + k] = x1_[k]; rb[j][k] = y0_[k]; rb[j][4 + k] = y1_[k]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 8; ++k) t_[j][k] = fmaf(sg, rb[j][k], ra[j][k]);
      }
#if defined(WINO_PROF)
      const unsigned long long q3 = __builtin_readcyclecounter();
      pw[5] += q3 - q2;
#endif
      const char* const wb = lds + (stg ? W4_OFF1 : W4_OFF0) + fw;
#define W4_V(NU, K) (((NU) == 0) ? t_[0][K] - t_[2][K] : ((NU) == 1) ? t_[1][K] + t_[2][K] : ((NU) == 2) ? t_[1][K] - t_[2][K] : t_[1][K] - t_[3][K])
#define W4_MFMA(P, N, WW, VX) acc[P][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WW, __builtin_bit_cast(f16x8, VX), acc[P][N], 0, 0, 0);
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        float v_[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v_[k] = W4_V(nu, k);
        u32x4 vh_, vl_;
        if (SC) split8s(v_, in_s, vh_, vl_); else split8(v_, vh_, vl_);
        const f16x8 w00 = *reinterpret_cast<const f16x8*>(wb + (nu * 3 + 0) * 1024);
        const f16x8 w01 = *reinterpret_cast<const f16x8*>(wb + (nu * 3 + 1) * 1024);
        const f16x8 w10 = *reinterpret_cast<const f16x8*>(wb + (nu * 3 + 2) * 1024);
        W4_MFMA(nu, 0, w00, vh_)
        W4_MFMA(nu, 1, w10, vh_)
        W4_MFMA(nu, 0, w01, vh_)
        W4_MFMA(nu, 1, w11[nu], vh_)
        W4_MFMA(nu, 0, w00, vl_)
        W4_MFMA(nu, 1, w10, vl_)
        if (nu == 0) { W4_ISSUE_W(2, 5, stg ^ 1) }
        else if (nu == 1) { W4_ISSUE_W(5, 7, stg ^ 1) }
        else if (nu == 2) { W4_ISSUE_W(7, 9, stg ^ 1) }
        W4_LOAD_W11(nu)                           // (this position's fragment of the cursor's chunk, into the register just read)
      }
      ++uc;
#if defined(WINO_PROF)
      pw[6] += __builtin_readcyclecounter() - q3;
#endif
#undef W4_V
#undef W4_MFMA
    }

#if defined(WINO_PROF)
    const unsigned long long qe0 = __builtin_readcyclecounter();
#endif
    int lane_e = lane;                            // (opaque: the epilogue's lane-constant addresses are recomputed per unit, not kept
    asm volatile("" : "+v"(lane_e));              //  in registers through the chunk loop, where there are none to spare)
    wino64_epilogue<RES>(a, lds, TAB4_OFF, lds + (((g - 1) & 1) ? W4_OFF0 + W4L_BYTES : W4_OFF0), acc, eb, ey0, ex0, H, W, slope, slope2, slope_f, wave, lane_e);
#if defined(WINO_PROF)
    pw[3] += __builtin_readcyclecounter() - qe0;
#endif
    u = un;
    if (u >= nunits) break;
  }
  if (a.clk != nullptr && blockIdx.x == 0 && tid == 0) {
    a.clk[0] += __builtin_readcyclecounter() - a.clk[2];       // (launches of one stream are serial: no atomics; the probe is for
    a.clk[1] += __builtin_amdgcn_s_memrealtime() - a.clk[3];   //  single-stream regions, bench.py's roofline leg)
  }
#if defined(WINO_PROF)
  if (a.dbg && lane == 0 && (blockIdx.x & 31) == 17) {
    atomicAdd(a.dbg + 0, pw[0]); atomicAdd(a.dbg + 1, pw[1]); atomicAdd(a.dbg + 2, __builtin_readcyclecounter() - pw_t0);
    atomicAdd(a.dbg + 3, pw[3]); atomicAdd(a.dbg + 4, 1ull); atomicAdd(a.dbg + 5, pw[5]); atomicAdd(a.dbg + 6, pw[6]);
    atomicAdd(a.dbg + 7, pw[7]);
  }
#endif
#undef W4_SETUP_UNIT
#undef W4_DMA
#undef W4_A_SLOT
#undef W4_ISSUE_A
#undef W4_ISSUE_W
#undef W4_LOAD_W11
#undef W4_CHUNK_SCALARS
}


static inline int launch(const Args& a, int ncu, hipStream_t st, int version = 2) {
  if (a.nsrc < 1 || a.nsrc > 3 || !a.wpack || !a.bias || !a.scale || !a.ovf || !a.zeros || !a.out || a.nchunk < 1) return -1;
  if (a.ntile_n < 1 || a.ntile_n > 2 || (a.cout & 3)) return -6;
  int kt = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    if (!a.src[i].p || (a.src[i].n & 15) || (a.src[i].cs & 3) || (a.src[i].c0 & 3) || (reinterpret_cast<uintptr_t>(a.src[i].p) & 15)) return -6;
    if ((long long)a.B * a.H * a.W * a.src[i].cs * 4 >= 0x7fffe000LL) return -6;       // 31-bit byte offsets (out-of-range = padding)
    kt += a.src[i].n >> 4;
  }
  if (kt != a.nchunk) return -1;
  if (((a.out_cs | a.out_c0) & 3) || (reinterpret_cast<uintptr_t>(a.out) & 15)) return -6;
  if (a.res1 && (((a.res1_cs | a.res1_c0) & 3) || (reinterpret_cast<uintptr_t>(a.res1) & 15))) return -6;
  if (a.res2 && (((a.res2_cs | a.res2_c0) & 3) || (reinterpret_cast<uintptr_t>(a.res2) & 15))) return -6;
  if (a.res2 && !a.res1) return -1;
  if ((long long)a.B * a.H * a.W >= (1LL << 24)) return -6;                            // 24-bit pixel index (mul24)
  if (version == 4 && (a.ntile_n != 2 || a.cout > 64)) return -6;
  const int th = (version == 2) ? v2::TH2 : (version == 4) ? v4::TH4 : TH;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + th - 1) / th;
  const long long nunits = (long long)a.B * tiles_x * tiles_y * (version == 4 ? 1 : a.ntile_n);
  if (nunits < 1 || nunits > 0x7fffffffLL) return -1;
  const unsigned grid = (unsigned)(nunits < ncu ? nunits : ncu);
  // (the opt-in to > 64 KB of dynamic LDS is per device: several GPUs in one process, e.g. nn.DataParallel replicas)
  static bool attr_dev[64][3][8] = {};
  int dev_ = 0;
  if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return -2;
  bool (&attr)[3][8] = attr_dev[dev_];
  if (a.pre && (version != 2 || a.res1 || a.res2 || ((a.pre_cs | a.pre_c0) & 3) || (reinterpret_cast<uintptr_t>(a.pre) & 15))) return -6;
  if (a.out2 && (a.res1 || a.pre || ((a.out2_cs | a.out2_c0) & 3) || (reinterpret_cast<uintptr_t>(a.out2) & 15))) return -6;
  if (a.out2 && version != 4 && (version != 2 || a.ntile_n != 1 || a.out2_split != 16)) return -6;
  if (a.f_w && (version != 4 || a.res1 || a.res2 || a.out2 || !a.f_bias || !a.f_scale || (reinterpret_cast<uintptr_t>(a.f_w) & 15))) return -6;
  const int res = (a.pre || a.f_w) ? 3 : a.res2 ? 2 : a.res1 ? 1 : 0;
  // data-gradient form: plain store (+ the fused epilogue backward, 32-channel kernel only) or accumulation into `out` (res1 = out)
  const bool sc = a.in_max != nullptr;
  if (sc && (res > 1 || a.out2 || (version != 2 && version != 4))) return -6;
  if (a.fb_y && (!sc || version != 2 || res != 0 || a.ntile_n != 1 || !a.fb_part || ((a.fb_y_cs | a.fb_y_c0) & 3) ||
                 (reinterpret_cast<uintptr_t>(a.fb_y) & 15))) return -6;
  const int ldsb = (version == 2) ? (sc ? v2::LDS2S_BYTES : v2::LDS2_BYTES) : (version == 4) ? v4::LDS4_BYTES : LDS_BYTES;
  const int vi = (version == 2) ? 1 : (version == 4) ? 2 : 0;
  const int ai = res + (sc ? 4 : 0);
  auto go = [&](auto fn, int threads) {
    if (!attr[vi][ai]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) != hipSuccess) return -2;
      attr[vi][ai] = true;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), ldsb, st, a, (int)nunits);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  };
  if (sc && version == 4) return res == 0 ? go(conv_wino4_kernel<0, true>, 512) : go(conv_wino4_kernel<1, true>, 512);
  if (sc && version == 2) return res == 0 ? go(conv_wino2_kernel<0, true>, 512) : go(conv_wino2_kernel<1, true>, 512);
  if (version == 4) {
    if (res == 0) return go(conv_wino4_kernel<0>, 512);
    if (res == 1) return go(conv_wino4_kernel<1>, 512);
    if (res == 3) return go(conv_wino4_kernel<3>, 512);
    return go(conv_wino4_kernel<2>, 512);
  }
  if (version == 2) {
    if (res == 0) return go(conv_wino2_kernel<0>, 512);
    if (res == 1) return go(conv_wino2_kernel<1>, 512);
    if (res == 3) return go(conv_wino2_kernel<3>, 512);
    return go(conv_wino2_kernel<2>, 512);
  }
  return -6;
}
#endif  // __HIPCC__

}  // namespace wino
}  // namespace hcf
