// EXPERIMENT (round 1): wave-specialised variant of the f16x3 3x3 convolution (plain epilogue only).
//
// hcf_conv_f16x3.hip gives every wave both jobs (stage the next chunk, run the MFMAs) and separates them with
// two barriers per chunk; PMC counters put the matrix pipe at ~59 % busy. Here a block has 8 waves:
//   waves 0-3  consumers: ds_read fragments + MFMAs only, on LDS buffer (c & 1);
//   waves 4-7  producers: global loads -> hi/lo split -> ds_write of chunk c+1 into buffer ((c+1) & 1), then the
//              global loads of chunk c+2 (they stay in flight across the barrier);
// one barrier per chunk, LDS double-buffered (2 x (A + B) = 91 / 128 KB -> one block = 8 waves per CU, i.e. one
// consumer and one producer wave per SIMD). Selected by tools/conv_bench.py --ablate 16; measurements in
// profiles/r01_f16x3_notes.md. Same numerics, layouts and weight packs as the shipped kernel.
#include "hcf_common.h"

namespace hcf {
namespace f16x3ws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1)))* gfptr;
typedef const f32x4 __attribute__((address_space(1)))* gf4ptr;

constexpr int KC = 16, TW = 32, TH = 8, REC = 80;
constexpr float SPLIT = 2048.f;

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}
__device__ __forceinline__ gfptr uniform_ptr(const float* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gfptr)(((uint64_t)hi << 32) | lo);
}

template <int NTB>
__global__ __launch_bounds__(512, 1) void conv_f16x3_ws_kernel(const ConvArgs a) {
  constexpr int TAPS = 9, PAD = 1, NP = 256;                  // NP = producer threads
  constexpr int HH = TH + 2 * PAD, HW = TW + 2 * PAD, HP = HH * HW;
  constexpr int NLOAD = HP * (KC / 4);
  constexpr int NSLOT = (NLOAD + NP - 1) / NP;
  constexpr int NPAD = NTB * 32;
  constexpr int MT = 2 * NTB;
  constexpr int A_BYTES = HP * REC;
  constexpr int BHALF = NPAD * 16;
  constexpr int B_BYTES = TAPS * 2 * 2 * BHALF;
  constexpr int BV = B_BYTES / 16;
  constexpr int BSLOT = (BV + NP - 1) / NP;
  constexpr int BUF = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

  const int tid = threadIdx.x;
  const bool producer = tid >= 256;                            // wave-uniform
  const int ptid = tid & 255;
  const int lane = tid & 63, wave = (tid >> 6) & 3, half = lane >> 5, li = lane & 31;
  const int wm = (NTB == 2) ? (wave >> 1) : wave;
  const int wn = (NTB == 2) ? (wave & 1) : 0;
  const int H = a.H, W = a.W;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int txb = bid % tiles_x, tyb = (bid / tiles_x) % tiles_y, b = bid / (tiles_x * tiles_y);
  const int x0 = txb * TW, y0 = tyb * TH;
  const int nchunk = a.nchunk;

  if (producer) {
    // ---------------------------------------------------------------- producers: stage chunks into LDS
    int pix0[NSLOT];
    unsigned okmask = 0;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int q = ptid + NP * s;
      const int hp = min(q >> 2, HP - 1);
      const int hy = hp / HW, hx = hp - hy * HW;
      const int y = y0 + hy - PAD, x = x0 + hx - PAD;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      okmask |= ok ? (1u << s) : 0u;
      pix0[s] = (b * H + min(max(y, 0), H - 1)) * W + min(max(x, 0), W - 1);
    }
    const int uq = ptid & 3;
    const int u0 = (a.src[0].n + 3) >> 2;
    const int u1 = u0 + ((a.nsrc > 1) ? ((a.src[1].n + 3) >> 2) : 0);
    const int u2 = u1 + ((a.nsrc > 2) ? ((a.src[2].n + 3) >> 2) : 0);
    const gfptr sp0 = uniform_ptr(a.src[0].p + a.src[0].c0);
    const gfptr sp1 = uniform_ptr((a.nsrc > 1) ? a.src[1].p + a.src[1].c0 : a.src[0].p);
    const gfptr sp2 = uniform_ptr((a.nsrc > 2) ? a.src[2].p + a.src[2].c0 : a.src[0].p);
    const int cs0 = __builtin_amdgcn_readfirstlane(a.src[0].cs), cs1 = __builtin_amdgcn_readfirstlane(a.src[1].cs),
              cs2 = __builtin_amdgcn_readfirstlane(a.src[2].cs);
    const int n0 = __builtin_amdgcn_readfirstlane(a.src[0].n), n1 = __builtin_amdgcn_readfirstlane(a.src[1].n),
              n2 = __builtin_amdgcn_readfirstlane(a.src[2].n);
    const gf4ptr wq = (gf4ptr)uniform_ptr(a.wpack) + ptid;
    const gfptr zpage = uniform_ptr(a.zeros);
    int stg_valid = 0;
    f32x4 stg[NSLOT], stb[BSLOT];
#define WS_LOAD(CHUNK)                                                                            \
    {                                                                                             \
      const int u = (CHUNK) * 4 + uq;                                                             \
      const bool in0 = u < u0, in1 = u < u1, uok = u < u2;                                        \
      const int ul = in0 ? u : in1 ? (u - u0) : (u - u1);                                         \
      gfptr sp = in0 ? sp0 : in1 ? sp1 : sp2;                                                     \
      const int css = in0 ? cs0 : in1 ? cs1 : uok ? cs2 : cs0;                                    \
      const int nn = in0 ? n0 : in1 ? n1 : n2;                                                    \
      stg_valid = uok ? (nn - 4 * ul) : 0;                                                        \
      sp = uok ? sp + 4 * ul : sp0;                                                               \
      _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                         \
        gfptr p = sp + (unsigned)(pix0[s] * css);                                                 \
        p = ((okmask >> s) & 1u) ? p : zpage;                                                     \
        stg[s] = *(gf4ptr)(p);                                                                    \
      }                                                                                           \
      _Pragma("unroll") for (int s = 0; s < BSLOT; ++s) {                                         \
        const int q = ptid + NP * s;                                                              \
        stb[s] = wq[(size_t)(CHUNK) * BV + ((q < BV) ? NP * s : 0)];                              \
      }                                                                                           \
    }
#define WS_SPLIT_WRITE(BASE)                                                                      \
    {                                                                                             \
      _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                         \
        f32x4 v = stg[s];                                                                         \
        if (stg_valid < 4) {                                                                      \
          v.x = (stg_valid > 0) ? v.x : 0.f; v.y = (stg_valid > 1) ? v.y : 0.f;                   \
          v.z = (stg_valid > 2) ? v.z : 0.f; v.w = 0.f;                                           \
        }                                                                                         \
        f16x4 hh, ll;                                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
          const _Float16 h = (_Float16)v[e];                                                      \
          hh[e] = h;                                                                              \
          ll[e] = (_Float16)(v[e] - (float)h);                                                    \
        }                                                                                         \
        const int q = ptid + NP * s;                                                              \
        if (q < NLOAD) {                                                                          \
          char* rec = (BASE) + (q >> 2) * REC + (q & 3) * 8;                                      \
          *reinterpret_cast<f16x4*>(rec) = hh;                                                    \
          *reinterpret_cast<f16x4*>(rec + 32) = ll;                                               \
        }                                                                                         \
      }                                                                                           \
      _Pragma("unroll") for (int s = 0; s < BSLOT; ++s) {                                         \
        const int q = ptid + NP * s;                                                              \
        if (q < BV) *reinterpret_cast<f32x4*>((BASE) + A_BYTES + q * 16) = stb[s];                \
      }                                                                                           \
    }
    WS_LOAD(0)
    WS_SPLIT_WRITE(lds)
    if (nchunk > 1) WS_LOAD(1)
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
      if (c + 1 < nchunk) {
        WS_SPLIT_WRITE(lds + ((c + 1) & 1) * BUF)
        if (c + 2 < nchunk) WS_LOAD(c + 2)
      }
      __syncthreads();
    }
#undef WS_LOAD
#undef WS_SPLIT_WRITE
    return;
  }

  // ------------------------------------------------------------------ consumers: fragments + MFMAs + epilogue
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int abase = ((MT * wm) * HW + li) * REC + half * 16;
  const int bbase = half * BHALF + (wn * 32 + li) * 16;
  __syncthreads();                                              // chunk 0 is staged
  __builtin_amdgcn_s_setprio(1);
  for (int c = 0; c < nchunk; ++c) {
    const char* base = lds + (c & 1) * BUF;
    const char* ldsB = base + A_BYTES;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int dy = t / 3, dx = t % 3;
      const char* bt = ldsB + bbase + t * (4 * BHALF);
      const f16x8 b1 = *reinterpret_cast<const f16x8*>(bt);
      const f16x8 b2 = *reinterpret_cast<const f16x8*>(bt + 2 * BHALF);
      f16x8 ahi[MT], alo[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const char* rec = base + abase + ((m + dy) * HW + dx) * REC;
        ahi[m] = *reinterpret_cast<const f16x8*>(rec);
        alo[m] = *reinterpret_cast<const f16x8*>(rec + 32);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b1, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[m], b2, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[m], b1, acc[m], 0, 0, 0);
    }
    __syncthreads();
  }
  __builtin_amdgcn_s_setprio(0);

  const int cout = a.out.n;
  const int oc = wn * 32 + li;
  const bool ocok = oc < cout;
  constexpr float UNSPLIT = 1.0f / SPLIT;
  float chk = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) chk = fmaf(acc[m][r], 0.f, chk);
  if (__any(chk != chk)) {
    if (lane == 0) atomicOr(a.ovf, 1);
  }
  const float bias = a.bias[oc], scale = a.scale[oc];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int y = y0 + MT * wm + m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (ocok && y < H && x < W) {
        const size_t pix = (size_t)((size_t)b * H + y) * W + x;
        float v = (acc[m][r] * UNSPLIT + bias) * scale;
        if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (a.act == ACT_LRELU) v = (v >= 0.f) ? v : v * 0.2f;
        if (a.res1.p) v = v * a.rs1 + a.res1.p[pix * a.res1.cs + a.res1.c0 + oc];
        if (a.res2.p) v = v * a.rs2 + a.res2.p[pix * a.res2.cs + a.res2.c0 + oc];
        a.out.p[pix * a.out.cs + a.out.c0 + oc] = v;
      }
    }
  }
}

}  // namespace f16x3ws

// plain 3x3 conv, aligned windows, no upsampled source; returns HCF_ERR_UNSUPPORTED otherwise
int launch_conv_f16x3_ws(const ConvArgs& a, hipStream_t st) {
  const int nt = (a.out.n + 31) / 32;
  if (nt < 1 || nt > 2 || a.tC > 0 || a.w2 || a.in_max || !a.ovf || !a.zeros) return HCF_ERR_UNSUPPORTED;
  for (int i = 0; i < a.nsrc; ++i) {
    if (a.src[i].up) return HCF_ERR_UNSUPPORTED;
    if (((a.src[i].cs | a.src[i].c0) & 3) || (reinterpret_cast<uintptr_t>(a.src[i].p) & 15)) return HCF_ERR_UNSUPPORTED;
    if ((long long)a.B * a.H * a.W * a.src[i].cs >= 0x7fffffffLL) return HCF_ERR_UNSUPPORTED;
  }
  const long long nblk = (long long)a.B * ((a.W + 31) / 32) * ((a.H + 7) / 8);
  if (nblk <= 0 || nblk > 0x7fffffffLL) return HCF_ERR_ARG;
  if (nt == 1) hipLaunchKernelGGL(f16x3ws::conv_f16x3_ws_kernel<1>, dim3((unsigned)nblk), dim3(512), 0, st, a);
  else hipLaunchKernelGGL(f16x3ws::conv_f16x3_ws_kernel<2>, dim3((unsigned)nblk), dim3(512), 0, st, a);
  return hipGetLastError() == hipSuccess ? HCF_OK : HCF_ERR_HIP;
}

}  // namespace hcf
