/* hcflow_net.c -- plain-C CPU ORACLE of the WHOLE hot path (second, independent restatement beside oracle/hcflow_oracle.py):
 * HCFlowNet_SR / HCFlowNet_Rescaling forward (NLL / encode) and inverse (sampling / decode), NCHW fp32, scalar loops that gcc
 * vectorises, OpenMP over (sample, output channel, row band). TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into
 * oracle/_build/libhcflow_net.so, loaded by oracle/hcflow_c.py; only tests/ and bench.py's cpu_baseline leg use it. Never linked
 * into the product. Every function cites the reference code it restates (paths under codes/models/modules/).
 * Pinned by the reference-generated fixtures tests/golden/net_*.npz (tests/test_oracle_c_net.py): parity pinned.
 *
 * Parameters arrive as (state_dict key, pointer, shape) triples; the layer structure (flow.layers: squeeze / flow steps / split per
 * level, additional flow steps, channel counts, Affine3shift modes) is re-derived HERE from the keys and shapes, the way the
 * reference's constructors register their modules (FlowNet_SR_x4.py:33-64, FlowNet_SR_x8.py:33-70, FlowNet_Rescaling_x4.py:33-67).
 * Random draws are injected (eps per level in sampling order, the dequantisation noise): Basic.py:96-100, HCFlowNet_SR_arch.py:52.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PAR_MIN 262144L      /* elements below which an elementwise / copy loop stays on the calling thread */

/* ------------------------------------------------------------------ tensors (dense NCHW fp32) */
typedef struct { float* d; int B, C, H, W; } T;

static size_t t_n(T t) { return (size_t)t.B * t.C * t.H * t.W; }
static T t_new(int B, int C, int H, int W) {
  T t; t.B = B; t.C = C; t.H = H; t.W = W;
  t.d = (float*)malloc(sizeof(float) * (t_n(t) ? t_n(t) : 1));
  return t;
}
static void t_free(T* t) { free(t->d); t->d = NULL; }
static T t_copy(T x) {
  T o = t_new(x.B, x.C, x.H, x.W);
  const size_t hw = (size_t)x.H * x.W;
#pragma omp parallel for schedule(static) if ((long)t_n(o) > PAR_MIN)
  for (int bc = 0; bc < x.B * x.C; ++bc) memcpy(o.d + (size_t)bc * hw, x.d + (size_t)bc * hw, sizeof(float) * hw);
  return o;
}
static T t_wrap(const float* p, int B, int C, int H, int W) {       /* owning copy of caller memory */
  T o = t_new(B, C, H, W); memcpy(o.d, p, sizeof(float) * t_n(o)); return o;
}
/* x[:, c0:c1] */
static T t_slice(T x, int c0, int c1) {
  T o = t_new(x.B, c1 - c0, x.H, x.W);
  const size_t hw = (size_t)x.H * x.W;
#pragma omp parallel for collapse(2) schedule(static) if ((long)t_n(o) > PAR_MIN)
  for (int b = 0; b < x.B; ++b)
    for (int c = 0; c < o.C; ++c)
      memcpy(o.d + ((size_t)b * o.C + c) * hw, x.d + ((size_t)b * x.C + c0 + c) * hw, sizeof(float) * hw);
  return o;
}
/* x[:, start::2] (thops.split_feature type="cross", thops.py:44-45) */
static T t_cross(T x, int start) {
  T o = t_new(x.B, (x.C - start + 1) / 2, x.H, x.W);
  const size_t hw = (size_t)x.H * x.W;
#pragma omp parallel for collapse(2) schedule(static) if ((long)t_n(o) > PAR_MIN)
  for (int b = 0; b < x.B; ++b)
    for (int c = 0; c < o.C; ++c)
      memcpy(o.d + ((size_t)b * o.C + c) * hw, x.d + ((size_t)b * x.C + start + 2 * c) * hw, sizeof(float) * hw);
  return o;
}
/* torch.cat(parts, 1) */
static T t_cat(const T* parts, int n) {
  int C = 0;
  for (int i = 0; i < n; ++i) C += parts[i].C;
  T o = t_new(parts[0].B, C, parts[0].H, parts[0].W);
  const size_t hw = (size_t)o.H * o.W;
#pragma omp parallel for collapse(2) schedule(static) if ((long)t_n(o) > PAR_MIN)
  for (int b = 0; b < o.B; ++b)
    for (int c = 0; c < C; ++c) {
      int i = 0, c0 = 0;
      while (c >= c0 + parts[i].C) { c0 += parts[i].C; ++i; }
      memcpy(o.d + ((size_t)b * C + c) * hw, parts[i].d + ((size_t)b * parts[i].C + (c - c0)) * hw, sizeof(float) * hw);
    }
  return o;
}
/* F.interpolate(scale_factor=f, mode="nearest") (FlowNet_SR_x4.py:98,117) */
static T t_up(T x, int f) {
  T o = t_new(x.B, x.C, x.H * f, x.W * f);
#pragma omp parallel for schedule(static) if ((long)t_n(o) > PAR_MIN)
  for (int bc = 0; bc < x.B * x.C; ++bc)
    for (int y = 0; y < o.H; ++y)
      for (int xx = 0; xx < o.W; ++xx)
        o.d[((size_t)bc * o.H + y) * o.W + xx] = x.d[((size_t)bc * x.H + y / f) * x.W + xx / f];
  return o;
}

/* ------------------------------------------------------------------ parameters */
typedef struct { char* name; const float* p; int nd; int d[4]; } Par;
typedef struct {
  Par* par; int npar, cap;
  int sr, haar, quant_set; float quant;
  int perm_invconv, coup3, nn_dense;        /* main flow steps: flow_permutation, flow_coupling, nn_module */
  int c_perm_invconv, c_coup3, c_nn_dense;  /* splitOff (conditional) flow steps */
  int nb0, nb1;                             /* RRDB_nb */
  char err[256];
} Net;

void* hcfnet_create(void) { return calloc(1, sizeof(Net)); }
void hcfnet_free(void* h) {
  Net* n = (Net*)h;
  if (!n) return;
  for (int i = 0; i < n->npar; ++i) free(n->par[i].name);
  free(n->par); free(n);
}
const char* hcfnet_error(void* h) { return ((Net*)h)->err; }
int hcfnet_add_param(void* h, const char* name, const float* data, int nd, const int* dims) {
  Net* n = (Net*)h;
  if (n->npar == n->cap) { n->cap = n->cap ? 2 * n->cap : 2048; n->par = (Par*)realloc(n->par, sizeof(Par) * n->cap); }
  Par* q = &n->par[n->npar++];
  q->name = strdup(name); q->p = data; q->nd = nd;
  for (int i = 0; i < 4; ++i) q->d[i] = i < nd ? dims[i] : 1;
  return 0;
}
/* the yml values the constructors read (NetConfig.from_opt): nothing structural beyond what the keys already say */
int hcfnet_configure(void* h, int sr, int haar, float quant, int perm_invconv, int coup3, int nn_dense, int c_perm_invconv,
                     int c_coup3, int c_nn_dense, int nb0, int nb1) {
  Net* n = (Net*)h;
  n->sr = sr; n->haar = haar; n->quant = quant; n->perm_invconv = perm_invconv; n->coup3 = coup3; n->nn_dense = nn_dense;
  n->c_perm_invconv = c_perm_invconv; n->c_coup3 = c_coup3; n->c_nn_dense = c_nn_dense; n->nb0 = nb0; n->nb1 = nb1;
  return 0;
}
static const Par* find(const Net* n, const char* name) {
  for (int i = 0; i < n->npar; ++i)
    if (strcmp(n->par[i].name, name) == 0) return &n->par[i];
  return NULL;
}
static const Par* findf(const Net* n, const char* pre, const char* suf) {
  char key[320];
  snprintf(key, sizeof key, "%s%s", pre, suf);
  return find(n, key);
}
static const float* need(Net* n, const char* pre, const char* suf) {
  const Par* q = findf(n, pre, suf);
  if (!q) { snprintf(n->err, sizeof n->err, "missing parameter %s%s", pre, suf); return NULL; }
  return q->p;
}

/* ------------------------------------------------------------------ convolution */
/* F.conv2d(x, w, bias, stride 1, padding k/2): cross-correlation, as every conv on the path (Basic.py:51,70,350-355,380-384;
 * ConditionalFlow.py:100-103). w is [cout][cin][k][k]. One task = (sample, block of OCB output channels, band of RB rows). */
#define RB 8
#define OCB 4                                        /* output channels per task: each input row is read once for OCB accumulator rows */
static T conv2d(T x, const float* w, const float* bias, int Cout, int k) {
  T o = t_new(x.B, Cout, x.H, x.W);
  const int pad = k / 2, H = x.H, W = x.W, Cin = x.C;
  const int nband = (H + RB - 1) / RB, nocb = (Cout + OCB - 1) / OCB;
  const long ntask = (long)x.B * nocb * nband;
#pragma omp parallel for schedule(dynamic, 2)
  for (long task = 0; task < ntask; ++task) {
    const int band = (int)(task % nband), ob = (int)((task / nband) % nocb), b = (int)(task / ((long)nband * nocb));
    const int ya = band * RB, yb = ya + RB < H ? ya + RB : H;
    const int oc0 = ob * OCB, noc = oc0 + OCB <= Cout ? OCB : Cout - oc0;
    float* op[OCB];
    const float* wp[OCB];
    for (int j = 0; j < OCB; ++j) {
      const int oc = oc0 + (j < noc ? j : 0);
      op[j] = o.d + ((size_t)b * Cout + oc) * H * W;
    }
    for (int j = 0; j < noc; ++j) {
      const float bv = bias ? bias[oc0 + j] : 0.f;
      for (int y = ya; y < yb; ++y)
        for (int xx = 0; xx < W; ++xx) op[j][y * W + xx] = bv;
    }
    for (int ic = 0; ic < Cin; ++ic) {
      const float* ip = x.d + ((size_t)b * Cin + ic) * H * W;
      for (int j = 0; j < OCB; ++j) wp[j] = w + ((size_t)(oc0 + (j < noc ? j : 0)) * Cin + ic) * k * k;
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
          const int dy = ky - pad, dx = kx - pad;
          const int y0 = ya > -dy ? ya : -dy, y1 = yb < H - dy ? yb : H - dy;
          const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? W - dx : W;
          if (noc == OCB) {
            const float w0 = wp[0][ky * k + kx], w1 = wp[1][ky * k + kx], w2 = wp[2][ky * k + kx], w3 = wp[3][ky * k + kx];
            for (int y = y0; y < y1; ++y) {
              float* restrict r0 = op[0] + (size_t)y * W; float* restrict r1 = op[1] + (size_t)y * W;
              float* restrict r2 = op[2] + (size_t)y * W; float* restrict r3 = op[3] + (size_t)y * W;
              const float* restrict irow = ip + (size_t)(y + dy) * W + dx;
              for (int xx = x0; xx < x1; ++xx) {
                const float v = irow[xx];
                r0[xx] += w0 * v; r1[xx] += w1 * v; r2[xx] += w2 * v; r3[xx] += w3 * v;
              }
            }
          } else {
            for (int j = 0; j < noc; ++j) {
              const float wv = wp[j][ky * k + kx];
              for (int y = y0; y < y1; ++y) {
                float* orow = op[j] + (size_t)y * W;
                const float* irow = ip + (size_t)(y + dy) * W + dx;
                for (int xx = x0; xx < x1; ++xx) orow[xx] += wv * irow[xx];
              }
            }
          }
        }
    }
  }
  return o;
}
static T conv_named(Net* n, T x, const char* pre, int with_bias) {
  const Par* w = findf(n, pre, ".weight");
  if (!w) { snprintf(n->err, sizeof n->err, "missing parameter %s.weight", pre); return t_new(0, 0, 0, 0); }
  if (w->d[1] != x.C) { snprintf(n->err, sizeof n->err, "%s.weight expects %d input channels, got %d", pre, w->d[1], x.C); return t_new(0, 0, 0, 0); }
  const float* bias = with_bias ? need(n, pre, ".bias") : NULL;
  return conv2d(x, w->p, bias, w->d[0], w->d[2]);
}

/* ------------------------------------------------------------------ elementwise layers */
/* ActNorm2d: (x + bias) * exp(logs) forward, x * exp(-logs) - bias reverse (ActNorms.py:45-66,87-94), in place */
static void actnorm(T x, const float* bias, const float* logs, int reverse) {
  const size_t hw = (size_t)x.H * x.W;
#pragma omp parallel for collapse(2) schedule(static) if ((long)t_n(x) > PAR_MIN)
  for (int b = 0; b < x.B; ++b)
    for (int c = 0; c < x.C; ++c) {
      float* p = x.d + ((size_t)b * x.C + c) * hw;
      const float e = expf(reverse ? -logs[c] : logs[c]), bc = bias[c];
      if (reverse) for (size_t i = 0; i < hw; ++i) p[i] = p[i] * e - bc;
      else for (size_t i = 0; i < hw; ++i) p[i] = (p[i] + bc) * e;
    }
}
static void relu_(T x) {
  const long n = (long)t_n(x);
#pragma omp parallel for schedule(static) if (n > PAR_MIN)
  for (long i = 0; i < n; ++i) x.d[i] = x.d[i] > 0.f ? x.d[i] : 0.f;
}
static void lrelu_(T x) {
  const long n = (long)t_n(x);
#pragma omp parallel for schedule(static) if (n > PAR_MIN)
  for (long i = 0; i < n; ++i) x.d[i] = x.d[i] >= 0.f ? x.d[i] : 0.2f * x.d[i];
}

/* fp64 Gauss-Jordan inverse with partial pivoting + log|det| (Permutations.py:70 slogdet, :74 inverse(W.double()).float()) */
static int inverse_f64(const double* Win, int n, double* inv, double* logabsdet) {
  double* a = (double*)malloc(sizeof(double) * n * n);
  memcpy(a, Win, sizeof(double) * n * n);
  for (int i = 0; i < n * n; ++i) inv[i] = 0.0;
  for (int i = 0; i < n; ++i) inv[i * n + i] = 1.0;
  double lad = 0.0;
  for (int col = 0; col < n; ++col) {
    int piv = col;
    double best = fabs(a[col * n + col]);
    for (int r = col + 1; r < n; ++r)
      if (fabs(a[r * n + col]) > best) { best = fabs(a[r * n + col]); piv = r; }
    if (best == 0.0) { free(a); return -1; }
    if (piv != col)
      for (int c = 0; c < n; ++c) {
        double t = a[piv * n + c]; a[piv * n + c] = a[col * n + c]; a[col * n + c] = t;
        t = inv[piv * n + c]; inv[piv * n + c] = inv[col * n + c]; inv[col * n + c] = t;
      }
    const double d = a[col * n + col];
    lad += log(fabs(d));
    for (int c = 0; c < n; ++c) { a[col * n + c] /= d; inv[col * n + c] /= d; }
    for (int r = 0; r < n; ++r) {
      if (r == col) continue;
      const double f = a[r * n + col];
      if (f == 0.0) continue;
      for (int c = 0; c < n; ++c) { a[r * n + c] -= f * a[col * n + c]; inv[r * n + c] -= f * inv[col * n + c]; }
    }
  }
  if (logabsdet) *logabsdet = lad;
  free(a);
  return 0;
}
/* z = conv2d(x, M[:, :, None, None]): out[c] = sum_k M[c][k] x[k] per pixel (Permutations.py:99-105) */
static T matmul_channels(T x, const float* M) {
  T o = t_new(x.B, x.C, x.H, x.W);
  const size_t hw = (size_t)x.H * x.W;
  const int C = x.C;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < x.B; ++b)
    for (int c = 0; c < C; ++c) {
      float* op = o.d + ((size_t)b * C + c) * hw;
      for (size_t i = 0; i < hw; ++i) op[i] = 0.f;
      for (int k = 0; k < C; ++k) {
        const float m = M[c * C + k];
        const float* ip = x.d + ((size_t)b * C + k) * hw;
        for (size_t i = 0; i < hw; ++i) op[i] += m * ip[i];
      }
    }
  return o;
}
/* The 1x1 weight of a flow step and its log|det| per pixel. Plain: W (forward), inverse(W.double()).float() (reverse),
 * slogdet(W) (Permutations.py:66-76). LU-decomposed (Permutations.py:78-92): l = l o mask + I, u = u o mask^T + diag(sign_s exp(log_s)),
 * forward w = p (l u), reverse w = inverse(u.double()).float() (inverse(l.double()).float() p^-1), dlogdet = sum(log_s). */
static int invconv_weight(Net* n, const char* pre, int C, int reverse, float* Wout, double* logdet_px) {
  const Par* lw = findf(n, pre, ".l");
  if (!lw) {
    const float* W = need(n, pre, ".weight");
    if (!W) return -1;
    double* Wd = (double*)malloc(sizeof(double) * C * C), *inv = (double*)malloc(sizeof(double) * C * C);
    for (int i = 0; i < C * C; ++i) Wd[i] = W[i];
    int rc = inverse_f64(Wd, C, inv, logdet_px);
    if (rc) snprintf(n->err, sizeof n->err, "%s.weight is singular", pre);
    for (int i = 0; i < C * C; ++i) Wout[i] = reverse ? (float)inv[i] : W[i];
    free(Wd); free(inv);
    return rc;
  }
  const float *l_ = lw->p, *u_ = need(n, pre, ".u"), *log_s = need(n, pre, ".log_s"), *P = need(n, pre, ".p"), *sign_s = need(n, pre, ".sign_s");
  if (!u_ || !log_s || !P || !sign_s) return -1;
  float* l = (float*)calloc((size_t)C * C, sizeof(float)); float* u = (float*)calloc((size_t)C * C, sizeof(float));
  double ld = 0.0;
  for (int i = 0; i < C; ++i) {
    for (int j = 0; j < C; ++j) {
      l[i * C + j] = j < i ? l_[i * C + j] : (i == j ? 1.f : 0.f);
      u[i * C + j] = j > i ? u_[i * C + j] : (i == j ? sign_s[i] * expf(log_s[i]) : 0.f);
    }
    ld += log_s[i];
  }
  if (logdet_px) *logdet_px = ld;
  float* tmp = (float*)calloc((size_t)C * C, sizeof(float));
  int rc = 0;
  if (!reverse) {                                    /* w = p @ (l @ u), fp32 */
    for (int i = 0; i < C; ++i) for (int j = 0; j < C; ++j) { float a = 0.f; for (int k = 0; k < C; ++k) a += l[i * C + k] * u[k * C + j]; tmp[i * C + j] = a; }
    for (int i = 0; i < C; ++i) for (int j = 0; j < C; ++j) { float a = 0.f; for (int k = 0; k < C; ++k) a += P[i * C + k] * tmp[k * C + j]; Wout[i * C + j] = a; }
  } else {                                           /* w = inv(u) @ (inv(l) @ p^-1) */
    double* d = (double*)malloc(sizeof(double) * C * C), *di = (double*)malloc(sizeof(double) * C * C);
    float* li = (float*)malloc(sizeof(float) * C * C), *ui = (float*)malloc(sizeof(float) * C * C), *pi = (float*)malloc(sizeof(float) * C * C);
    for (int i = 0; i < C * C; ++i) d[i] = l[i];
    rc |= inverse_f64(d, C, di, NULL); for (int i = 0; i < C * C; ++i) li[i] = (float)di[i];
    for (int i = 0; i < C * C; ++i) d[i] = u[i];
    rc |= inverse_f64(d, C, di, NULL); for (int i = 0; i < C * C; ++i) ui[i] = (float)di[i];
    for (int i = 0; i < C * C; ++i) d[i] = P[i];     /* p.inverse() of a permutation matrix (fp32 in the reference: exact) */
    rc |= inverse_f64(d, C, di, NULL); for (int i = 0; i < C * C; ++i) pi[i] = (float)di[i];
    for (int i = 0; i < C; ++i) for (int j = 0; j < C; ++j) { float a = 0.f; for (int k = 0; k < C; ++k) a += li[i * C + k] * pi[k * C + j]; tmp[i * C + j] = a; }
    for (int i = 0; i < C; ++i) for (int j = 0; j < C; ++j) { float a = 0.f; for (int k = 0; k < C; ++k) a += ui[i * C + k] * tmp[k * C + j]; Wout[i * C + j] = a; }
    free(d); free(di); free(li); free(ui); free(pi);
    if (rc) snprintf(n->err, sizeof n->err, "%s: singular LU factor", pre);
  }
  free(l); free(u); free(tmp);
  return rc;
}

/* ------------------------------------------------------------------ conv sub-networks */
/* Basic.Conv2d with do_actnorm=True (Basic.py:14-53): bias-free conv then ActNorm */
static T conv_actnorm(Net* n, T x, const char* pre) {
  T y = conv_named(n, x, pre, 0);
  char an[320];
  snprintf(an, sizeof an, "%s.actnorm", pre);
  const float *b = need(n, an, ".bias"), *l = need(n, an, ".logs");
  if (y.d && b && l && y.B) actnorm(y, b, l, 0);
  return y;
}
/* Basic.Conv2dZeros (Basic.py:57-72): (conv3x3 + bias) * exp(logs * 3) */
static T conv_zeros(Net* n, T x, const char* pre) {
  T y = conv_named(n, x, pre, 1);
  const float* logs = need(n, pre, ".logs");
  if (!logs || !y.B) return y;
  const size_t hw = (size_t)y.H * y.W;
#pragma omp parallel for collapse(2) schedule(static) if ((long)t_n(y) > PAR_MIN)
  for (int b = 0; b < y.B; ++b)
    for (int c = 0; c < y.C; ++c) {
      const float e = expf(logs[c] * 3.f);
      float* p = y.d + ((size_t)b * y.C + c) * hw;
      for (size_t i = 0; i < hw; ++i) p[i] *= e;
    }
  return y;
}
/* Basic.FCN.forward (Basic.py:441-447) */
static T fcn(Net* n, T x, const char* pre) {
  char k[320];
  snprintf(k, sizeof k, "%s.conv1", pre); T a = conv_actnorm(n, x, k); relu_(a);
  snprintf(k, sizeof k, "%s.conv2", pre); T b = conv_actnorm(n, a, k); relu_(b); t_free(&a);
  snprintf(k, sizeof k, "%s.conv3", pre); T c = conv_zeros(n, b, k); t_free(&b);
  return c;
}
/* DenseBlock / ResidualDenseBlock body (Basic.py:349-356, 379-385): five 3x3 convs, dense concatenation, LeakyReLU(0.2) x4 */
static T dense5(Net* n, T x, const char* pre) {
  T feats[5]; feats[0] = x;
  char k[320];
  for (int i = 1; i <= 4; ++i) {
    T in = t_cat(feats, i);
    snprintf(k, sizeof k, "%s.conv%d", pre, i);
    feats[i] = conv_named(n, in, k, 1); t_free(&in);
    if (!feats[i].B) { for (int j = 1; j < i; ++j) t_free(&feats[j]); return feats[i]; }
    lrelu_(feats[i]);
  }
  T in = t_cat(feats, 5);
  snprintf(k, sizeof k, "%s.conv5", pre);
  T o = conv_named(n, in, k, 1); t_free(&in);
  for (int j = 1; j <= 4; ++j) t_free(&feats[j]);
  return o;
}
static void axpby_(T y, float a, T x) {
  const long n = (long)t_n(y);
#pragma omp parallel for schedule(static) if (n > PAR_MIN)
  for (long i = 0; i < n; ++i) y.d[i] = y.d[i] * a + x.d[i];
}
/* ResidualDenseBlock.forward (Basic.py:379-385): x5 * 0.2 + x */
static T rdb(Net* n, T x, const char* pre) { T o = dense5(n, x, pre); if (o.B) axpby_(o, 0.2f, x); return o; }
/* RRDB.forward (Basic.py:394-398) */
static T rrdb(Net* n, T x, const char* pre) {
  char k[320];
  snprintf(k, sizeof k, "%s.RDB1", pre); T a = rdb(n, x, k);
  snprintf(k, sizeof k, "%s.RDB2", pre); T b = rdb(n, a, k); t_free(&a);
  snprintf(k, sizeof k, "%s.RDB3", pre); T c = rdb(n, b, k); t_free(&b);
  if (c.B) axpby_(c, 0.2f, x);
  return c;
}
static T coupling_net(Net* n, T x, const char* pre, int dense) { return dense ? dense5(n, x, pre) : fcn(n, x, pre); }

/* ------------------------------------------------------------------ couplings */
/* AffineCoupling (AffineCouplings.py:30-87) / AffineCoupling3shift (:118-160), in place on z; sl[b] += sum of logscale
 * (forward, when sl != NULL). lrv = LR_vs_others of the 3shift form. */
static int coupling(Net* n, T z, const T* u, const char* pre, int three, int dense, int lrv, int reverse, double* sl) {
  char f[320];
  snprintf(f, sizeof f, "%s.f", pre);
  const int C = z.C;
  const size_t hw = (size_t)z.H * z.W;
  int c_lo, c_hi, t_lo, t_hi, affine = 1;          /* conditioning channels [c_lo, c_hi), transformed [t_lo, t_hi) */
  if (!three) { c_lo = 0; c_hi = C / 2; t_lo = C / 2; t_hi = C; }
  else if (lrv) { c_lo = 0; c_hi = 3; t_lo = 3; t_hi = C; }
  else { c_lo = 3; c_hi = C; t_lo = 0; t_hi = 3; affine = 0; }
  T z1 = t_slice(z, c_lo, c_hi);
  T in = z1;
  /* the shift-only reverse branch ignores u (AffineCouplings.py:152) */
  const int use_u = u != NULL && !(three && !lrv && reverse);
  if (use_u) { T parts[2] = {z1, *u}; in = t_cat(parts, 2); }
  T h = coupling_net(n, in, f, dense);
  if (use_u) t_free(&in);
  t_free(&z1);
  if (!h.B) return -1;
  const int nt = t_hi - t_lo;
  if (h.C != (affine ? 2 * nt : nt)) { snprintf(n->err, sizeof n->err, "%s: coupling net returns %d channels for %d", pre, h.C, nt); t_free(&h); return -1; }
  for (int b = 0; b < z.B; ++b) {
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc) if ((long)nt * (long)hw > PAR_MIN)
    for (int j = 0; j < nt; ++j) {
      float* zp = z.d + ((size_t)b * C + t_lo + j) * hw;
      if (affine) {
        const float* sh = h.d + ((size_t)b * h.C + 2 * j) * hw;       /* shift = h[:, 0::2], scale = h[:, 1::2] */
        const float* sc = sh + hw;
        for (size_t i = 0; i < hw; ++i) {
          const float ls = 0.318f * atanf(2.f * sc[i]);               /* AffineCouplings.py:53,83 */
          if (!reverse) { zp[i] = (zp[i] + sh[i]) * expf(ls); acc += ls; }
          else zp[i] = zp[i] * expf(-ls) - sh[i];
        }
      } else {
        const float* sh = h.d + ((size_t)b * h.C + j) * hw;
        for (size_t i = 0; i < hw; ++i) zp[i] = reverse ? zp[i] - sh[i] : zp[i] + sh[i];
      }
    }
    if (sl && affine && !reverse) sl[b] += acc;
  }
  t_free(&h);
  return 0;
}

/* FlowStep.normal_flow (FlowStep.py:40-51): actnorm, permutation, coupling; z replaced; logdet (per sample, may be NULL) updated */
static int flowstep_forward(Net* n, T* z, const T* u, double* logdet, const char* pre, int invconv, int three, int dense, int lrv) {
  char k[320];
  const int C = z->C, pix = z->H * z->W;
  snprintf(k, sizeof k, "%s.actnorm", pre);
  const float *ab = need(n, k, ".bias"), *al = need(n, k, ".logs");
  if (!ab || !al) return -1;
  actnorm(*z, ab, al, 0);
  if (logdet) { double s = 0.0; for (int c = 0; c < C; ++c) s += al[c]; for (int b = 0; b < z->B; ++b) logdet[b] += s * pix; }
  if (invconv) {
    snprintf(k, sizeof k, "%s.permute", pre);
    float* W = (float*)malloc(sizeof(float) * C * C);
    double ld = 0.0;
    if (invconv_weight(n, k, C, 0, W, &ld)) { free(W); return -1; }
    T y = matmul_channels(*z, W); free(W);
    t_free(z); *z = y;
    if (logdet) for (int b = 0; b < z->B; ++b) logdet[b] += ld * pix;
  }
  snprintf(k, sizeof k, "%s.affine", pre);
  return coupling(n, *z, u, k, three, dense, lrv, 0, logdet);
}
/* FlowStep.reverse_flow (FlowStep.py:53-64): coupling^-1, permute^-1, actnorm^-1 */
static int flowstep_inverse(Net* n, T* z, const T* u, const char* pre, int invconv, int three, int dense, int lrv) {
  char k[320];
  const int C = z->C;
  snprintf(k, sizeof k, "%s.affine", pre);
  if (coupling(n, *z, u, k, three, dense, lrv, 1, NULL)) return -1;
  if (invconv) {
    snprintf(k, sizeof k, "%s.permute", pre);
    float* W = (float*)malloc(sizeof(float) * C * C);
    if (invconv_weight(n, k, C, 1, W, NULL)) { free(W); return -1; }
    T y = matmul_channels(*z, W); free(W);
    t_free(z); *z = y;
  }
  snprintf(k, sizeof k, "%s.actnorm", pre);
  const float *ab = need(n, k, ".bias"), *al = need(n, k, ".logs");
  if (!ab || !al) return -1;
  actnorm(*z, ab, al, 1);
  return 0;
}

/* ------------------------------------------------------------------ index ops */
/* squeeze2d / unsqueeze2d factor 2 (Basic.py:127-157): out[b, c*4+i*2+j, h, w] = x[b, c, 2h+i, 2w+j] */
static T squeeze2d(T x) {
  T o = t_new(x.B, 4 * x.C, x.H / 2, x.W / 2);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < x.B; ++b) for (int c = 0; c < x.C; ++c) for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
    for (int h = 0; h < o.H; ++h) for (int w = 0; w < o.W; ++w)
      o.d[(((size_t)b * o.C + c * 4 + i * 2 + j) * o.H + h) * o.W + w] = x.d[(((size_t)b * x.C + c) * x.H + 2 * h + i) * x.W + 2 * w + j];
  return o;
}
static T unsqueeze2d(T x) {
  T o = t_new(x.B, x.C / 4, 2 * x.H, 2 * x.W);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < x.B; ++b) for (int c = 0; c < o.C; ++c) for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
    for (int h = 0; h < x.H; ++h) for (int w = 0; w < x.W; ++w)
      o.d[(((size_t)b * o.C + c) * o.H + 2 * h + i) * o.W + 2 * w + j] = x.d[(((size_t)b * x.C + c * 4 + i * 2 + j) * x.H + h) * x.W + w];
  return o;
}
/* haar_weights[k,0,i,j]: k=1 negates column j=1, k=2 row i=1, k=3 the anti-diagonal (Basic.py:455-464) */
static float haar_sign(int k, int i, int j) { return ((k == 1 && j == 1) || (k == 2 && i == 1) || (k == 3 && i != j)) ? -1.f : 1.f; }
/* HaarDownsampling forward (Basic.py:470-478): out[b, k*C + c] = sum_ij s_k(i,j) x[b,c,2h+i,2w+j] / 4 */
static T haar_forward(T x) {
  T o = t_new(x.B, 4 * x.C, x.H / 2, x.W / 2);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < x.B; ++b) for (int c = 0; c < x.C; ++c) for (int k = 0; k < 4; ++k)
    for (int h = 0; h < o.H; ++h) for (int w = 0; w < o.W; ++w) {
      float acc = 0.f;
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc += haar_sign(k, i, j) * x.d[(((size_t)b * x.C + c) * x.H + 2 * h + i) * x.W + 2 * w + j];
      o.d[(((size_t)b * o.C + k * x.C + c) * o.H + h) * o.W + w] = acc / 4.0f;
    }
  return o;
}
/* HaarDownsampling reverse (Basic.py:479-487): x[b,c,2h+i,2w+j] = sum_k s_k(i,j) y[b, k*C + c, h, w] */
static T haar_inverse(T y) {
  const int C = y.C / 4;
  T o = t_new(y.B, C, 2 * y.H, 2 * y.W);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < y.B; ++b) for (int c = 0; c < C; ++c) for (int h = 0; h < y.H; ++h) for (int w = 0; w < y.W; ++w)
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc += haar_sign(k, i, j) * y.d[(((size_t)b * y.C + k * C + c) * y.H + h) * y.W + w];
      o.d[(((size_t)b * C + c) * o.H + 2 * h + i) * o.W + 2 * w + j] = acc;
    }
  return o;
}

/* ------------------------------------------------------------------ layer plan from the state dict */
enum { L_SQUEEZE, L_STEP, L_SPLIT };
typedef struct { int idx, type, level, C, lrv, n_split; } Layer;
typedef struct { Layer l[256]; int n, L, after[4]; } Plan;
static int make_plan(Net* n, Plan* pl) {
  char k[320];
  pl->n = 0; pl->L = 0;
  for (int level = 0; level < 4; ++level) {
    snprintf(k, sizeof k, "flow.level%d_condFlow.conv_first.weight", level);
    if (!find(n, k)) break;
    pl->L = level + 1;
    int a = 0;
    for (;; ++a) { snprintf(k, sizeof k, "flow.level%d_condFlow.additional_flow_steps.%d.actnorm.bias", level, a); if (!find(n, k)) break; }
    pl->after[level] = a;
  }
  if (!pl->L) { snprintf(n->err, sizeof n->err, "no flow.level*_condFlow parameters"); return -1; }
  int idx = 0, C = 3;
  for (int level = 0; level < pl->L; ++level) {
    Layer s = {idx++, L_SQUEEZE, level, C, 1, 0};
    pl->l[pl->n++] = s;
    C *= 4;
    for (;;) {
      snprintf(k, sizeof k, "flow.layers.%d.actnorm.bias", idx);
      const Par* q = find(n, k);
      if (!q) break;
      if (q->d[1] != C) { snprintf(n->err, sizeof n->err, "flow step %d: ActNorm over %d channels at a %d-channel position", idx, q->d[1], C); return -1; }
      snprintf(k, sizeof k, "flow.layers.%d.affine.f.conv1.weight", idx);
      const Par* w = find(n, k);
      Layer st = {idx++, L_STEP, level, C, (n->sr || !w) ? 1 : (w->d[1] == 3), 0};
      if (pl->n >= 250) return -1;
      pl->l[pl->n++] = st;
    }
    snprintf(k, sizeof k, "flow.level%d_condFlow.f.logs", level);
    const Par* fl = find(n, k);
    if (!fl) { snprintf(n->err, sizeof n->err, "missing %s", k); return -1; }
    const int ns = C - fl->d[0] / 2;
    Layer sp = {idx++, L_SPLIT, level, C, 1, ns};
    pl->l[pl->n++] = sp;
    C = ns;
  }
  return 0;
}

/* ------------------------------------------------------------------ conditional flow */
/* ConditionalFlow.get_conditional_feature_SR / _Rescaling (ConditionalFlow.py:99-110) */
static T cond_features(Net* n, T u, const char* pre) {
  char k[320];
  snprintf(k, sizeof k, "%s.conv_first", pre);
  T first = conv_named(n, u, k, 1);
  if (!first.B) return first;
  T f1 = t_copy(first);
  for (int i = 0; i < n->nb0; ++i) { snprintf(k, sizeof k, "%s.RRDB_trunk0.%d", pre, i); T nx = rrdb(n, f1, k); t_free(&f1); f1 = nx; if (!f1.B) return f1; }
  T t = t_copy(f1);
  for (int i = 0; i < n->nb1; ++i) { snprintf(k, sizeof k, "%s.RRDB_trunk1.%d", pre, i); T nx = rrdb(n, t, k); t_free(&t); t = nx; if (!t.B) return t; }
  snprintf(k, sizeof k, "%s.trunk_conv1", pre);
  T f2 = conv_named(n, t, k, 1); t_free(&t);
  if (!f2.B) return f2;
  axpby_(f2, 1.f, first); t_free(&first);
  if (n->sr) { T parts[2] = {f1, f2}; T o = t_cat(parts, 2); t_free(&f1); t_free(&f2); return o; }
  t_free(&f1);
  return f2;
}
/* u of a level: cat(z, up2(cf[level+1]), up4(cf[level+2])) (FlowNet_SR_x4.py:95-99,114-118; FlowNet_SR_x8.py:104-114) */
static T level_input(T z, const T* cfs, int level, int L) {
  T parts[4]; int np = 0; T ups[4]; int nu = 0;
  parts[np++] = z;
  for (int l2 = level + 1; l2 < L; ++l2) { ups[nu] = t_up(cfs[l2], 1 << (l2 - level)); parts[np++] = ups[nu++]; }
  T o = t_cat(parts, np);
  for (int i = 0; i < nu; ++i) t_free(&ups[i]);
  return o;
}

/* FlowNet.reverse_flow (FlowNet_SR_x4.py:106-123, FlowNet_SR_x8.py:121-144, FlowNet_Rescaling_x4.py:111-128) with
 * ConditionalFlow.forward reverse=True (ConditionalFlow.py:59-69 SR, 84-96 rescaling) and GaussianDiag.sample (Basic.py:96-101:
 * mean + exp(logs) * eps, eps injected, NULL = zeros). eps[k]: the k-th draw in sampling order (deepest level first). */
int hcfnet_inverse(void* h, const float* lr, int B, int hh, int ww, const float* const* eps, int neps, int clamp, float* out) {
  Net* n = (Net*)h;
  n->err[0] = 0;
  Plan pl;
  if (make_plan(n, &pl)) return -1;
  T z = t_wrap(lr, B, 3, hh, ww);
  T cfs[4]; int have[4] = {0, 0, 0, 0};
  int draw = 0, rc = 0;
  char pre[320];
  for (int li = pl.n - 1; li >= 0 && !rc; --li) {
    const Layer* e = &pl.l[li];
    snprintf(pre, sizeof pre, "flow.layers.%d", e->idx);
    if (e->type == L_STEP) rc = flowstep_inverse(n, &z, NULL, pre, n->perm_invconv, n->coup3, n->nn_dense, e->lrv);
    else if (e->type == L_SQUEEZE) { T y = n->haar ? haar_inverse(z) : unsqueeze2d(z); t_free(&z); z = y; }
    else {
      const int level = e->level;
      T u = level_input(z, cfs, level, pl.L);
      snprintf(pre, sizeof pre, "flow.level%d_condFlow", level);
      T cf = cond_features(n, u, pre); t_free(&u);
      if (!cf.B) { rc = -1; break; }
      char k[320];
      snprintf(k, sizeof k, "%s.f", pre);
      T hd = conv_zeros(n, cf, k);
      if (!hd.B) { t_free(&cf); rc = -1; break; }
      T mean = t_cross(hd, 0), s = t_cross(hd, 1); t_free(&hd);
      const float* ep = (eps && draw < neps) ? eps[draw] : NULL;
      ++draw;
      const size_t cnt = t_n(mean);
      for (size_t i = 0; i < cnt; ++i) {
        const float logs = n->sr ? s.d[i] : 0.318f * atanf(2.f * s.d[i]);
        mean.d[i] = mean.d[i] + expf(logs) * (ep ? ep[i] : 0.f);
      }
      t_free(&s);
      T a = mean;
      for (int k2 = pl.after[level] - 1; k2 >= 0 && !rc; --k2) {
        snprintf(k, sizeof k, "%s.additional_flow_steps.%d", pre, k2);
        rc = flowstep_inverse(n, &a, &cf, k, n->c_perm_invconv, n->c_coup3, n->c_nn_dense, 1);
      }
      cfs[level] = cf; have[level] = 1;
      T parts[2] = {z, a};
      T zz = t_cat(parts, 2);                       /* Basic.Split reverse (Basic.py:498-499) */
      t_free(&z); t_free(&a); z = zz;
    }
  }
  if (!rc) {
    const size_t cnt = t_n(z);
    for (size_t i = 0; i < cnt; ++i) out[i] = clamp ? fminf(fmaxf(z.d[i], 0.f), 1.f) : z.d[i];
  }
  t_free(&z);
  for (int l = 0; l < 4; ++l) if (have[l]) t_free(&cfs[l]);
  return rc;
}

/* FlowNet.normal_flow (FlowNet_SR_x4.py:84-101, FlowNet_SR_x8.py:91-116, FlowNet_Rescaling_x4.py:89-106) with
 * ConditionalFlow.forward reverse=False (ConditionalFlow.py:46-57 SR, 70-82 rescaling). logdet: per-sample running value
 * (SR) or NULL; z_out: the LR-sized latent [B,3,h,w]; fake_z[level] (rescaling, may be NULL): [B, C_level - n_split, ...]. */
static int flownet_forward(Net* n, T x, double* logdet, float* z_out, float* const* fake_z) {
  Plan pl;
  if (make_plan(n, &pl)) return -1;
  T z = t_copy(x);
  T ys[4], as[4], cfs[4]; int nlev = 0, have[4] = {0, 0, 0, 0};
  int rc = 0;
  char pre[320];
  for (int li = 0; li < pl.n && !rc; ++li) {
    const Layer* e = &pl.l[li];
    snprintf(pre, sizeof pre, "flow.layers.%d", e->idx);
    if (e->type == L_SQUEEZE) { T y = n->haar ? haar_forward(z) : squeeze2d(z); t_free(&z); z = y; }
    else if (e->type == L_STEP) rc = flowstep_forward(n, &z, NULL, logdet, pre, n->perm_invconv, n->coup3, n->nn_dense, e->lrv);
    else {                                           /* Basic.Split forward (Basic.py:495-497) */
      T a = t_slice(z, e->n_split, z.C), zz = t_slice(z, 0, e->n_split);
      t_free(&z); z = zz;
      ys[nlev] = t_copy(z); as[nlev] = a; ++nlev;
    }
  }
  for (int level = pl.L - 1; level >= 0 && !rc; --level) {   /* hierarchical conditional prior, deepest level first */
    T u = level_input(ys[level], cfs, level, pl.L);
    snprintf(pre, sizeof pre, "flow.level%d_condFlow", level);
    T cf = cond_features(n, u, pre); t_free(&u);
    if (!cf.B) { rc = -1; break; }
    cfs[level] = cf; have[level] = 1;
    T a = as[level];
    char k[320];
    for (int k2 = 0; k2 < pl.after[level] && !rc; ++k2) {
      snprintf(k, sizeof k, "%s.additional_flow_steps.%d", pre, k2);
      rc = flowstep_forward(n, &a, &cf, logdet, k, n->c_perm_invconv, n->c_coup3, n->c_nn_dense, 1);
    }
    as[level] = a;
    if (rc) break;
    snprintf(k, sizeof k, "%s.f", pre);
    T hd = conv_zeros(n, cf, k);
    if (!hd.B) { rc = -1; break; }
    T mean = t_cross(hd, 0), s = t_cross(hd, 1); t_free(&hd);
    const size_t per = (size_t)a.C * a.H * a.W;
    if (n->sr) {                                     /* GaussianDiag.logp (Basic.py:78-94) */
      for (int b = 0; b < a.B; ++b) {
        double acc = 0.0;
        for (size_t i = 0; i < per; ++i) {
          const size_t o = (size_t)b * per + i;
          const float d = a.d[o] - mean.d[o];
          acc += -0.5f * (s.d[o] * 2.f + (d * d) / expf(s.d[o] * 2.f) + 1.8378770664093453f);
        }
        logdet[b] += acc;
      }
    } else if (fake_z && fake_z[level]) {            /* (z - mean) * exp(-logscale) (ConditionalFlow.py:78-82) */
      for (size_t i = 0; i < per * a.B; ++i) fake_z[level][i] = (a.d[i] - mean.d[i]) * expf(-0.318f * atanf(2.f * s.d[i]));
    }
    t_free(&mean); t_free(&s);
  }
  if (!rc) memcpy(z_out, z.d, sizeof(float) * t_n(z));
  t_free(&z);
  for (int l = 0; l < nlev; ++l) { t_free(&ys[l]); t_free(&as[l]); }
  for (int l = 0; l < 4; ++l) if (have[l]) t_free(&cfs[l]);
  return rc;
}

/* HCFlowNet_SR.normal_flow_diracLR (HCFlowNet_SR_arch.py:47-67): x = hr + noise / quant; logdet0 = -log(quant) pixels;
 * zq = Quant(z) (Basic.py:186-196); objective = logdet + logp(lr, logs = -6, zq); nll = mean(-objective / (ln 2 pixels)).
 * Outputs: lr_hat = clamp(zq) [B,3,H/s,W/s], nll (1), and (optional) the pre-quantisation latent z_raw and logdet[B]. */
int hcfnet_sr_forward(void* h, const float* hr, const float* lr, const float* noise, int B, int H, int W, int lh, int lw,
                      float* lr_hat, float* nll, float* z_raw, float* logdet_out) {
  Net* n = (Net*)h;
  n->err[0] = 0;
  T x = t_new(B, 3, H, W);
  const size_t cnt = t_n(x);
  for (size_t i = 0; i < cnt; ++i) x.d[i] = hr[i] + noise[i] / n->quant;
  const double pixels = (double)H * W;
  double* ld = (double*)malloc(sizeof(double) * B);
  for (int b = 0; b < B; ++b) ld[b] = (double)(float)(-log((double)n->quant) * pixels);
  const size_t zc = (size_t)B * 3 * lh * lw;
  float* z = (float*)malloc(sizeof(float) * zc);
  int rc = flownet_forward(n, x, ld, z, NULL);
  t_free(&x);
  if (!rc) {
    if (z_raw) memcpy(z_raw, z, sizeof(float) * zc);
    if (logdet_out) for (int b = 0; b < B; ++b) logdet_out[b] = (float)ld[b];
    const size_t per = zc / B;
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
      double acc = 0.0;
      for (size_t i = 0; i < per; ++i) {
        const size_t o = (size_t)b * per + i;
        const float c = fminf(fmaxf(z[o], 0.f), 1.f);
        const float q = rintf(c * 255.f) / 255.f;                     /* round half to even, as torch.round */
        lr_hat[o] = fminf(fmaxf(q, 0.f), 1.f);
        const float d = q - lr[o], logs = -6.f;
        acc += -0.5f * (logs * 2.f + (d * d) / expf(logs * 2.f) + 1.8378770664093453f);
      }
      const double objective = ld[b] + acc;
      tot += -objective / (log(2.0) * pixels);
    }
    *nll = (float)(tot / B);
  }
  free(z); free(ld);
  return rc;
}

/* HCFlowNet_Rescaling.normal_flow_diracLR (HCFlowNet_Rescaling_arch.py:39-46): (clamp(LR^), fake_z per level) */
int hcfnet_rescale_forward(void* h, const float* hr, int B, int H, int W, int lh, int lw, float* lr_hat, float* const* fake_z) {
  Net* n = (Net*)h;
  n->err[0] = 0;
  T x = t_wrap(hr, B, 3, H, W);
  int rc = flownet_forward(n, x, NULL, lr_hat, fake_z);
  t_free(&x);
  if (!rc) { const size_t zc = (size_t)B * 3 * lh * lw; for (size_t i = 0; i < zc; ++i) lr_hat[i] = fminf(fmaxf(lr_hat[i], 0.f), 1.f); }
  return rc;
}

#ifdef _OPENMP
#include <omp.h>
int hcfnet_threads(int nthreads) { if (nthreads > 0) omp_set_num_threads(nthreads); return omp_get_max_threads(); }
#else
int hcfnet_threads(int nthreads) { (void)nthreads; return 1; }
#endif
