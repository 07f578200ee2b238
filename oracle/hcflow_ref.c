/* hcflow_ref.c -- plain-C CPU ORACLE (scalar loops, NCHW fp32) for the per-op arithmetic of the HCFlow
 * forward / inverse path. TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into
 * oracle/_build/libhcflow_ref.so and loaded by tests/test_oracle_c.py; never linked into the product.
 * Each function cites the reference code it restates (paths under codes/models/modules/).
 * It is checked against the reference-generated fixtures in tests/golden/ (parity pinned).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(b, c, y, x, C, H, W) ((((size_t)(b) * (C) + (c)) * (H) + (y)) * (W) + (x))

/* F.conv2d(x, w, bias, stride 1, padding k/2): cross-correlation, as every conv on the path
 * (Basic.py:51,70,350-355,380-384; ConditionalFlow.py:100-103). w is [cout][cin][k][k]. */
void ref_conv2d(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W, int Cout,
                int k) {
  const int pad = k / 2;
  for (int b = 0; b < B; ++b)
    for (int oc = 0; oc < Cout; ++oc)
      for (int y = 0; y < H; ++y)
        for (int xx = 0; xx < W; ++xx) {
          float acc = bias ? bias[oc] : 0.f;
          for (int ic = 0; ic < Cin; ++ic)
            for (int ky = 0; ky < k; ++ky) {
              const int iy = y + ky - pad;
              if (iy < 0 || iy >= H) continue;
              for (int kx = 0; kx < k; ++kx) {
                const int ix = xx + kx - pad;
                if (ix < 0 || ix >= W) continue;
                acc += x[IDX4(b, ic, iy, ix, Cin, H, W)] * w[(((size_t)oc * Cin + ic) * k + ky) * k + kx];
              }
            }
          out[IDX4(b, oc, y, xx, Cout, H, W)] = acc;
        }
}

/* ActNorm2d: (x + bias) * exp(logs) forward, x * exp(-logs) - bias reverse (ActNorms.py:45-66) */
void ref_actnorm(const float* x, const float* bias, const float* logs, float* out, int B, int C, int HW, int reverse) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float e = expf(reverse ? -logs[c] : logs[c]);
      for (int i = 0; i < HW; ++i) {
        const size_t o = ((size_t)b * C + c) * HW + i;
        out[o] = reverse ? x[o] * e - bias[c] : (x[o] + bias[c]) * e;
      }
    }
}

/* fp64 Gauss-Jordan inverse + log|det|; returns 0 on success (Permutations.py:70,74) */
int ref_inverse_f64(const float* Wm, int n, float* inv_out, double* logabsdet) {
  double* a = (double*)malloc(sizeof(double) * n * n);
  double* inv = (double*)calloc((size_t)n * n, sizeof(double));
  if (!a || !inv) return -1;
  for (int i = 0; i < n * n; ++i) a[i] = Wm[i];
  for (int i = 0; i < n; ++i) inv[i * n + i] = 1.0;
  double lad = 0.0;
  for (int col = 0; col < n; ++col) {
    int piv = col;
    double best = fabs(a[col * n + col]);
    for (int r = col + 1; r < n; ++r)
      if (fabs(a[r * n + col]) > best) { best = fabs(a[r * n + col]); piv = r; }
    if (best == 0.0) { free(a); free(inv); return -2; }
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
  for (int i = 0; i < n * n; ++i) inv_out[i] = (float)inv[i];
  if (logabsdet) *logabsdet = lad;
  free(a);
  free(inv);
  return 0;
}

/* InvertibleConv1x1: z = conv2d(x, M[:, :, None, None]) i.e. out[c] = sum_k M[c][k] x[k] per pixel
 * (Permutations.py:99-105); pass W for forward, inverse(W) for reverse. */
void ref_invconv(const float* x, const float* M, float* out, int B, int C, int HW) {
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < HW; ++i)
      for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        for (int k = 0; k < C; ++k) acc += M[c * C + k] * x[((size_t)b * C + k) * HW + i];
        out[((size_t)b * C + c) * HW + i] = acc;
      }
}

/* AffineCoupling given h = f(z1 [,u]) : channels [ns, C) transformed with (shift, scale) = h[0::2], h[1::2],
 * logscale = 0.318 atan(2 scale) (AffineCouplings.py:30-87). logdet[b] += sum logscale (forward only). */
void ref_affine_coupling(const float* z, const float* h, float* out, float* logdet, int B, int C, int ns, int HW,
                         int reverse) {
  const int n2 = C - ns;
  for (int b = 0; b < B; ++b) {
    double ld = 0.0;
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < HW; ++i) {
        const size_t o = ((size_t)b * C + c) * HW + i;
        if (c < ns) { out[o] = z[o]; continue; }
        const int j = c - ns;
        const float shift = h[((size_t)b * 2 * n2 + 2 * j) * HW + i];
        const float scale = h[((size_t)b * 2 * n2 + 2 * j + 1) * HW + i];
        const float ls = 0.318f * atanf(2.f * scale);
        if (!reverse) { out[o] = (z[o] + shift) * expf(ls); ld += ls; }
        else out[o] = z[o] * expf(-ls) - shift;
      }
    if (logdet && !reverse) logdet[b] = (float)ld;
  }
}

/* squeeze2d / unsqueeze2d factor 2: out[b, c*4+i*2+j, h, w] = x[b, c, 2h+i, 2w+j] (Basic.py:127-157) */
void ref_squeeze2d(const float* x, float* out, int B, int C, int H, int W) {
  const int H2 = H / 2, W2 = W / 2;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          for (int h = 0; h < H2; ++h)
            for (int w = 0; w < W2; ++w)
              out[IDX4(b, c * 4 + i * 2 + j, h, w, 4 * C, H2, W2)] = x[IDX4(b, c, 2 * h + i, 2 * w + j, C, H, W)];
}

void ref_unsqueeze2d(const float* x, float* out, int B, int C4, int H, int W) {
  const int C = C4 / 4;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
              out[IDX4(b, c, 2 * h + i, 2 * w + j, C, 2 * H, 2 * W)] = x[IDX4(b, c * 4 + i * 2 + j, h, w, C4, H, W)];
}

static float haar_sign(int k, int i, int j) {
  /* haar_weights[k,0,i,j]: k=1 negates column j=1, k=2 row i=1, k=3 the anti-diagonal (Basic.py:455-464) */
  return ((k == 1 && j == 1) || (k == 2 && i == 1) || (k == 3 && i != j)) ? -1.f : 1.f;
}

/* HaarDownsampling forward: out[b, k*C + c, h, w] = sum_ij s_k(i,j) x[b,c,2h+i,2w+j] / 4 (Basic.py:470-478) */
void ref_haar_forward(const float* x, float* out, int B, int C, int H, int W) {
  const int H2 = H / 2, W2 = W / 2;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < 4; ++k)
        for (int h = 0; h < H2; ++h)
          for (int w = 0; w < W2; ++w) {
            float acc = 0.f;
            for (int i = 0; i < 2; ++i)
              for (int j = 0; j < 2; ++j) acc += haar_sign(k, i, j) * x[IDX4(b, c, 2 * h + i, 2 * w + j, C, H, W)];
            out[IDX4(b, k * C + c, h, w, 4 * C, H2, W2)] = acc / 4.0f;
          }
}

/* HaarDownsampling reverse: x[b,c,2h+i,2w+j] = sum_k s_k(i,j) y[b, k*C + c, h, w] (Basic.py:479-487) */
void ref_haar_inverse(const float* y, float* out, int B, int C4, int H, int W) {
  const int C = C4 / 4;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
          for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
              float acc = 0.f;
              for (int k = 0; k < 4; ++k) acc += haar_sign(k, i, j) * y[IDX4(b, k * C + c, h, w, C4, H, W)];
              out[IDX4(b, c, 2 * h + i, 2 * w + j, C, 2 * H, 2 * W)] = acc;
            }
}

/* GaussianDiag.logp: sum_chw -0.5 (2 logs + (x-mean)^2 / exp(2 logs) + ln 2pi) (Basic.py:78-94) */
void ref_gauss_logp(const float* mean, const float* logs, const float* x, float* out, int B, int CHW) {
  for (int b = 0; b < B; ++b) {
    double acc = 0.0;
    for (int i = 0; i < CHW; ++i) {
      const size_t o = (size_t)b * CHW + i;
      const float d = x[o] - mean[o];
      acc += -0.5f * (logs[o] * 2.f + (d * d) / expf(logs[o] * 2.f) + 1.8378770664093453f);
    }
    out[b] = (float)acc;
  }
}

/* Basic.Quant.forward: round(clamp(x,0,1)*255)/255, round half to even (Basic.py:187-191) */
void ref_quant(const float* x, float* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const float c = fminf(fmaxf(x[i], 0.f), 1.f);
    out[i] = rintf(c * 255.f) / 255.f;
  }
}

static void relu_inplace(float* x, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f;
}

/* Basic.FCN.forward (Basic.py:441-447): relu(AN(conv3x3)) -> relu(AN(conv1x1)) -> (conv3x3 + b) * exp(3 logs) */
void ref_fcn(const float* x, int B, int Cin, int H, int W, int hid, int Cout, const float* w1, const float* b1,
             const float* l1, const float* w2, const float* b2, const float* l2, const float* w3, const float* b3,
             const float* l3, float* out) {
  const size_t nh = (size_t)B * hid * H * W;
  float* t1 = (float*)malloc(nh * sizeof(float));
  float* t2 = (float*)malloc(nh * sizeof(float));
  ref_conv2d(x, w1, NULL, t1, B, Cin, H, W, hid, 3);
  ref_actnorm(t1, b1, l1, t1, B, hid, H * W, 0);
  relu_inplace(t1, nh);
  ref_conv2d(t1, w2, NULL, t2, B, hid, H, W, hid, 1);
  ref_actnorm(t2, b2, l2, t2, B, hid, H * W, 0);
  relu_inplace(t2, nh);
  ref_conv2d(t2, w3, b3, out, B, hid, H, W, Cout, 3);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < Cout; ++c) {
      const float e = expf(l3[c] * 3.f);
      for (int i = 0; i < H * W; ++i) out[((size_t)b * Cout + c) * H * W + i] *= e;
    }
  free(t1);
  free(t2);
}

/* FlowStep.reverse_flow with an FCN coupling net and invconv (FlowStep.py:53-64); u may be NULL (cond = 0) */
int ref_flowstep_inverse(const float* z, const float* u, int cond, float* out, int B, int C, int H, int W, int hid,
                         const float* an_bias, const float* an_logs, const float* Wm, const float* w1, const float* b1,
                         const float* l1, const float* w2, const float* b2, const float* l2, const float* w3,
                         const float* b3, const float* l3) {
  const int ns = C / 2, HW = H * W, fin = ns + cond, fout = (C - ns) * 2;
  float* in = (float*)malloc((size_t)B * fin * HW * sizeof(float));
  float* h = (float*)malloc((size_t)B * fout * HW * sizeof(float));
  float* t = (float*)malloc((size_t)B * C * HW * sizeof(float));
  float* Wi = (float*)malloc((size_t)C * C * sizeof(float));
  for (int b = 0; b < B; ++b) {
    memcpy(in + (size_t)b * fin * HW, z + (size_t)b * C * HW, (size_t)ns * HW * sizeof(float));
    if (cond) memcpy(in + ((size_t)b * fin + ns) * HW, u + (size_t)b * cond * HW, (size_t)cond * HW * sizeof(float));
  }
  ref_fcn(in, B, fin, H, W, hid, fout, w1, b1, l1, w2, b2, l2, w3, b3, l3, h);
  ref_affine_coupling(z, h, t, NULL, B, C, ns, HW, 1);
  int rc = ref_inverse_f64(Wm, C, Wi, NULL);
  if (rc == 0) {
    ref_invconv(t, Wi, out, B, C, HW);
    ref_actnorm(out, an_bias, an_logs, out, B, C, HW, 1);
  }
  free(in); free(h); free(t); free(Wi);
  return rc;
}
