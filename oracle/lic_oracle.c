/*
 * lic_oracle.c -- CPU restatement (plain C, fp32, NCHW like the reference) of the
 * learned-image-compression hot path of achraf-15/neural_image_compression.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path
 * (neural_image_compression_amd/) never imports, links or calls anything in oracle/.
 *
 * Parity status: every function below is pinned by fixtures under tests/golden/ that were
 * generated from the reference's own Python modules (oracle/make_golden.py) EXCEPT the
 * GDN/IGDN arithmetic, which lives in the third-party `compressai` package (unpinned in
 * the reference's requirements.txt:10, absent offline): GDN is "parity unpinned" and
 * follows the public CompressAI definition.
 *
 * Each function cites the reference file:line it restates (paths under /root/reference).
 * Layout: activations NCHW contiguous fp32; conv weights [Cout,Cin,kh,kw];
 * transposed-conv weights [Cin,Cout,kh,kw] (PyTorch conventions, Components.py:6-122).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LIC_API __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int ceil_div(int a, int b) { return (a >= 0) ? (a + b - 1) / b : -((-a) / b); }
static inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

/* ------------------------------------------------------------------------------------------
 * Convolution family.  nn.Conv2d / nn.ConvTranspose2d arithmetic is PyTorch's own
 * (Components.py:10-16,39-45,69-73,99-103; Layers.py:21,38,40,43,74,99,101,103;
 * ParametersModels.py:22-34; ContextModels.py:20).  Weight element is addressed through
 * explicit strides so that two cores serve conv / convT forward and both data gradients.
 * ------------------------------------------------------------------------------------------ */

/* gather: y[b,o,oh,ow] = bias[o] + sum_{i,r,s} x[b,i,oh*st-pad+r,ow*st-pad+s] * w[o*wso+i*wsi+r*kw+s] */
static void conv_gather_core(const float* x, const float* w, long wso, long wsi, const float* bias,
                             float* y, int B, int Ci, int H, int W, int Co, int kh, int kw, int st,
                             int pad, int Ho, int Wo) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < Co; ++o) {
      float* yp = y + ((long)(b * Co + o) * Ho) * Wo;
      const float bv = bias ? bias[o] : 0.0f;
      for (long t = 0; t < (long)Ho * Wo; ++t) yp[t] = bv;
      for (int i = 0; i < Ci; ++i) {
        const float* xp = x + ((long)(b * Ci + i) * H) * W;
        for (int r = 0; r < kh; ++r)
          for (int s = 0; s < kw; ++s) {
            const float wv = w[o * wso + i * wsi + r * kw + s];
            const int ow_lo = imax(0, ceil_div(pad - s, st));
            const int ow_hi = imin(Wo - 1, floor_div(W - 1 + pad - s, st));
            for (int oh = 0; oh < Ho; ++oh) {
              const int ih = oh * st - pad + r;
              if (ih < 0 || ih >= H) continue;
              const float* xr = xp + (long)ih * W - pad + s;
              float* yr = yp + (long)oh * Wo;
              if (st == 1)
                for (int ow = ow_lo; ow <= ow_hi; ++ow) yr[ow] += wv * xr[ow];
              else
                for (int ow = ow_lo; ow <= ow_hi; ++ow) yr[ow] += wv * xr[ow * st];
            }
          }
      }
    }
}

/* scatter: y[b,o,ih*st-pad+r,iw*st-pad+s] += x[b,i,ih,iw] * w[i*wsi+o*wso+r*kw+s]; y starts at bias */
static void conv_scatter_core(const float* x, const float* w, long wsi, long wso, const float* bias,
                              float* y, int B, int Ci, int H, int W, int Co, int kh, int kw, int st,
                              int pad, int Ho, int Wo) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < Co; ++o) {
      float* yp = y + ((long)(b * Co + o) * Ho) * Wo;
      const float bv = bias ? bias[o] : 0.0f;
      for (long t = 0; t < (long)Ho * Wo; ++t) yp[t] = bv;
      for (int i = 0; i < Ci; ++i) {
        const float* xp = x + ((long)(b * Ci + i) * H) * W;
        for (int r = 0; r < kh; ++r)
          for (int s = 0; s < kw; ++s) {
            const float wv = w[i * wsi + o * wso + r * kw + s];
            const int iw_lo = imax(0, ceil_div(pad - s, st));
            const int iw_hi = imin(W - 1, floor_div(Wo - 1 + pad - s, st));
            for (int ih = 0; ih < H; ++ih) {
              const int oy = ih * st - pad + r;
              if (oy < 0 || oy >= Ho) continue;
              const float* xr = xp + (long)ih * W;
              float* yr = yp + (long)oy * Wo - pad + s;
              if (st == 1)
                for (int iw = iw_lo; iw <= iw_hi; ++iw) yr[iw] += wv * xr[iw];
              else
                for (int iw = iw_lo; iw <= iw_hi; ++iw) yr[iw * st] += wv * xr[iw];
            }
          }
      }
    }
}

/* weight gradient: g(a,c,r,s) = sum_{b,h,w} small[b,a,h,w] * large[b,c,h*st-pad+r,w*st-pad+s]
 * -> dw[a*sa + c*sc + r*kw + s].  Row dot products in fp32, totals across rows in fp64. */
static void conv_wgrad_core(const float* small, const float* large, float* dw, long sa, long sc,
                            int B, int Ca, int Hs, int Ws, int Cc, int Hl, int Wl, int kh, int kw,
                            int st, int pad) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int a = 0; a < Ca; ++a)
    for (int c = 0; c < Cc; ++c)
      for (int r = 0; r < kh; ++r)
        for (int s = 0; s < kw; ++s) {
          const int w_lo = imax(0, ceil_div(pad - s, st));
          const int w_hi = imin(Ws - 1, floor_div(Wl - 1 + pad - s, st));
          double tot = 0.0;
          for (int b = 0; b < B; ++b) {
            const float* sp = small + ((long)(b * Ca + a) * Hs) * Ws;
            const float* lp = large + ((long)(b * Cc + c) * Hl) * Wl;
            for (int h = 0; h < Hs; ++h) {
              const int hl = h * st - pad + r;
              if (hl < 0 || hl >= Hl) continue;
              const float* sr = sp + (long)h * Ws;
              const float* lr = lp + (long)hl * Wl - pad + s;
              float acc = 0.0f;
              for (int w = w_lo; w <= w_hi; ++w) acc += sr[w] * lr[w * st];
              tot += (double)acc;
            }
          }
          dw[a * sa + c * sc + r * kw + s] = (float)tot;
        }
}

static void bias_grad(const float* dy, float* db, int B, int C, long HW) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
      const float* p = dy + ((long)(b * C + c)) * HW;
      float acc = 0.0f;
      for (long t = 0; t < HW; ++t) acc += p[t];
      tot += (double)acc;
    }
    db[c] = (float)tot;
  }
}

/* nn.Conv2d forward */
LIC_API void lic_oracle_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                                   int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                   int stride, int pad) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  conv_gather_core(x, w, (long)Cin * kh * kw, (long)kh * kw, bias, y, B, Cin, H, W, Cout, kh, kw,
                   stride, pad, Ho, Wo);
}
/* autograd of nn.Conv2d: dx, dw, db (any may be NULL) */
LIC_API void lic_oracle_conv2d_bwd(const float* x, const float* w, const float* dy, float* dx,
                                   float* dw, float* db, int B, int Cin, int H, int W, int Cout,
                                   int kh, int kw, int stride, int pad) {
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (dx) /* dx[b,i,oh*st-pad+r,..] += dy[b,o,oh,ow] w[o,i,r,s] */
    conv_scatter_core(dy, w, (long)Cin * kh * kw, (long)kh * kw, NULL, dx, B, Cout, Ho, Wo, Cin, kh,
                      kw, stride, pad, H, W);
  if (dw) /* dw[o,i,r,s] = sum dy[b,o,oh,ow] x[b,i,oh*st-pad+r,..] */
    conv_wgrad_core(dy, x, dw, (long)Cin * kh * kw, (long)kh * kw, B, Cout, Ho, Wo, Cin, H, W, kh,
                    kw, stride, pad);
  if (db) bias_grad(dy, db, B, Cout, (long)Ho * Wo);
}
/* nn.ConvTranspose2d forward; w is [Cin,Cout,kh,kw] */
LIC_API void lic_oracle_convT2d_fwd(const float* x, const float* w, const float* bias, float* y,
                                    int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                    int stride, int pad, int out_pad) {
  const int Ho = (H - 1) * stride - 2 * pad + kh + out_pad;
  const int Wo = (W - 1) * stride - 2 * pad + kw + out_pad;
  conv_scatter_core(x, w, (long)Cout * kh * kw, (long)kh * kw, bias, y, B, Cin, H, W, Cout, kh, kw,
                    stride, pad, Ho, Wo);
}
LIC_API void lic_oracle_convT2d_bwd(const float* x, const float* w, const float* dy, float* dx,
                                    float* dw, float* db, int B, int Cin, int H, int W, int Cout,
                                    int kh, int kw, int stride, int pad, int out_pad) {
  const int Ho = (H - 1) * stride - 2 * pad + kh + out_pad;
  const int Wo = (W - 1) * stride - 2 * pad + kw + out_pad;
  if (dx) /* dx[b,i,ih,iw] = sum_{o,r,s} dy[b,o,ih*st-pad+r,..] w[i,o,r,s] */
    conv_gather_core(dy, w, (long)Cout * kh * kw, (long)kh * kw, NULL, dx, B, Cout, Ho, Wo, Cin, kh,
                     kw, stride, pad, H, W);
  if (dw) /* dw[i,o,r,s] = sum x[b,i,ih,iw] dy[b,o,ih*st-pad+r,..] */
    conv_wgrad_core(x, dy, dw, (long)Cout * kh * kw, (long)kh * kw, B, Cin, H, W, Cout, Ho, Wo, kh,
                    kw, stride, pad);
  if (db) bias_grad(dy, db, B, Cout, (long)Ho * Wo);
}

/* nn.LeakyReLU(0.01) (Components.py:70,72,100,102; Layers.py:39,73,100; ParametersModels.py:23,25) */
LIC_API void lic_oracle_leaky_relu_fwd(const float* x, float* y, long n, float slope) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) y[i] = x[i] > 0.0f ? x[i] : x[i] * slope;
}
/* in-place variant keeps only the output; the derivative is recovered from its sign */
LIC_API void lic_oracle_leaky_relu_bwd(const float* y, const float* dy, float* dx, long n, float slope) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) dx[i] = y[i] > 0.0f ? dy[i] : dy[i] * slope;
}

/* ------------------------------------------------------------------------------------------
 * GDN / IGDN -- third-party compressai.layers.gdn.GDN, call sites Components.py:11,13,15,
 * 40,42,44; Layers.py:41,75 (beta_min=1e-6, gamma_init=.1).  PARITY UNPINNED: public
 * CompressAI definition, SURVEY.md Appendix B.
 *   reparam(p; bound, pedestal) = max(p, bound)^2 - pedestal
 *   norm[b,i,h,w] = beta_e[i] + sum_j gamma_e[i,j] x[b,j,h,w]^2
 *   y = x * rsqrt(norm) (GDN)  |  x * sqrt(norm) (IGDN)
 * ------------------------------------------------------------------------------------------ */
LIC_API void lic_oracle_gdn_reparam(const float* p, float* out, long n, float bound, float pedestal) {
  for (long i = 0; i < n; ++i) {
    const float v = p[i] > bound ? p[i] : bound;
    out[i] = v * v - pedestal;
  }
}
/* LowerBound backward: pass iff (p >= bound) or (grad wrt max-output < 0) */
LIC_API void lic_oracle_gdn_reparam_bwd(const float* p, const float* dout, float* dp, long n,
                                        float bound) {
  for (long i = 0; i < n; ++i) {
    const float v = p[i] > bound ? p[i] : bound;
    const float g = dout[i] * 2.0f * v; /* d/dv (v^2 - pedestal) */
    dp[i] = (p[i] >= bound || g < 0.0f) ? g : 0.0f;
  }
}
LIC_API void lic_oracle_gdn_fwd(const float* x, const float* beta_e, const float* gamma_e, float* y,
                                float* norm_out, int B, int C, long HW, int inverse) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    float* nrm = (float*)malloc(sizeof(float) * HW);
    for (int i = 0; i < C; ++i) {
      for (long t = 0; t < HW; ++t) nrm[t] = beta_e[i];
      for (int j = 0; j < C; ++j) {
        const float g = gamma_e[(long)i * C + j];
        const float* xj = x + ((long)b * C + j) * HW;
        for (long t = 0; t < HW; ++t) nrm[t] += g * (xj[t] * xj[t]);
      }
      const float* xi = x + ((long)b * C + i) * HW;
      float* yi = y + ((long)b * C + i) * HW;
      for (long t = 0; t < HW; ++t) {
        const float f = inverse ? sqrtf(nrm[t]) : 1.0f / sqrtf(nrm[t]);
        yi[t] = xi[t] * f;
      }
      if (norm_out) memcpy(norm_out + ((long)b * C + i) * HW, nrm, sizeof(float) * HW);
    }
    free(nrm);
  }
}
/* backward wrt x, beta_e, gamma_e given norm (from forward).  dgamma_e/dbeta_e accumulate in fp64. */
LIC_API void lic_oracle_gdn_bwd(const float* x, const float* norm, const float* gamma_e,
                                const float* dy, float* dx, float* dbeta_e, float* dgamma_e, int B,
                                int C, long HW, int inverse) {
  const long N = (long)B * C * HW;
  float* t = (float*)malloc(sizeof(float) * N); /* t = dL/dnorm */
#pragma omp parallel for schedule(static)
  for (long i = 0; i < N; ++i) {
    const float n = norm[i];
    if (inverse) /* y = x sqrt(n): dy/dn = x * 0.5 / sqrt(n) */
      t[i] = dy[i] * x[i] * 0.5f / sqrtf(n);
    else /* y = x n^-1/2: dy/dn = -0.5 x n^-3/2 */
      t[i] = -0.5f * dy[i] * x[i] / (n * sqrtf(n));
  }
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < C; ++j) {
      const float* xj = x + ((long)b * C + j) * HW;
      const float* nj = norm + ((long)b * C + j) * HW;
      const float* gj = dy + ((long)b * C + j) * HW;
      float* dxj = dx + ((long)b * C + j) * HW;
      for (long p = 0; p < HW; ++p)
        dxj[p] = gj[p] * (inverse ? sqrtf(nj[p]) : 1.0f / sqrtf(nj[p]));
      for (int i = 0; i < C; ++i) { /* d norm_i / d x_j = 2 gamma[i,j] x_j */
        const float g = 2.0f * gamma_e[(long)i * C + j];
        const float* ti = t + ((long)b * C + i) * HW;
        for (long p = 0; p < HW; ++p) dxj[p] += g * ti[p] * xj[p];
      }
    }
  if (dbeta_e) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < C; ++i) {
      double tot = 0.0;
      for (int b = 0; b < B; ++b) {
        const float* ti = t + ((long)b * C + i) * HW;
        for (long p = 0; p < HW; ++p) tot += (double)ti[p];
      }
      dbeta_e[i] = (float)tot;
    }
  }
  if (dgamma_e) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) {
        double tot = 0.0;
        for (int b = 0; b < B; ++b) {
          const float* ti = t + ((long)b * C + i) * HW;
          const float* xj = x + ((long)b * C + j) * HW;
          for (long p = 0; p < HW; ++p) tot += (double)(ti[p] * (xj[p] * xj[p]));
        }
        dgamma_e[(long)i * C + j] = (float)tot;
      }
  }
  free(t);
}

/* ------------------------------------------------------------------------------------------
 * Entropy-parameter activations (ParametersModels.py:43-64).
 *   K == 1: raw [B,2M,h,w] = [mu | sigma_raw];  sigma = softplus(sigma_raw) + 1e-6
 *   K  > 1: raw [B,3KM,h,w] = [w | mu | sigma_raw], channel k*M+m in each third;
 *           weights = softmax over k; sigmas = softplus + 1e-6.
 * F.softplus: beta=1, threshold=20.
 * ------------------------------------------------------------------------------------------ */
static inline float softplus_f(float v) { return v > 20.0f ? v : log1pf(expf(v)); }
static inline float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }
static inline float softplus_grad_f(float v) { return v > 20.0f ? 1.0f : sigmoid_f(v); }

LIC_API void lic_oracle_entropy_params_fwd(const float* raw, float* out, int B, int M, int K,
                                           long HW) {
  const int G = (K == 1) ? 2 : 3;
  const long CH = (long)G * K * M;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    const float* r = raw + b * CH * HW;
    float* o = out + b * CH * HW;
    if (K == 1) {
      memcpy(o, r, sizeof(float) * M * HW);
      for (long t = 0; t < (long)M * HW; ++t) o[M * HW + t] = softplus_f(r[M * HW + t]) + 1e-6f;
    } else {
      const long T = (long)K * M * HW; /* one third */
      for (int m = 0; m < M; ++m)
        for (long p = 0; p < HW; ++p) {
          float mx = -INFINITY;
          for (int k = 0; k < K; ++k) mx = fmaxf(mx, r[((long)k * M + m) * HW + p]);
          float den = 0.0f;
          for (int k = 0; k < K; ++k) den += expf(r[((long)k * M + m) * HW + p] - mx);
          for (int k = 0; k < K; ++k)
            o[((long)k * M + m) * HW + p] = expf(r[((long)k * M + m) * HW + p] - mx) / den;
        }
      memcpy(o + T, r + T, sizeof(float) * T);
      for (long t = 0; t < T; ++t) o[2 * T + t] = softplus_f(r[2 * T + t]) + 1e-6f;
    }
  }
}
LIC_API void lic_oracle_entropy_params_bwd(const float* raw, const float* out, const float* dout,
                                           float* draw, int B, int M, int K, long HW) {
  const int G = (K == 1) ? 2 : 3;
  const long CH = (long)G * K * M;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    const float* r = raw + b * CH * HW;
    const float* o = out + b * CH * HW;
    const float* g = dout + b * CH * HW;
    float* d = draw + b * CH * HW;
    if (K == 1) {
      memcpy(d, g, sizeof(float) * M * HW);
      for (long t = 0; t < (long)M * HW; ++t)
        d[M * HW + t] = g[M * HW + t] * softplus_grad_f(r[M * HW + t]);
    } else {
      const long T = (long)K * M * HW;
      for (int m = 0; m < M; ++m)
        for (long p = 0; p < HW; ++p) {
          float dot = 0.0f;
          for (int k = 0; k < K; ++k) {
            const long a = ((long)k * M + m) * HW + p;
            dot += g[a] * o[a];
          }
          for (int k = 0; k < K; ++k) {
            const long a = ((long)k * M + m) * HW + p;
            d[a] = o[a] * (g[a] - dot);
          }
        }
      memcpy(d + T, g + T, sizeof(float) * T);
      for (long t = 0; t < T; ++t) d[2 * T + t] = g[2 * T + t] * softplus_grad_f(r[2 * T + t]);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Gaussian / Gaussian-mixture conditional likelihood (EntropyModels.py:188-233, utils.py:6-8,
 * clamp EntropyModels.py:29-31, log Models.py:84,87).
 *   Phi(t) = 0.5 (1 + erf(t / sqrt 2));  mass_k = Phi((x+.5-mu)/s) - Phi((x-.5-mu)/s)
 *   p_raw = sum_k w_k mass_k (K==1: w=1);  p = max(p_raw, 1e-9);  logp = log p
 * params layout = output of lic_oracle_entropy_params_fwd.
 * ------------------------------------------------------------------------------------------ */
static inline float gauss_cdf_f(float t) { return 0.5f * (1.0f + erff(t / 1.41421356237309515f)); }
static inline float gauss_pdf_f(float t) { return 0.398942280401432678f * expf(-0.5f * t * t); }

LIC_API void lic_oracle_gmm_likelihood_fwd(const float* x, const float* params, float* p,
                                           float* logp, int B, int M, int K, long HW, float bound) {
  const int G = (K == 1) ? 2 : 3;
  const long CH = (long)G * K * M, T = (long)K * M * HW;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    const float* q = params + b * CH * HW;
    const float* wv = (K == 1) ? NULL : q;
    const float* mu = (K == 1) ? q : q + T;
    const float* sg = (K == 1) ? q + T : q + 2 * T;
    for (int m = 0; m < M; ++m)
      for (long t = 0; t < HW; ++t) {
        const long xi = ((long)b * M + m) * HW + t;
        const float xv = x[xi];
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) {
          const long a = ((long)k * M + m) * HW + t;
          const float up = (xv + 0.5f - mu[a]) / sg[a];
          const float lo = (xv - 0.5f - mu[a]) / sg[a];
          const float mass = gauss_cdf_f(up) - gauss_cdf_f(lo);
          acc = (K == 1) ? mass : acc + wv[a] * mass;
        }
        const float pc = acc > bound ? acc : bound;
        p[xi] = pc;
        logp[xi] = logf(pc);
      }
  }
}
/* dp, dlogp: upstream grads of the two outputs (either may be NULL) */
LIC_API void lic_oracle_gmm_likelihood_bwd(const float* x, const float* params, const float* dp,
                                           const float* dlogp, float* dx, float* dparams, int B,
                                           int M, int K, long HW, float bound) {
  const int G = (K == 1) ? 2 : 3;
  const long CH = (long)G * K * M, T = (long)K * M * HW;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    const float* q = params + b * CH * HW;
    float* dq = dparams + b * CH * HW;
    const float* wv = (K == 1) ? NULL : q;
    const float* mu = (K == 1) ? q : q + T;
    const float* sg = (K == 1) ? q + T : q + 2 * T;
    float* dwv = (K == 1) ? NULL : dq;
    float* dmu = (K == 1) ? dq : dq + T;
    float* dsg = (K == 1) ? dq + T : dq + 2 * T;
    for (int m = 0; m < M; ++m)
      for (long t = 0; t < HW; ++t) {
        const long xi = ((long)b * M + m) * HW + t;
        const float xv = x[xi];
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) {
          const long a = ((long)k * M + m) * HW + t;
          const float up = (xv + 0.5f - mu[a]) / sg[a];
          const float lo = (xv - 0.5f - mu[a]) / sg[a];
          const float mass = gauss_cdf_f(up) - gauss_cdf_f(lo);
          acc = (K == 1) ? mass : acc + wv[a] * mass;
        }
        const float pc = acc > bound ? acc : bound;
        float g = 0.0f; /* grad wrt clamped p */
        if (dp) g += dp[xi];
        if (dlogp) g += dlogp[xi] / pc;
        if (!(acc >= bound)) g = 0.0f; /* clamp_min backward mask: input >= min */
        float dxa = 0.0f;
        for (int k = 0; k < K; ++k) {
          const long a = ((long)k * M + m) * HW + t;
          const float s = sg[a];
          const float up = (xv + 0.5f - mu[a]) / s;
          const float lo = (xv - 0.5f - mu[a]) / s;
          const float pu = gauss_pdf_f(up), pl = gauss_pdf_f(lo);
          const float wk = (K == 1) ? 1.0f : wv[a];
          if (K > 1) dwv[a] = g * (gauss_cdf_f(up) - gauss_cdf_f(lo));
          const float gm = g * wk; /* grad wrt mass_k */
          dmu[a] = -gm * (pu - pl) / s;
          dsg[a] = -gm * (pu * up - pl * lo) / s;
          dxa += gm * (pu - pl) / s;
        }
        dx[xi] = dxa;
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * Factorised entropy bottleneck (EntropyModels.py:49-151): per channel c a 1-3-3-3-1 MLP
 *   L(v): h = v; for i in 0..3: h = softplus(M_i) h + b_i; if i<3: h += tanh(f_i) * tanh(h)
 *   lower = L(x-.5), upper = L(x+.5); s = -sign(lower+upper) (detached)
 *   p_raw = |sigmoid(s*upper) - sigmoid(s*lower)|;  p = max(p_raw, 1e-9);  logp = log p
 * Parameter layout (reference shapes): matrices (C,3,1),(C,3,3),(C,3,3),(C,1,3) raw;
 * biases (C,3,1),(C,3,1),(C,3,1),(C,1,1); factors (C,3,1) x3 raw.
 * Packed here per channel as 58 floats: [M0 3][M1 9][M2 9][M3 3][b0 3][b1 3][b2 3][b3 1]
 * [f0 3][f1 3][f2 3] -- 24 matrix + 10 bias + 9 factor = 43.  (3+9+9+3=24, 3+3+3+1=10, 9.)
 * ------------------------------------------------------------------------------------------ */
#define FE_NPARAM 43
static const int FE_MOFF[4] = {0, 3, 12, 21};
static const int FE_BOFF[4] = {24, 27, 30, 33};
static const int FE_FOFF[3] = {34, 37, 40};
static const int FE_DIN[4] = {1, 3, 3, 3};
static const int FE_DOUT[4] = {3, 3, 3, 1};

typedef struct {
  float h[4][3];   /* input of layer i */
  float pre[4][3]; /* affine output of layer i */
} fe_trace;

static float fe_logits(const float* P, float v, fe_trace* tr) {
  float h[3] = {v, 0, 0};
  for (int i = 0; i < 4; ++i) {
    float pre[3] = {0, 0, 0};
    for (int o = 0; o < FE_DOUT[i]; ++o) {
      float acc = 0.0f;
      for (int k = 0; k < FE_DIN[i]; ++k)
        acc += softplus_f(P[FE_MOFF[i] + o * FE_DIN[i] + k]) * h[k];
      pre[o] = acc + P[FE_BOFF[i] + o];
    }
    if (tr)
      for (int k = 0; k < 3; ++k) {
        tr->h[i][k] = h[k];
        tr->pre[i][k] = pre[k < FE_DOUT[i] ? k : 0];
      }
    if (i < 3)
      for (int o = 0; o < 3; ++o) h[o] = pre[o] + tanhf(P[FE_FOFF[i] + o]) * tanhf(pre[o]);
    else
      h[0] = pre[0];
  }
  return h[0];
}
/* backprop dlogit through L; accumulates parameter grads into dP (double), returns dL/dv */
static float fe_logits_bwd(const float* P, const fe_trace* tr, float dlogit, double* dP) {
  float dh[3] = {dlogit, 0, 0}; /* grad wrt output of layer i (after nonlinearity) */
  for (int i = 3; i >= 0; --i) {
    float dpre[3] = {0, 0, 0};
    if (i < 3) {
      for (int o = 0; o < 3; ++o) {
        const float tf = tanhf(P[FE_FOFF[i] + o]);
        const float tp = tanhf(tr->pre[i][o]);
        dpre[o] = dh[o] * (1.0f + tf * (1.0f - tp * tp));
        dP[FE_FOFF[i] + o] += (double)(dh[o] * tp * (1.0f - tf * tf));
      }
    } else {
      dpre[0] = dh[0];
    }
    float dhin[3] = {0, 0, 0};
    for (int o = 0; o < FE_DOUT[i]; ++o) {
      dP[FE_BOFF[i] + o] += (double)dpre[o];
      for (int k = 0; k < FE_DIN[i]; ++k) {
        const float raw = P[FE_MOFF[i] + o * FE_DIN[i] + k];
        dP[FE_MOFF[i] + o * FE_DIN[i] + k] += (double)(dpre[o] * tr->h[i][k] * softplus_grad_f(raw));
        dhin[k] += dpre[o] * softplus_f(raw);
      }
    }
    for (int k = 0; k < 3; ++k) dh[k] = dhin[k];
  }
  return dh[0];
}
static inline float sign_f(float v) { return (v > 0.0f) - (v < 0.0f); }

/* params: [C][43] packed as above.  x: [B,C,HW] */
LIC_API void lic_oracle_factorized_fwd(const float* x, const float* params, float* p, float* logp,
                                       int B, int C, long HW, float bound) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float* P = params + (long)c * FE_NPARAM;
      for (long t = 0; t < HW; ++t) {
        const long i = ((long)b * C + c) * HW + t;
        const float lo = fe_logits(P, x[i] - 0.5f, NULL);
        const float up = fe_logits(P, x[i] + 0.5f, NULL);
        const float s = -sign_f(lo + up);
        const float pr = fabsf(sigmoid_f(s * up) - sigmoid_f(s * lo));
        const float pc = pr > bound ? pr : bound;
        p[i] = pc;
        logp[i] = logf(pc);
      }
    }
}
LIC_API void lic_oracle_factorized_bwd(const float* x, const float* params, const float* dp,
                                       const float* dlogp, float* dx, float* dparams, int B, int C,
                                       long HW, float bound) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    const float* P = params + (long)c * FE_NPARAM;
    double dP[FE_NPARAM];
    for (int k = 0; k < FE_NPARAM; ++k) dP[k] = 0.0;
    for (int b = 0; b < B; ++b)
      for (long t = 0; t < HW; ++t) {
        const long i = ((long)b * C + c) * HW + t;
        fe_trace trl, tru;
        const float lo = fe_logits(P, x[i] - 0.5f, &trl);
        const float up = fe_logits(P, x[i] + 0.5f, &tru);
        const float s = -sign_f(lo + up);
        const float su = sigmoid_f(s * up), sl = sigmoid_f(s * lo);
        const float diff = su - sl;
        const float pr = fabsf(diff);
        const float pc = pr > bound ? pr : bound;
        float g = 0.0f;
        if (dp) g += dp[i];
        if (dlogp) g += dlogp[i] / pc;
        if (!(pr >= bound)) g = 0.0f;
        const float gd = g * sign_f(diff); /* abs backward */
        const float dup = gd * su * (1.0f - su) * s;
        const float dlo = -gd * sl * (1.0f - sl) * s;
        float dxa = fe_logits_bwd(P, &tru, dup, dP);
        dxa += fe_logits_bwd(P, &trl, dlo, dP);
        dx[i] = dxa;
      }
    for (int k = 0; k < FE_NPARAM; ++k) dparams[(long)c * FE_NPARAM + k] = (float)dP[k];
  }
}
/* channel_logits_cumulative / channel_cdf / channel_pmf (EntropyModels.py:153-184) */
LIC_API void lic_oracle_factorized_channel_logits(const float* params, int ch, const float* xs,
                                                  float* out, long n) {
  const float* P = params + (long)ch * FE_NPARAM;
  for (long i = 0; i < n; ++i) out[i] = fe_logits(P, xs[i], NULL);
}

/* ------------------------------------------------------------------------------------------
 * Quantisation surrogate (Models.py:55-64): training: v + (u - 0.5); eval: round-half-even.
 * ------------------------------------------------------------------------------------------ */
LIC_API void lic_oracle_quantize(const float* v, const float* u, float* out, long n, int training) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) out[i] = training ? v[i] + (u[i] - 0.5f) : rintf(v[i]);
}

/* ------------------------------------------------------------------------------------------
 * Rate-distortion loss (RateDistortionLoss.py:5-49).
 * out[0..10] = loss, bpp_y, bpp_z, bpp_total, mse, psnr, bits_y, bits_z, bits_total, 0, 0
 * mse_img[B], psnr_img[B].
 * ------------------------------------------------------------------------------------------ */
LIC_API void lic_oracle_rd_loss_fwd(const float* logp_y, long ny, const float* logp_z, long nz,
                                    const float* x_hat, const float* x, long nx, int B,
                                    long num_pixels, float lambda_rd, float* out, float* mse_img,
                                    float* psnr_img) {
  const float ln2 = (float)log(2.0);
  double sby = 0, sbz = 0, sbpy = 0, sbpz = 0, smse = 0;
  for (int b = 0; b < B; ++b) {
    double sy = 0, sz = 0, se = 0;
    for (long i = 0; i < ny; ++i) sy += (double)logp_y[b * ny + i];
    for (long i = 0; i < nz; ++i) sz += (double)logp_z[b * nz + i];
    for (long i = 0; i < nx; ++i) {
      const float d = x_hat[b * nx + i] - x[b * nx + i];
      se += (double)(d * d);
    }
    const float bits_y = (float)(-sy) / ln2, bits_z = (float)(-sz) / ln2;
    const float mse_b = (float)(se / (double)nx);
    sby += bits_y;
    sbz += bits_z;
    sbpy += bits_y / (float)num_pixels;
    sbpz += bits_z / (float)num_pixels;
    smse += mse_b;
    mse_img[b] = mse_b;
    psnr_img[b] = -10.0f * log10f(mse_b + 1e-8f);
  }
  const float bpp_y = (float)(sbpy / B), bpp_z = (float)(sbpz / B), mse = (float)(smse / B);
  const float bpp_total = bpp_y + bpp_z;
  out[0] = bpp_total + lambda_rd * (255.0f * 255.0f) * mse;
  out[1] = bpp_y;
  out[2] = bpp_z;
  out[3] = bpp_total;
  out[4] = mse;
  out[5] = -10.0f * log10f(mse + 1e-8f);
  out[6] = (float)(sby / B);
  out[7] = (float)(sbz / B);
  out[8] = (float)((sby + sbz) / B);
}
/* d loss / d {logp_y, logp_z, x_hat} for upstream gradient gl */
LIC_API void lic_oracle_rd_loss_bwd(const float* x_hat, const float* x, long ny, long nz, long nx,
                                    int B, long num_pixels, float lambda_rd, float gl,
                                    float* dlogp_y, float* dlogp_z, float* dx_hat) {
  const float ln2 = (float)log(2.0);
  const float gy = -gl / (ln2 * (float)num_pixels * (float)B);
  for (long i = 0; i < ny * B; ++i) dlogp_y[i] = gy;
  for (long i = 0; i < nz * B; ++i) dlogp_z[i] = gy;
  const float c = gl * lambda_rd * (255.0f * 255.0f) * 2.0f / ((float)nx * (float)B);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < nx * B; ++i) dx_hat[i] = c * (x_hat[i] - x[i]);
}

LIC_API int lic_oracle_version(void) { return 1; }
