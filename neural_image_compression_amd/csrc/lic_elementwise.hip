// HBM-bound kernels of liblic_hip.so (gfx950): layout helpers, GDN parameter transforms,
// quantisation surrogate, entropy-parameter activations, Gaussian / mixture / factorised
// likelihoods (forward + backward) and the rate-distortion reductions.
// All are grid-stride, 16-byte vectorised where the layout allows, wave64 reductions.
#include "lic_common.h"
#include "lic_patch.h"

#define EW_BLOCK 256
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// layout helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EW_BLOCK) void permute3_kernel(const float* src, float* dst, int n0, int n1,
                                                            int n2, long s0, long s1, long s2, long d0,
                                                            long d1, long d2) {
  const long total = (long)n0 * n1 * n2;
  for (long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * EW_BLOCK) {
    const int k = (int)(i % n2);
    const long t = i / n2;
    const int j = (int)(t % n1);
    const int a = (int)(t / n1);
    dst[a * d0 + j * d1 + k * d2] = src[a * s0 + j * s1 + k * s2];
  }
}
LIC_EXPORT int lic_permute3(const float* src, float* dst, int32_t n0, int32_t n1, int32_t n2,
                            int64_t s0, int64_t s1, int64_t s2, int64_t d0, int64_t d1, int64_t d2,
                            lic_stream_t stream) {
  if (!src || !dst || n0 <= 0 || n1 <= 0 || n2 <= 0) return LIC_ERR_INVALID;
  const long total = (long)n0 * n1 * n2;
  hipLaunchKernelGGL(permute3_kernel, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0,
                     (hipStream_t)stream, src, dst, n0, n1, n2, (long)s0, (long)s1, (long)s2, (long)d0,
                     (long)d1, (long)d2);
  return lic_check_launch();
}

__global__ __launch_bounds__(EW_BLOCK) void im2col_kernel(const float* x, float* col, int B, int H, int W,
                                                          int C, int Ho, int Wo, int kh, int kw,
                                                          int stride, int pad, int Kpad) {
  const long total = (long)B * Ho * Wo * Kpad;
  const int K = kh * kw * C;
  for (long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * EW_BLOCK) {
    const int k = (int)(i % Kpad);
    const long pix = i / Kpad;
    float v = 0.0f;
    if (k < K) {
      const int c = k % C, tap = k / C;
      const int r = tap / kw, s = tap - r * kw;
      const int ow = (int)(pix % Wo);
      const long t = pix / Wo;
      const int oh = (int)(t % Ho);
      const int b = (int)(t / Ho);
      const int ih = oh * stride - pad + r, iw = ow * stride - pad + s;
      if (ih >= 0 && iw >= 0 && ih < H && iw < W) v = x[(((long)b * H + ih) * W + iw) * C + c];
    }
    col[i] = v;
  }
}
LIC_EXPORT int lic_im2col(const float* x, float* col, int32_t B, int32_t H, int32_t W, int32_t C,
                          int32_t Ho, int32_t Wo, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                          int32_t Kpad, lic_stream_t stream) {
  if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || Kpad < kh * kw * C)
    return LIC_ERR_INVALID;
  const long total = (long)B * Ho * Wo * Kpad;
  if (Kpad <= LIC_PATCH_MAXK && Kpad % 4 == 0 && total / 4 < 0x7FFFFFFFL && al16(col) && kh < 256 && kw < 256 &&
      C < 256 && (long)H * W * C < 0x7FFFFFFFL)
    hipLaunchKernelGGL((im2col_vec_kernel<float, 4>), dim3(ew_grid(total / 4, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, col, (unsigned)((long)B * Ho * Wo), H, W, C, Ho, Wo, kh, kw, stride, pad,
                       Kpad);
  else
    hipLaunchKernelGGL(im2col_kernel, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0,
                       (hipStream_t)stream, x, col, B, H, W, C, Ho, Wo, kh, kw, stride, pad, Kpad);
  return lic_check_launch();
}

__global__ __launch_bounds__(EW_BLOCK) void col2im_kernel(const float* col, const float* bias, float* out,
                                                          int B, int Hi, int Wi, int C, int Ho, int Wo,
                                                          int kh, int kw, int stride, int pad, int Kpad) {
  const long total = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * EW_BLOCK) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int ox = (int)(pix % Wo);
    const long t = pix / Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float v = bias ? bias[c] : 0.0f;
    for (int r = 0; r < kh; ++r) {
      const int nh = oy + pad - r;
      if (nh < 0 || (nh % stride) != 0) continue;
      const int ih = nh / stride;
      if (ih >= Hi) continue;
      for (int s = 0; s < kw; ++s) {
        const int nw = ox + pad - s;
        if (nw < 0 || (nw % stride) != 0) continue;
        const int iw = nw / stride;
        if (iw >= Wi) continue;
        v += col[(((long)b * Hi + ih) * Wi + iw) * Kpad + (r * kw + s) * C + c];
      }
    }
    out[i] = v;
  }
}
LIC_EXPORT int lic_col2im(const float* col, const float* bias, float* out, int32_t B, int32_t Hi,
                          int32_t Wi, int32_t C, int32_t Ho, int32_t Wo, int32_t kh, int32_t kw,
                          int32_t stride, int32_t pad, int32_t Kpad, lic_stream_t stream) {
  if (!col || !out || B <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || stride < 1 ||
      Kpad < kh * kw * C)
    return LIC_ERR_INVALID;
  const long total = (long)B * Ho * Wo * C;
  if (total < 0x7FFFFFFFL && (long)B * Hi * Wi < 0x7FFFFFFFL)
    hipLaunchKernelGGL((col2im_fast_kernel<float>), dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, col,
                       bias, out, (unsigned)total, Hi, Wi, C, Ho, Wo, kh, kw, stride, pad, Kpad);
  else
    hipLaunchKernelGGL(col2im_kernel, dim3(ew_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0,
                       (hipStream_t)stream, col, bias, out, B, Hi, Wi, C, Ho, Wo, kh, kw, stride, pad, Kpad);
  return lic_check_launch();
}

// ---------------------------------------------------------------------------------------------
// column sums (bias gradients): two deterministic stages
// ---------------------------------------------------------------------------------------------
#define CS_MAXCHUNK 256
// block = 16 column groups (float4 = 64 columns) x 16 row lanes; grid (ceil(C/64), nchunk)
__global__ __launch_bounds__(256) void colsum_stage1(const float* in, long ld, long P, int C, float* part,
                                                     int nchunk, int vec) {
  __shared__ float red[16][64 + 4];
  const int cg = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cg * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    if (vec) {
      for (long pr = (long)blockIdx.y * 16 + ry; pr < P; pr += (long)nchunk * 16)
        acc += *reinterpret_cast<const f32x4*>(in + pr * ld + c);
    } else {
      for (long pr = (long)blockIdx.y * 16 + ry; pr < P; pr += (long)nchunk * 16)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < C) acc[e] += in[pr * ld + c + e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[ry][cg * 4 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
    if (cc < C) part[(long)blockIdx.y * C + cc] = s;
  }
}
// C == 3 with packed rows (the RGB head's bias gradient over B*H*W pixels): the vector path above needs
// C % 4 == 0 and the scalar one keeps 16 lanes of a block busy.  Here a lane takes 4 rows = 12 floats as three
// 16-byte loads; element e of the twelve belongs to channel e % 3.
__global__ __launch_bounds__(256) void colsum3_stage1(const float* in, long P, float* part, int nchunk) {
  __shared__ float red[4][3];
  const long ngroup = P / 4;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < ngroup; g += (long)nchunk * 256) {
    const f32x4* q = reinterpret_cast<const f32x4*>(in + g * 12);
    a0 += q[0];
    a1 += q[1];
    a2 += q[2];
  }
  // flat e = 0..11 -> channel e % 3: a0 = (0,1,2,0), a1 = (1,2,0,1), a2 = (2,0,1,2)
  float c0 = a0[0] + a0[3] + a1[2] + a2[1];
  float c1 = a0[1] + a1[0] + a1[3] + a2[2];
  float c2 = a0[2] + a1[1] + a2[0] + a2[3];
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long r = ngroup * 4; r < P; ++r) {
      c0 += in[r * 3];
      c1 += in[r * 3 + 1];
      c2 += in[r * 3 + 2];
    }
  c0 = wave_sum(c0);
  c1 = wave_sum(c1);
  c2 = wave_sum(c2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[w][0] = c0;
    red[w][1] = c1;
    red[w][2] = c2;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    part[(long)blockIdx.x * 3 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// block = 16 columns x 16 chunk lanes (short dependent chains: this stage is pure latency)
__global__ __launch_bounds__(256) void colsum_stage2(const float* part, int C, int nchunk, float scale,
                                                     float* out) {
  __shared__ double red[16][17];
  const int cx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  double acc = 0.0;
  if (c < C)
    for (int y = ly; y < nchunk; y += 16) acc += (double)part[(long)y * C + c];
  red[ly][cx] = acc;
  __syncthreads();
  if (ly == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cx];
    out[c] = (float)(t * (double)scale);
  }
}
static int colsum_chunks(int64_t P) {
  int64_t n = cdiv64(P, 16 * 8);
  if (n > CS_MAXCHUNK) n = CS_MAXCHUNK;
  if (n < 1) n = 1;
  return (int)n;
}
LIC_EXPORT size_t lic_colsum_workspace_bytes(int64_t P, int32_t C) {
  if (P <= 0 || C <= 0) return 0;
  return (size_t)colsum_chunks(P) * C * sizeof(float);
}
// stage 1 only; `job` receives stage 2 for a later lic_reduce_batch
LIC_EXPORT int lic_colsum_partial(const float* in, int64_t ld, int64_t P, int32_t C, float scale, float* out,
                                  void* workspace, size_t workspace_bytes, lic_reduce_job* job, lic_stream_t stream) {
  if (!in || !out || !workspace || !job || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  const int nchunk = colsum_chunks(P);
  if (workspace_bytes < (size_t)nchunk * C * sizeof(float)) return LIC_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int vec = (C % 4 == 0) && (ld % 4 == 0) && al16(in);
  if (C == 3 && ld == 3 && al16(in))
    hipLaunchKernelGGL(colsum3_stage1, dim3(nchunk), dim3(256), 0, s, in, (long)P, (float*)workspace, nchunk);
  else
    hipLaunchKernelGGL(colsum_stage1, dim3((C + 63) / 64, nchunk), dim3(256), 0, s, in, (long)ld, (long)P, C,
                       (float*)workspace, nchunk, vec);
  *job = lic_reduce_job{};
  job->src = (const float*)workspace;
  job->dst = out;
  job->kind = LIC_REDUCE_COLUMNS;
  job->splitk = nchunk;
  job->Cn = C;
  job->scale = scale;
  return lic_check_launch();
}
LIC_EXPORT int lic_colsum(const float* in, int64_t ld, int64_t P, int32_t C, float scale, float* out,
                          void* workspace, size_t workspace_bytes, lic_stream_t stream) {
  if (!in || !out || !workspace || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  const int nchunk = colsum_chunks(P);
  if (workspace_bytes < (size_t)nchunk * C * sizeof(float)) return LIC_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int vec = (C % 4 == 0) && (ld % 4 == 0) && al16(in);
  if (C == 3 && ld == 3 && al16(in))
    hipLaunchKernelGGL(colsum3_stage1, dim3(nchunk), dim3(256), 0, s, in, (long)P, (float*)workspace, nchunk);
  else
    hipLaunchKernelGGL(colsum_stage1, dim3((C + 63) / 64, nchunk), dim3(256), 0, s, in, (long)ld, (long)P, C,
                       (float*)workspace, nchunk, vec);
  int rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  hipLaunchKernelGGL(colsum_stage2, dim3((C + 15) / 16), dim3(256), 0, s, (const float*)workspace, C,
                     nchunk, scale, out);
  return lic_check_launch();
}

// ---------------------------------------------------------------------------------------------
// small elementwise ops
// ---------------------------------------------------------------------------------------------
template <typename F>
__global__ __launch_bounds__(EW_BLOCK) void ew_kernel(long n, F f) {
  for (long i = (long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * EW_BLOCK) f(i);
}
template <typename F>
static int ew_launch(long n, lic_stream_t stream, F f) {
  if (n <= 0) return LIC_OK;
  hipLaunchKernelGGL((ew_kernel<F>), dim3(ew_grid(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, n, f);
  return lic_check_launch();
}
// 16-byte vectorised variant: f4(i4) handles elements [4*i4, 4*i4+4); tail by f1
template <typename F4, typename F1>
static int ew_launch_vec(long n, bool can_vec, lic_stream_t stream, F4 f4, F1 f1) {
  if (n <= 0) return LIC_OK;
  if (!can_vec) return ew_launch(n, stream, f1);
  const long n4 = n / 4;
  int rc = ew_launch(n4, stream, f4);
  if (rc != LIC_OK) return rc;
  const long tail0 = n4 * 4;
  if (tail0 < n) return ew_launch(n - tail0, stream, [=] __device__(long i) { f1(tail0 + i); });
  return LIC_OK;
}

LIC_EXPORT int lic_mul_inplace(float* w, const float* mask, int64_t n, lic_stream_t stream) {
  if (!w || !mask || n < 0) return LIC_ERR_INVALID;
  return ew_launch(n, stream, [=] __device__(long i) { w[i] *= mask[i]; });
}

LIC_EXPORT int lic_leaky_bwd(const float* y, const float* dy, float* dx, int64_t n, float slope,
                             lic_stream_t stream) {
  if (!y || !dy || !dx || n < 0) return LIC_ERR_INVALID;
  return ew_launch_vec(
      n, al16(y) && al16(dy) && al16(dx), stream,
      [=] __device__(long i) {
        const f32x4 a = reinterpret_cast<const f32x4*>(y)[i], g = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] > 0.0f ? g[e] : g[e] * slope;
        reinterpret_cast<f32x4*>(dx)[i] = o;
      },
      [=] __device__(long i) { dx[i] = y[i] > 0.0f ? dy[i] : dy[i] * slope; });
}

LIC_EXPORT int lic_gdn_reparam(const float* p, float* out, int64_t n, float bound, float pedestal,
                               lic_stream_t stream) {
  if (!p || !out || n < 0) return LIC_ERR_INVALID;
  return ew_launch(n, stream, [=] __device__(long i) { out[i] = lic_reparam(p[i], bound, pedestal); });
}
LIC_EXPORT int lic_gdn_reparam_bwd(const float* p, const float* dout, float* dp, int64_t n, float bound,
                                   lic_stream_t stream) {
  if (!p || !dout || !dp || n < 0) return LIC_ERR_INVALID;
  return ew_launch(n, stream, [=] __device__(long i) {
    const float v = p[i] > bound ? p[i] : bound;
    const float g = dout[i] * 2.0f * v;
    dp[i] = (p[i] >= bound || g < 0.0f) ? g : 0.0f;
  });
}

// the same for the two parameters of one GDN (beta [C], gamma [C][C]) in ONE launch: element i < nb is beta's,
// the rest gamma's (either half may be absent: n = 0)
LIC_EXPORT int lic_gdn_reparam_bwd2(const float* pb, const float* dob, float* dpb, int64_t nb, float bound_b,
                                    const float* pg, const float* dog, float* dpg, int64_t ng, float bound_g,
                                    lic_stream_t stream) {
  if (nb < 0 || ng < 0 || (nb > 0 && (!pb || !dob || !dpb)) || (ng > 0 && (!pg || !dog || !dpg))) return LIC_ERR_INVALID;
  if (nb + ng == 0) return LIC_OK;
  return ew_launch(nb + ng, stream, [=] __device__(long i) {
    const bool isb = i < nb;
    const long j = isb ? i : i - nb;
    const float pv = isb ? pb[j] : pg[j], dv = isb ? dob[j] : dog[j], bound = isb ? bound_b : bound_g;
    const float v = pv > bound ? pv : bound;
    const float g = dv * 2.0f * v;
    (isb ? dpb : dpg)[j] = (pv >= bound || g < 0.0f) ? g : 0.0f;
  });
}

__device__ __forceinline__ float gdn_t1(float g, float x, float n, int inverse) {
  const float rs = __builtin_amdgcn_rsqf(n);  // v_rsq_f32, 1 ulp
  return inverse ? 0.5f * g * x * rs : -0.5f * g * x * rs * (rs * rs);
}
LIC_EXPORT int lic_gdn_dnorm(const float* g, const float* x, const float* norm, float* t, int64_t n,
                             int32_t inverse, lic_stream_t stream) {
  if (!g || !x || !norm || !t || n < 0) return LIC_ERR_INVALID;
  return ew_launch_vec(
      n, al16(g) && al16(x) && al16(norm) && al16(t), stream,
      [=] __device__(long i) {
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i], xv = reinterpret_cast<const f32x4*>(x)[i],
                    nv = reinterpret_cast<const f32x4*>(norm)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gdn_t1(gv[e], xv[e], nv[e], inverse);
        reinterpret_cast<f32x4*>(t)[i] = o;
      },
      [=] __device__(long i) { t[i] = gdn_t1(g[i], x[i], norm[i], inverse); });
}

LIC_EXPORT int lic_quantize(const float* v, const float* u, float* out, int64_t n, int32_t training,
                            lic_stream_t stream) {
  if (!v || !out || n < 0 || (training && !u)) return LIC_ERR_INVALID;
  if (training) return ew_launch(n, stream, [=] __device__(long i) { out[i] = v[i] + (u[i] - 0.5f); });
  return ew_launch(n, stream, [=] __device__(long i) { out[i] = rintf(v[i]); });
}

// ---------------------------------------------------------------------------------------------
// entropy-parameter activations (ParametersModels.py:43-64), NHWC rows of G*K*M channels
// ---------------------------------------------------------------------------------------------
#define LIC_MAXK 8
LIC_EXPORT int lic_entropy_params_fwd(const float* raw, float* out, int64_t P, int32_t M, int32_t K,
                                      lic_stream_t stream) {
  if (!raw || !out || P <= 0 || M <= 0 || K < 1 || K > LIC_MAXK) return LIC_ERR_INVALID;
  const long CH = (long)(K == 1 ? 2 : 3) * K * M;
  const long T = (long)K * M;
  return ew_launch(P * M, stream, [=] __device__(long i) {
    const long pix = i / M;
    const int m = (int)(i - pix * M);
    const float* r = raw + pix * CH;
    float* o = out + pix * CH;
    if (K == 1) {
      o[m] = r[m];
      o[M + m] = lic_softplus(r[M + m]) + 1e-6f;
    } else {
      float mx = -INFINITY;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, r[k * M + m]);
      float den = 0.0f;
      for (int k = 0; k < K; ++k) den += expf(r[k * M + m] - mx);
      for (int k = 0; k < K; ++k) {
        o[k * M + m] = expf(r[k * M + m] - mx) / den;
        o[T + k * M + m] = r[T + k * M + m];
        o[2 * T + k * M + m] = lic_softplus(r[2 * T + k * M + m]) + 1e-6f;
      }
    }
  });
}
LIC_EXPORT int lic_entropy_params_bwd(const float* raw, const float* out, const float* dout, float* draw,
                                      int64_t P, int32_t M, int32_t K, lic_stream_t stream) {
  if (!raw || !out || !dout || !draw || P <= 0 || M <= 0 || K < 1 || K > LIC_MAXK) return LIC_ERR_INVALID;
  const long CH = (long)(K == 1 ? 2 : 3) * K * M;
  const long T = (long)K * M;
  return ew_launch(P * M, stream, [=] __device__(long i) {
    const long pix = i / M;
    const int m = (int)(i - pix * M);
    const float* r = raw + pix * CH;
    const float* o = out + pix * CH;
    const float* g = dout + pix * CH;
    float* d = draw + pix * CH;
    if (K == 1) {
      d[m] = g[m];
      d[M + m] = g[M + m] * lic_softplus_grad(r[M + m]);
    } else {
      float dot = 0.0f;
      for (int k = 0; k < K; ++k) dot += g[k * M + m] * o[k * M + m];
      for (int k = 0; k < K; ++k) {
        d[k * M + m] = o[k * M + m] * (g[k * M + m] - dot);
        d[T + k * M + m] = g[T + k * M + m];
        d[2 * T + k * M + m] = g[2 * T + k * M + m] * lic_softplus_grad(r[2 * T + k * M + m]);
      }
    }
  });
}

// ---------------------------------------------------------------------------------------------
// Gaussian / mixture likelihood (EntropyModels.py:188-233; utils.py:6-8) -- same formula as the
// reference: 0.5*(1+erf(t/sqrt2)) differences, no erfc reformulation.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gauss_cdf(float t) { return 0.5f * (1.0f + erff(t / 1.41421356237309515f)); }
__device__ __forceinline__ float gauss_pdf(float t) { return 0.398942280401432678f * expf(-0.5f * t * t); }

LIC_EXPORT int lic_gmm_likelihood_fwd(const float* x, const float* params, float* p, float* logp, int64_t P,
                                      int32_t M, int32_t K, float bound, lic_stream_t stream) {
  if (!x || !params || !p || !logp || P <= 0 || M <= 0 || K < 1 || K > LIC_MAXK) return LIC_ERR_INVALID;
  const long CH = (long)(K == 1 ? 2 : 3) * K * M;
  const long T = (long)K * M;
  return ew_launch(P * M, stream, [=] __device__(long i) {
    const long pix = i / M;
    const int m = (int)(i - pix * M);
    const float* q = params + pix * CH;
    const float xv = x[i];
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float mu = (K == 1) ? q[m] : q[T + k * M + m];
      const float sg = (K == 1) ? q[M + m] : q[2 * T + k * M + m];
      const float up = (xv + 0.5f - mu) / sg, lo = (xv - 0.5f - mu) / sg;
      const float mass = gauss_cdf(up) - gauss_cdf(lo);
      acc = (K == 1) ? mass : acc + q[k * M + m] * mass;
    }
    const float pc = acc > bound ? acc : bound;
    p[i] = pc;
    logp[i] = logf(pc);
  });
}
LIC_EXPORT int lic_gmm_likelihood_bwd(const float* x, const float* params, const float* dp,
                                      const float* dlogp, float* dx, float* dparams, int64_t P, int32_t M,
                                      int32_t K, float bound, lic_stream_t stream) {
  if (!x || !params || !dx || !dparams || P <= 0 || M <= 0 || K < 1 || K > LIC_MAXK) return LIC_ERR_INVALID;
  const long CH = (long)(K == 1 ? 2 : 3) * K * M;
  const long T = (long)K * M;
  return ew_launch(P * M, stream, [=] __device__(long i) {
    const long pix = i / M;
    const int m = (int)(i - pix * M);
    const float* q = params + pix * CH;
    float* dq = dparams + pix * CH;
    const float xv = x[i];
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float mu = (K == 1) ? q[m] : q[T + k * M + m];
      const float sg = (K == 1) ? q[M + m] : q[2 * T + k * M + m];
      const float up = (xv + 0.5f - mu) / sg, lo = (xv - 0.5f - mu) / sg;
      const float mass = gauss_cdf(up) - gauss_cdf(lo);
      acc = (K == 1) ? mass : acc + q[k * M + m] * mass;
    }
    const float pc = acc > bound ? acc : bound;
    float g = 0.0f;
    if (dp) g += dp[i];
    if (dlogp) g += dlogp[i] / pc;
    if (!(acc >= bound)) g = 0.0f;  // clamp_min backward: passes where input >= min
    float dxa = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float mu = (K == 1) ? q[m] : q[T + k * M + m];
      const float sg = (K == 1) ? q[M + m] : q[2 * T + k * M + m];
      const float up = (xv + 0.5f - mu) / sg, lo = (xv - 0.5f - mu) / sg;
      const float pu = gauss_pdf(up), pl = gauss_pdf(lo);
      const float wk = (K == 1) ? 1.0f : q[k * M + m];
      const float gm = g * wk;
      if (K == 1) {
        dq[m] = -gm * (pu - pl) / sg;
        dq[M + m] = -gm * (pu * up - pl * lo) / sg;
      } else {
        dq[k * M + m] = g * (gauss_cdf(up) - gauss_cdf(lo));
        dq[T + k * M + m] = -gm * (pu - pl) / sg;
        dq[2 * T + k * M + m] = -gm * (pu * up - pl * lo) / sg;
      }
      dxa += gm * (pu - pl) / sg;
    }
    dx[i] = dxa;
  });
}

// ---------------------------------------------------------------------------------------------
// factorised bottleneck (EntropyModels.py:49-151): one workgroup per channel; the 43 raw
// parameters are transformed once into LDS (softplus on matrices, tanh on factors).
// ---------------------------------------------------------------------------------------------
__constant__ int FE_MOFF[4] = {0, 3, 12, 21};
__constant__ int FE_BOFF[4] = {24, 27, 30, 33};
__constant__ int FE_FOFF[3] = {34, 37, 40};
__constant__ int FE_DIN[4] = {1, 3, 3, 3};
__constant__ int FE_DOUT[4] = {3, 3, 3, 1};

struct FeTrace {
  float h[4][3];
  float pre[4][3];
};
// T: transformed params (softplus(M), b, tanh(f))
__device__ __forceinline__ float fe_logits(const float* T, float v, FeTrace* tr) {
  float h[3] = {v, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int din = (i == 0) ? 1 : 3, dout = (i == 3) ? 1 : 3;
    const int mo = (i == 0) ? 0 : (i == 1) ? 3 : (i == 2) ? 12 : 21;
    const int bo = 24 + 3 * i;
    float pre[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int o = 0; o < 3; ++o)
      if (o < dout) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < din) acc += T[mo + o * din + k] * h[k];
        pre[o] = acc + T[bo + o];
      }
    if (tr) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        tr->h[i][k] = h[k];
        tr->pre[i][k] = pre[k];
      }
    }
    if (i < 3) {
#pragma unroll
      for (int o = 0; o < 3; ++o) h[o] = pre[o] + T[34 + 3 * i + o] * tanhf(pre[o]);
    } else {
      h[0] = pre[0];
    }
  }
  return h[0];
}
// accumulates d/d(raw params) into dP[43]; R: raw params (for softplus'/tanh' factors)
__device__ __forceinline__ float fe_logits_bwd(const float* T, const float* R, const FeTrace& tr,
                                               float dlogit, float* dP) {
  float dh[3] = {dlogit, 0.0f, 0.0f};
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    const int din = (i == 0) ? 1 : 3, dout = (i == 3) ? 1 : 3;
    const int mo = (i == 0) ? 0 : (i == 1) ? 3 : (i == 2) ? 12 : 21;
    const int bo = 24 + 3 * i;
    float dpre[3] = {0.0f, 0.0f, 0.0f};
    if (i < 3) {
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        const float tf = T[34 + 3 * i + o];
        const float tp = tanhf(tr.pre[i][o]);
        dpre[o] = dh[o] * (1.0f + tf * (1.0f - tp * tp));
        dP[34 + 3 * i + o] += dh[o] * tp * (1.0f - tf * tf);
      }
    } else {
      dpre[0] = dh[0];
    }
    float dhin[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int o = 0; o < 3; ++o)
      if (o < dout) {
        dP[bo + o] += dpre[o];
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < din) {
            dP[mo + o * din + k] += dpre[o] * tr.h[i][k] * lic_softplus_grad(R[mo + o * din + k]);
            dhin[k] += dpre[o] * T[mo + o * din + k];
          }
      }
#pragma unroll
    for (int k = 0; k < 3; ++k) dh[k] = dhin[k];
  }
  return dh[0];
}
__device__ __forceinline__ void fe_load(const float* raw, float* sR, float* sT) {
  if (threadIdx.x < LIC_FE_NPARAM) {
    const float v = raw[threadIdx.x];
    sR[threadIdx.x] = v;
    sT[threadIdx.x] = threadIdx.x < 24 ? lic_softplus(v) : (threadIdx.x < 34 ? v : tanhf(v));
  }
  __syncthreads();
}
__device__ __forceinline__ float signf_(float v) { return (float)((v > 0.0f) - (v < 0.0f)); }

__global__ __launch_bounds__(256) void factorized_fwd_kernel(const float* x, const float* params, float* p,
                                                             float* logp, long P, int C, float bound) {
  __shared__ float sR[LIC_FE_NPARAM], sT[LIC_FE_NPARAM];
  const int c = blockIdx.x;
  fe_load(params + (long)c * LIC_FE_NPARAM, sR, sT);
  for (long pix = (long)blockIdx.y * 256 + threadIdx.x; pix < P; pix += (long)gridDim.y * 256) {
    const long i = pix * C + c;
    const float lo = fe_logits(sT, x[i] - 0.5f, nullptr);
    const float up = fe_logits(sT, x[i] + 0.5f, nullptr);
    const float s = -signf_(lo + up);
    const float pr = fabsf(lic_sigmoid(s * up) - lic_sigmoid(s * lo));
    const float pc = pr > bound ? pr : bound;
    p[i] = pc;
    logp[i] = logf(pc);
  }
}
// grid (C, 1): one workgroup reduces the channel's 43 parameter gradients deterministically
__global__ __launch_bounds__(256) void factorized_bwd_kernel(const float* x, const float* params,
                                                             const float* dp, const float* dlogp, float* dx,
                                                             float* dparams, long P, int C, float bound) {
  __shared__ float sR[LIC_FE_NPARAM], sT[LIC_FE_NPARAM];
  __shared__ float red[4][LIC_FE_NPARAM];
  const int c = blockIdx.x;
  fe_load(params + (long)c * LIC_FE_NPARAM, sR, sT);
  float dP[LIC_FE_NPARAM];
#pragma unroll
  for (int k = 0; k < LIC_FE_NPARAM; ++k) dP[k] = 0.0f;
  for (long pix = threadIdx.x; pix < P; pix += 256) {
    const long i = pix * C + c;
    FeTrace trl, tru;
    const float lo = fe_logits(sT, x[i] - 0.5f, &trl);
    const float up = fe_logits(sT, x[i] + 0.5f, &tru);
    const float s = -signf_(lo + up);
    const float su = lic_sigmoid(s * up), sl = lic_sigmoid(s * lo);
    const float diff = su - sl;
    const float pr = fabsf(diff);
    const float pc = pr > bound ? pr : bound;
    float g = 0.0f;
    if (dp) g += dp[i];
    if (dlogp) g += dlogp[i] / pc;
    if (!(pr >= bound)) g = 0.0f;
    const float gd = g * signf_(diff);
    const float dup = gd * su * (1.0f - su) * s;
    const float dlo = -gd * sl * (1.0f - sl) * s;
    float dxa = fe_logits_bwd(sT, sR, tru, dup, dP);
    dxa += fe_logits_bwd(sT, sR, trl, dlo, dP);
    dx[i] = dxa;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < LIC_FE_NPARAM; ++k) {
    const float v = wave_sum(dP[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < LIC_FE_NPARAM)
    dparams[(long)c * LIC_FE_NPARAM + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void factorized_logits_kernel(const float* params, int ch, const float* xs,
                                                                float* out, long n) {
  __shared__ float sR[LIC_FE_NPARAM], sT[LIC_FE_NPARAM];
  fe_load(params + (long)ch * LIC_FE_NPARAM, sR, sT);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = fe_logits(sT, xs[i], nullptr);
}
LIC_EXPORT int lic_factorized_fwd(const float* x, const float* fe_params, float* p, float* logp, int64_t P,
                                  int32_t C, float bound, lic_stream_t stream) {
  if (!x || !fe_params || !p || !logp || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  int gy = (int)cdiv64(P, 256);
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(factorized_fwd_kernel, dim3(C, gy), dim3(256), 0, (hipStream_t)stream, x, fe_params, p,
                     logp, (long)P, C, bound);
  return lic_check_launch();
}
LIC_EXPORT int lic_factorized_bwd(const float* x, const float* fe_params, const float* dp,
                                  const float* dlogp, float* dx, float* dfe_params, int64_t P, int32_t C,
                                  float bound, lic_stream_t stream) {
  if (!x || !fe_params || !dx || !dfe_params || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  hipLaunchKernelGGL(factorized_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, fe_params, dp,
                     dlogp, dx, dfe_params, (long)P, C, bound);
  return lic_check_launch();
}
// The 11 parameter tensors of the factorised model (EntropyModels.py:62-86: matrices [C,3,1] [C,3,3] [C,3,3] [C,1,3],
// biases [C,3,1] x3 [C,1,1], factors [C,3,1] x3; each contiguous) <-> the [C][43] operand of the kernels above, in
// ONE launch each way.  The host side used torch.cat for the forward and left autograd to copy the 11 strided column
// slices of the [C][43] gradient one by one (12 launches per step for 5,504 floats).  `flat` is the parameter-major
// gradient buffer: parameter k's [C][n_k] block at C * prefix_k, i.e. each block has its parameter's own layout.
struct FePtrs {
  const float* p[11];
};
__constant__ int c_fe_n[11] = {3, 9, 9, 3, 3, 3, 3, 1, 3, 3, 3};
__constant__ int c_fe_pre[12] = {0, 3, 12, 21, 24, 27, 30, 33, 34, 37, 40, 43};
__global__ __launch_bounds__(256) void fe_pack_kernel(FePtrs ptrs, float* packed, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * 43) return;
  const int c = i / 43, j = i - c * 43;
  int k = 0;
#pragma unroll
  for (int q = 1; q < 11; ++q) k += (j >= c_fe_pre[q]) ? 1 : 0;
  packed[i] = ptrs.p[k][c * c_fe_n[k] + (j - c_fe_pre[k])];
}
__global__ __launch_bounds__(256) void fe_unpack_kernel(const float* dpk, float* flat, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * 43) return;
  const int c = i / 43, j = i - c * 43;
  int k = 0;
#pragma unroll
  for (int q = 1; q < 11; ++q) k += (j >= c_fe_pre[q]) ? 1 : 0;
  flat[(long)C * c_fe_pre[k] + c * c_fe_n[k] + (j - c_fe_pre[k])] = dpk[i];
}
LIC_EXPORT int lic_fe_pack(const void* const* params11_host, float* packed, int32_t C, lic_stream_t stream) {
  if (!params11_host || !packed || C <= 0) return LIC_ERR_INVALID;
  FePtrs ptrs;
  for (int k = 0; k < 11; ++k) {
    if (!params11_host[k]) return LIC_ERR_INVALID;
    ptrs.p[k] = (const float*)params11_host[k];
  }
  hipLaunchKernelGGL(fe_pack_kernel, dim3((C * 43 + 255) / 256), dim3(256), 0, (hipStream_t)stream, ptrs, packed, C);
  return lic_check_launch();
}
LIC_EXPORT int lic_fe_unpack(const float* dpacked, float* flat, int32_t C, lic_stream_t stream) {
  if (!dpacked || !flat || C <= 0) return LIC_ERR_INVALID;
  hipLaunchKernelGGL(fe_unpack_kernel, dim3((C * 43 + 255) / 256), dim3(256), 0, (hipStream_t)stream, dpacked, flat, C);
  return lic_check_launch();
}
LIC_EXPORT int lic_factorized_channel_logits(const float* fe_params, int32_t ch, const float* xs, float* out,
                                             int64_t n, lic_stream_t stream) {
  if (!fe_params || !xs || !out || n <= 0 || ch < 0) return LIC_ERR_INVALID;
  hipLaunchKernelGGL(factorized_logits_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     fe_params, ch, xs, out, (long)n);
  return lic_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Entropy-coder tables (the reference has no coder, D3): 16-bit cumulative frequency
// tables from the SAME distributions the likelihood kernels evaluate, for the host range coder
// (lic_codec.h).  A table has S symbols (index i <-> integer value lo + i) and S+1 entries:
//   cum[i] = floor(F_i * (65536 - S)) + i,   F_0 = 0, F_S = 1, F_i = CDF(lo + i - 0.5) otherwise,
// so cum[0] = 0, cum[S] = 65536, every symbol keeps frequency >= 1, index 0 carries the whole lower
// tail and index S-1 the whole upper tail (the coder escapes out-of-window values through them).
// F is forced non-decreasing (fp32 CDFs can step back by an ulp).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned cdf_quant(float F, int S) {
  const float f = fminf(fmaxf(F, 0.0f), 1.0f);
  unsigned c = (unsigned)floorf(f * (float)(65536 - S));
  const unsigned cap = (unsigned)(65536 - S);
  return c > cap ? cap : c;
}
// one workgroup per channel: F_i = sigmoid(L_c(lo + i - 0.5))  (EntropyModels.py:171-174)
__global__ __launch_bounds__(256) void factorized_cdf_tables_kernel(const float* params, int lo, int S,
                                                                    unsigned* out) {
  __shared__ float sR[LIC_FE_NPARAM], sT[LIC_FE_NPARAM];
  extern __shared__ unsigned sq[];
  const int c = blockIdx.x;
  fe_load(params + (long)c * LIC_FE_NPARAM, sR, sT);
  for (int i = threadIdx.x; i <= S; i += 256) {
    float F = i == 0 ? 0.0f : 1.0f;
    if (i > 0 && i < S) F = lic_sigmoid(fe_logits(sT, (float)(lo + i) - 0.5f, nullptr));
    sq[i] = cdf_quant(F, S);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int i = 0; i <= S; ++i) {
      run = sq[i] > run ? sq[i] : run;
      out[(long)c * (S + 1) + i] = (i == S ? (unsigned)(65536 - S) : run) + (unsigned)i;
    }
  }
}
LIC_EXPORT int lic_factorized_cdf_tables(const float* fe_params, int32_t C, int32_t lo, int32_t S, uint32_t* out,
                                         lic_stream_t stream) {
  if (!fe_params || !out || C <= 0 || S < 2 || S > 4096) return LIC_ERR_INVALID;
  hipLaunchKernelGGL(factorized_cdf_tables_kernel, dim3(C), dim3(256), (S + 1) * sizeof(unsigned),
                     (hipStream_t)stream, fe_params, lo, S, out);
  return lic_check_launch();
}
// centre = rint(sum_k w_k mu_k), window [centre - W, centre + W],
// F(x) = sum_k w_k Phi((x - mu_k) / sigma_k)  (EntropyModels.py:192-233, utils.py:6-8)
struct GmmElem {
  float wk[LIC_MAXK], mu[LIC_MAXK], sg[LIC_MAXK];
  int c;
};
__device__ __forceinline__ GmmElem gmm_elem(const float* params, long e, int M, int K, long CH, long T) {
  const long pix = e / M;
  const int m = (int)(e - pix * M);
  const float* q = params + pix * CH;
  GmmElem g;
  float mean = 0.0f;
  for (int k = 0; k < K; ++k) {
    g.wk[k] = (K == 1) ? 1.0f : q[k * M + m];
    g.mu[k] = (K == 1) ? q[m] : q[T + k * M + m];
    g.sg[k] = (K == 1) ? q[M + m] : q[2 * T + k * M + m];
    mean = __builtin_fmaf(g.wk[k], g.mu[k], mean);
  }
  g.c = (int)rintf(mean);
  return g;
}
// quantised CDF value of window entry i (the one expression both kernels below evaluate: encoder and decoder
// may take different kernels and must still build the same table)
__device__ __forceinline__ unsigned gmm_entry(const GmmElem& g, int K, int W, int S, int i) {
  const float x = (float)(g.c - W + i) - 0.5f;
  float F = 0.0f;
  for (int k = 0; k < K; ++k) F = __builtin_fmaf(g.wk[k], gauss_cdf((x - g.mu[k]) / g.sg[k]), F);
  return cdf_quant(F, S);
}
// few elements (the serial decoder: one latent pixel at a time): one WAVE per element, the window entries
// across the lanes, the running maximum as a wave prefix scan -- the thread-per-element loop over 2W+1 entries
// of K error functions each took 49 us for one pixel's 192 elements
__global__ __launch_bounds__(256) void gmm_cdf_tables_wave_kernel(const float* params, long n, int M, int K, int W,
                                                                  int* center, unsigned* out, long CH, long T) {
  const int lane = threadIdx.x & 63;
  const long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n) return;
  const int S = 2 * W + 1;
  const GmmElem g = gmm_elem(params, e, M, K, CH, T);
  unsigned* row = out + e * (long)(S + 1);
  if (lane == 0) {
    center[e] = g.c;
    row[0] = 0;
    row[S] = 65536u;
  }
  unsigned carry = 0;
  for (int i0 = 1; i0 < S; i0 += 64) {
    const int i = i0 + lane;
    unsigned v = i < S ? gmm_entry(g, K, W, S, i) : 0u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {  // inclusive prefix maximum over the lanes
      const unsigned t = (unsigned)__shfl_up((int)v, o, 64);
      if (lane >= o) v = t > v ? t : v;
    }
    v = v > carry ? v : carry;
    if (i < S) row[i] = v + (unsigned)i;
    carry = (unsigned)__shfl((int)v, 63, 64);
  }
}
LIC_EXPORT int lic_gmm_cdf_tables(const float* params, int64_t P, int32_t M, int32_t K, int32_t W,
                                  int32_t* center, uint32_t* out, lic_stream_t stream) {
  if (!params || !center || !out || P <= 0 || M <= 0 || K < 1 || K > LIC_MAXK || W < 1 || W > 2047)
    return LIC_ERR_INVALID;
  const long CH = (long)(K == 1 ? 2 : 3) * K * M;
  const long T = (long)K * M;
  const int S = 2 * W + 1;
  if (P * M <= 32768 && getenv("LIC_TABLES_NO_WAVE") == nullptr) {
    hipLaunchKernelGGL(gmm_cdf_tables_wave_kernel, dim3((unsigned)cdiv64(P * M, 4)), dim3(256), 0, (hipStream_t)stream,
                       params, (long)(P * M), M, K, W, center, out, CH, T);
    return lic_check_launch();
  }
  return ew_launch(P * M, stream, [=] __device__(long e) {
    const GmmElem g = gmm_elem(params, e, M, K, CH, T);
    center[e] = g.c;
    unsigned* row = out + e * (long)(S + 1);
    unsigned run = 0;
    row[0] = 0;
    for (int i = 1; i < S; ++i) {
      const unsigned v = gmm_entry(g, K, W, S, i);
      run = v > run ? v : run;
      row[i] = run + (unsigned)i;
    }
    row[S] = 65536u;
  });
}

// ---------------------------------------------------------------------------------------------
// rate-distortion loss (RateDistortionLoss.py:5-49)
// ---------------------------------------------------------------------------------------------
#define RD_CHUNKS 64
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// grid (RD_CHUNKS, B): partial[b][chunk][3] = sum logp_y, sum logp_z, sum (x_hat-x)^2
__global__ __launch_bounds__(256) void rd_partial_kernel(const float* logp_y, long ny, const float* logp_z,
                                                         long nz, const float* x_hat, const float* x,
                                                         long nx, double* partial) {
  __shared__ double sh[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const long stride = (long)RD_CHUNKS * 256;
  const long t0 = (long)ch * 256 + threadIdx.x;
  float ay = 0.0f, az = 0.0f, ae = 0.0f;
  const float* py = logp_y + (long)b * ny;
  for (long i = t0; i < ny; i += stride) ay += py[i];
  const float* pz = logp_z + (long)b * nz;
  for (long i = t0; i < nz; i += stride) az += pz[i];
  const float* ph = x_hat + (long)b * nx;
  const float* px = x + (long)b * nx;
  if ((nx & 3) == 0 && ((reinterpret_cast<uintptr_t>(ph) | reinterpret_cast<uintptr_t>(px)) & 15) == 0) {
    for (long i = t0; i < nx / 4; i += stride) {
      const f32x4 a = reinterpret_cast<const f32x4*>(ph)[i], c = reinterpret_cast<const f32x4*>(px)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = a[e] - c[e];
        ae += d * d;
      }
    }
  } else {
    for (long i = t0; i < nx; i += stride) {
      const float d = ph[i] - px[i];
      ae += d * d;
    }
  }
  const double sy = block_sum_d((double)ay, sh);
  const double sz = block_sum_d((double)az, sh);
  const double se = block_sum_d((double)ae, sh);
  if (threadIdx.x == 0) {
    double* o = partial + ((long)b * RD_CHUNKS + ch) * 3;
    o[0] = sy;
    o[1] = sz;
    o[2] = se;
  }
}
__global__ __launch_bounds__(64) void rd_final_kernel(const double* partial, int B, long nx, long num_pixels,
                                                      float lambda_rd, float* out) {
  // one wave; lane b handles image b (looping when B > 64)
  const float ln2 = 0.693147180559945309f;
  double sby = 0, sbz = 0, sbpy = 0, sbpz = 0, smse = 0;
  for (int b = threadIdx.x; b < B; b += 64) {
    double sy = 0, sz = 0, se = 0;
    for (int c = 0; c < RD_CHUNKS; ++c) {
      const double* o = partial + ((long)b * RD_CHUNKS + c) * 3;
      sy += o[0];
      sz += o[1];
      se += o[2];
    }
    const float bits_y = (float)(-sy) / ln2, bits_z = (float)(-sz) / ln2;
    const float mse_b = (float)(se / (double)nx);
    sby += bits_y;
    sbz += bits_z;
    sbpy += bits_y / (float)num_pixels;
    sbpz += bits_z / (float)num_pixels;
    smse += mse_b;
    out[16 + b] = mse_b;
    out[16 + B + b] = -10.0f * log10f(mse_b + 1e-8f);
  }
  sby = wave_sum_d(sby);
  sbz = wave_sum_d(sbz);
  sbpy = wave_sum_d(sbpy);
  sbpz = wave_sum_d(sbpz);
  smse = wave_sum_d(smse);
  if (threadIdx.x == 0) {
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
#pragma unroll
    for (int k = 9; k < 16; ++k) out[k] = 0.0f;   // (unused slots: the caller need not clear the buffer first)
  }
}
LIC_EXPORT size_t lic_rd_loss_workspace_bytes(int32_t B) {
  return B > 0 ? (size_t)B * RD_CHUNKS * 3 * sizeof(double) : 0;
}
LIC_EXPORT int lic_rd_loss_fwd(const float* logp_y, int64_t ny, const float* logp_z, int64_t nz,
                               const float* x_hat, const float* x, int64_t nx, int32_t B, int64_t num_pixels,
                               float lambda_rd, float* out, void* workspace, size_t workspace_bytes,
                               lic_stream_t stream) {
  if (!logp_y || !logp_z || !x_hat || !x || !out || !workspace || B <= 0 || ny <= 0 || nz <= 0 || nx <= 0 ||
      num_pixels <= 0)
    return LIC_ERR_INVALID;
  if (workspace_bytes < lic_rd_loss_workspace_bytes(B)) return LIC_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rd_partial_kernel, dim3(RD_CHUNKS, B), dim3(256), 0, s, logp_y, (long)ny, logp_z, (long)nz,
                     x_hat, x, (long)nx, (double*)workspace);
  int rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  hipLaunchKernelGGL(rd_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, B, (long)nx,
                     (long)num_pixels, lambda_rd, out);
  return lic_check_launch();
}
LIC_EXPORT int lic_rd_loss_bwd(const float* x_hat, const float* x, int64_t ny, int64_t nz, int64_t nx,
                               int32_t B, int64_t num_pixels, float lambda_rd, const float* gl, float* dlogp_y,
                               float* dlogp_z, float* dx_hat, lic_stream_t stream) {
  if (!x_hat || !x || !gl || !dlogp_y || !dlogp_z || !dx_hat || B <= 0) return LIC_ERR_INVALID;
  const float ln2 = 0.693147180559945309f;
  const float ky = -1.0f / (ln2 * (float)num_pixels * (float)B);
  const float kx = lambda_rd * (255.0f * 255.0f) * 2.0f / ((float)nx * (float)B);
  const long n = nx * B;
  if (al16(x_hat) && al16(x) && al16(dx_hat) && n % 4 == 0 && ny >= 0 && nz >= 0) {
    // one launch for the three gradients (they were three: the two constant fills are 1 M and 65 K elements behind a
    // launch each, in the one segment of a step where both streams wait for the loss): index ranges [0, n/4) image
    // gradient, 16 bytes per lane; then dlogp_y; then dlogp_z -- the same expressions, bit for bit
    const long n4 = n / 4, e1 = n4 + ny * B, e2 = e1 + nz * B;
    return ew_launch(e2, stream, [=] __device__(long i) {
      if (i < n4) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x_hat)[i], c = reinterpret_cast<const f32x4*>(x)[i];
        const float k = gl[0] * kx;
        reinterpret_cast<f32x4*>(dx_hat)[i] = (a - c) * k;
      } else if (i < e1) {
        dlogp_y[i - n4] = gl[0] * ky;
      } else {
        dlogp_z[i - e1] = gl[0] * ky;
      }
    });
  }
  int rc = ew_launch(ny * B, stream, [=] __device__(long i) { dlogp_y[i] = gl[0] * ky; });
  if (rc != LIC_OK) return rc;
  rc = ew_launch(nz * B, stream, [=] __device__(long i) { dlogp_z[i] = gl[0] * ky; });
  if (rc != LIC_OK) return rc;
  return ew_launch_vec(
      n, al16(x_hat) && al16(x) && al16(dx_hat), stream,
      [=] __device__(long i) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x_hat)[i], c = reinterpret_cast<const f32x4*>(x)[i];
        const float k = gl[0] * kx;
        reinterpret_cast<f32x4*>(dx_hat)[i] = (a - c) * k;
      },
      [=] __device__(long i) { dx_hat[i] = gl[0] * kx * (x_hat[i] - x[i]); });
}
