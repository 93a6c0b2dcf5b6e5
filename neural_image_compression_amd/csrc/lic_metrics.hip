// Evaluation metrics on the device: multi-scale SSIM as the reference's evaluator calls it
// (Evaluator.py:7,38,45: `ms_ssim(recon, orig, data_range=1.0, size_average=True)` from the
// third-party pytorch-msssim==0.2.1 -- absent offline, "parity unpinned": this follows the
// package's published algorithm: 11-tap Gaussian window (sigma 1.5), separable VALID filtering of
// X, Y, X^2, Y^2, XY per channel, K = (0.01, 0.03), 5 scales with 2x2 average pooling (odd sides
// zero-padded by one, divisor 4), weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), relu on the
// per-scale terms, product of powers, mean over channels).
//
// HBM-bound (a Kodak frame is 4.7 MB): one kernel per scale reads each input pixel once into an
// LDS tile, filters rows then columns out of LDS, and reduces the ssim / cs maps to per-block
// partial sums (fp64, fixed order => bitwise reproducible); a second tiny kernel finishes.
#include "lic_common.h"

namespace {

constexpr int MS_WIN = 11, MS_HALO = MS_WIN - 1;
constexpr int MS_T = 32;                  // output tile side
constexpr int MS_IN = MS_T + MS_HALO;     // input tile side (42)
constexpr int MS_LEVELS = 5;

struct MsWin {
  float g[MS_WIN];
};

struct Plane {  // element (b, c, h, w) at b*sb + c*sc + h*sh + w*sw
  const float* p;
  long sb, sc, sh, sw;
};

// one scale: partial[(bc * nblk + blk) * 2 + {0,1}] = sum over the block's valid outputs of
// ssim_map / cs_map
__global__ __launch_bounds__(256) void ssim_scale_kernel(Plane X, Plane Y, int C, int H, int W, int Ho, int Wo,
                                                         MsWin win, float C1, float C2, double* partial) {
  __shared__ float sx[MS_IN][MS_IN + 1], sy[MS_IN][MS_IN + 1];
  __shared__ float hr[5][MS_IN][MS_T + 1];  // row-filtered X, Y, XX, YY, XY
  __shared__ double red[2][4];
  const int bc = blockIdx.z, b = bc / C, c = bc - b * C;
  const int ox0 = blockIdx.x * MS_T, oy0 = blockIdx.y * MS_T;
  const float* xp = X.p + b * X.sb + c * X.sc;
  const float* yp = Y.p + b * Y.sb + c * Y.sc;
  for (int i = threadIdx.x; i < MS_IN * MS_IN; i += 256) {
    const int r = i / MS_IN, q = i - r * MS_IN;
    const int h = oy0 + r, w = ox0 + q;
    float vx = 0.0f, vy = 0.0f;
    if (h < H && w < W) {
      vx = xp[h * X.sh + w * X.sw];
      vy = yp[h * Y.sh + w * Y.sw];
    }
    sx[r][q] = vx;
    sy[r][q] = vy;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < MS_IN * MS_T; i += 256) {
    const int r = i / MS_T, q = i - r * MS_T;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
    for (int k = 0; k < MS_WIN; ++k) {
      const float g = win.g[k], vx = sx[r][q + k], vy = sy[r][q + k];
      a0 += g * vx;
      a1 += g * vy;
      a2 += g * (vx * vx);
      a3 += g * (vy * vy);
      a4 += g * (vx * vy);
    }
    hr[0][r][q] = a0;
    hr[1][r][q] = a1;
    hr[2][r][q] = a2;
    hr[3][r][q] = a3;
    hr[4][r][q] = a4;
  }
  __syncthreads();
  double s_ssim = 0.0, s_cs = 0.0;
  for (int i = threadIdx.x; i < MS_T * MS_T; i += 256) {
    const int r = i / MS_T, q = i - r * MS_T;
    if (oy0 + r >= Ho || ox0 + q >= Wo) continue;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < MS_WIN; ++k) {
      const float g = win.g[k];
      m1 += g * hr[0][r + k][q];
      m2 += g * hr[1][r + k][q];
      e11 += g * hr[2][r + k][q];
      e22 += g * hr[3][r + k][q];
      e12 += g * hr[4][r + k][q];
    }
    const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
    const float s11 = e11 - m11, s22 = e22 - m22, s12 = e12 - m12;
    const float cs = (2.0f * s12 + C2) / (s11 + s22 + C2);
    const float ss = ((2.0f * m12 + C1) / (m11 + m22 + C1)) * cs;
    s_ssim += (double)ss;
    s_cs += (double)cs;
  }
  // deterministic block reduction: lanes by shuffle, the four waves in fixed order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s_ssim += __shfl_down(s_ssim, off, 64);
    s_cs += __shfl_down(s_cs, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s_ssim;
    red[1][threadIdx.x >> 6] = s_cs;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x, nblk = (long)gridDim.x * gridDim.y;
    partial[((long)bc * nblk + blk) * 2 + 0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    partial[((long)bc * nblk + blk) * 2 + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}

// level_out[(level*BC + bc)*2 + {0,1}] = mean ssim, mean cs of this scale
__global__ __launch_bounds__(64) void ssim_finish_kernel(const double* partial, long nblk, double inv_count,
                                                         float* level_out, int level, int BC) {
  const int bc = blockIdx.x;
  double a = 0.0, b = 0.0;
  for (long i = threadIdx.x; i < nblk; i += 64) {
    a += partial[((long)bc * nblk + i) * 2 + 0];
    b += partial[((long)bc * nblk + i) * 2 + 1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  if (threadIdx.x == 0) {
    level_out[((long)level * BC + bc) * 2 + 0] = (float)(a * inv_count);
    level_out[((long)level * BC + bc) * 2 + 1] = (float)(b * inv_count);
  }
}

// 2x2 average pool, stride 2, zero padding `ph`/`pw` (0 or 1) on both sides, divisor always 4;
// planar [BC][Ho][Wo] output
__global__ __launch_bounds__(256) void avgpool2_kernel(Plane X, int C, int H, int W, int ph, int pw, int Ho, int Wo,
                                                       float* out, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int wo = (int)(i % Wo);
    long t = i / Wo;
    const int ho = (int)(t % Ho);
    const int bc = (int)(t / Ho);
    const int b = bc / C, c = bc - b * C;
    const float* xp = X.p + b * X.sb + c * X.sc;
    float acc = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int h = 2 * ho - ph + dy, w = 2 * wo - pw + dx;
        if (h >= 0 && h < H && w >= 0 && w < W) acc += xp[h * X.sh + w * X.sw];
      }
    out[i] = acc * 0.25f;
  }
}

// out[bc] = prod_l relu(term_l)^weight_l, term_l = cs for l < 4, ssim for l = 4
__global__ __launch_bounds__(64) void msssim_combine_kernel(const float* level_out, int BC, float* out) {
  const int bc = blockIdx.x * 64 + threadIdx.x;
  if (bc >= BC) return;
  const float wts[MS_LEVELS] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
  float v = 1.0f;
#pragma unroll
  for (int l = 0; l < MS_LEVELS; ++l) {
    const float t = level_out[((long)l * BC + bc) * 2 + (l == MS_LEVELS - 1 ? 0 : 1)];
    v *= powf(t > 0.0f ? t : 0.0f, wts[l]);
  }
  out[bc] = v;
}

struct MsPlan {
  int H[MS_LEVELS], W[MS_LEVELS];
  size_t plane_off[MS_LEVELS];  // float offset of level l's pooled X plane (l >= 1); Y follows X
  size_t partial_off;           // byte offset of the fp64 partial sums
  size_t bytes;
};

bool ms_plan(int B, int C, int H, int W, MsPlan* pl) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return false;
  const int smaller = H < W ? H : W;
  if (smaller <= (MS_WIN - 1) * 16) return false;  // the package's assertion
  size_t off = 0;
  size_t max_blk = 0;
  int h = H, w = W;
  for (int l = 0; l < MS_LEVELS; ++l) {
    pl->H[l] = h;
    pl->W[l] = w;
    pl->plane_off[l] = off;
    if (l > 0) off += 2 * (size_t)B * C * h * w;
    const size_t nb = (size_t)cdiv64(h - MS_HALO, MS_T) * cdiv64(w - MS_HALO, MS_T);
    if (nb > max_blk) max_blk = nb;
    const int ph = h & 1, pw = w & 1;
    h = (h + 2 * ph - 2) / 2 + 1;
    w = (w + 2 * pw - 2) / 2 + 1;
  }
  off = (off * sizeof(float) + 15) & ~(size_t)15;
  pl->partial_off = off;
  pl->bytes = off + max_blk * (size_t)B * C * 2 * sizeof(double);
  return true;
}

}  // namespace

LIC_EXPORT size_t lic_msssim_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W) {
  MsPlan pl;
  return ms_plan(B, C, H, W, &pl) ? pl.bytes : 0;
}

LIC_EXPORT int lic_msssim(const float* x, const float* y, int32_t B, int32_t C, int32_t H, int32_t W, int64_t sb,
                          int64_t sc, int64_t sh, int64_t sw, float data_range, float* out, float* level_out,
                          void* workspace, size_t workspace_bytes, lic_stream_t stream) {
  if (!x || !y || !out || !level_out || !workspace) return LIC_ERR_INVALID;
  MsPlan pl;
  if (!ms_plan(B, C, H, W, &pl)) return LIC_ERR_UNSUPPORTED;  // side <= 160: the package raises too
  if (workspace_bytes < pl.bytes) return LIC_ERR_WORKSPACE;
  if ((long)B * C > 65535) return LIC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  MsWin win;
  {  // g = exp(-(i - 5)^2 / (2 * 1.5^2)), normalised, in fp32 as the package builds it
    float sum = 0.0f;
    for (int i = 0; i < MS_WIN; ++i) {
      const float d = (float)(i - MS_WIN / 2);
      win.g[i] = expf(-(d * d) / (2.0f * 1.5f * 1.5f));
      sum += win.g[i];
    }
    for (int i = 0; i < MS_WIN; ++i) win.g[i] /= sum;
  }
  const float C1 = (0.01f * data_range) * (0.01f * data_range), C2 = (0.03f * data_range) * (0.03f * data_range);
  float* planes = (float*)workspace;
  double* partial = (double*)((char*)workspace + pl.partial_off);
  const int BC = B * C;
  Plane X{x, sb, sc, sh, sw}, Y{y, sb, sc, sh, sw};
  for (int l = 0; l < MS_LEVELS; ++l) {
    const int h = pl.H[l], w = pl.W[l], ho = h - MS_HALO, wo = w - MS_HALO;
    dim3 grid((unsigned)cdiv64(wo, MS_T), (unsigned)cdiv64(ho, MS_T), (unsigned)BC);
    hipLaunchKernelGGL(ssim_scale_kernel, grid, dim3(256), 0, s, X, Y, C, h, w, ho, wo, win, C1, C2, partial);
    int rc = lic_check_launch();
    if (rc != LIC_OK) return rc;
    hipLaunchKernelGGL(ssim_finish_kernel, dim3(BC), dim3(64), 0, s, (const double*)partial,
                       (long)grid.x * grid.y, 1.0 / ((double)ho * wo), level_out, l, BC);
    rc = lic_check_launch();
    if (rc != LIC_OK) return rc;
    if (l + 1 < MS_LEVELS) {
      const int h2 = pl.H[l + 1], w2 = pl.W[l + 1];
      float* nx = planes + pl.plane_off[l + 1];
      float* ny = nx + (size_t)BC * h2 * w2;
      const long total = (long)BC * h2 * w2;
      hipLaunchKernelGGL(avgpool2_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s, X, C, h, w, h & 1, w & 1, h2,
                         w2, nx, total);
      hipLaunchKernelGGL(avgpool2_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s, Y, C, h, w, h & 1, w & 1, h2,
                         w2, ny, total);
      rc = lic_check_launch();
      if (rc != LIC_OK) return rc;
      X = Plane{nx, (long)C * h2 * w2, (long)h2 * w2, (long)w2, 1};
      Y = Plane{ny, (long)C * h2 * w2, (long)h2 * w2, (long)w2, 1};
    }
  }
  hipLaunchKernelGGL(msssim_combine_kernel, dim3((BC + 63) / 64), dim3(64), 0, s, (const float*)level_out, BC, out);
  return lic_check_launch();
}

// ------------------------------------------------------------------------------------------------
// SURVEY 8(f).3 -- input pipeline and logging statistics on the device.
// ------------------------------------------------------------------------------------------------
// uint8 NHWC pixels -> fp32 in [0,1]: out = float(v) / 255 (a true division: bit-identical to
// torchvision's ToTensor(), Dataloader.py:7-9).  n % 4 == 0 takes the 4-pixel path.
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* in, float* out, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const uchar4 v = reinterpret_cast<const uchar4*>(in)[i];
    f32x4 o = {(float)v.x / 255.0f, (float)v.y / 255.0f, (float)v.z / 255.0f, (float)v.w / 255.0f};
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = (float)in[i] / 255.0f;
}
LIC_EXPORT int lic_u8_to_f32(const uint8_t* in, float* out, int64_t n, lic_stream_t stream) {
  if (!in || !out || n < 0) return LIC_ERR_INVALID;
  if (n == 0) return LIC_OK;
  if ((reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 15)) return LIC_ERR_INVALID;
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3(ew_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     (long)n);
  return lic_check_launch();
}

// Summary of a tensor for logging (what Trainer.py:167-217 ships to the host as whole tensors):
// stats[0..4] = count, sum, sum of squares, min, max (fp64) and a fixed-range histogram of `nbins`
// equal bins over [lo, hi] (values outside are clamped into the edge bins; NaNs are skipped and
// counted in stats[5]).  Integer bin counts via atomics (order-independent), fp64 sums via per-block
// partials reduced in fixed order => reproducible.
#define ST_BLOCKS 256
__global__ __launch_bounds__(256) void tensor_stats_kernel(const float* x, long n, int nbins, float lo, float inv_w,
                                                           double* partial, unsigned long long* hist) {
  extern __shared__ unsigned sh_hist[];
  __shared__ double red[6][4];
  for (int i = threadIdx.x; i < nbins; i += 256) sh_hist[i] = 0;
  __syncthreads();
  double s = 0.0, ss = 0.0, cnt = 0.0, nan = 0.0;
  double mn = 1e300, mx = -1e300;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    if (v != v) {
      nan += 1.0;
      continue;
    }
    s += (double)v;
    ss += (double)v * (double)v;
    cnt += 1.0;
    mn = (double)v < mn ? (double)v : mn;
    mx = (double)v > mx ? (double)v : mx;
    int b = (int)floorf((v - lo) * inv_w);
    b = b < 0 ? 0 : (b >= nbins ? nbins - 1 : b);
    atomicAdd(&sh_hist[b], 1u);
  }
  double vals[6] = {cnt, s, ss, mn, mx, nan};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double v = vals[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double o = __shfl_down(v, off, 64);
      v = (k == 3) ? (o < v ? o : v) : (k == 4 ? (o > v ? o : v) : v + o);
    }
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    double v = red[k][0];
    for (int w = 1; w < 4; ++w) v = (k == 3) ? (red[k][w] < v ? red[k][w] : v) : (k == 4 ? (red[k][w] > v ? red[k][w] : v) : v + red[k][w]);
    partial[(long)blockIdx.x * 6 + k] = v;
  }
  for (int i = threadIdx.x; i < nbins; i += 256)
    if (sh_hist[i]) atomicAdd(&hist[i], (unsigned long long)sh_hist[i]);
}
__global__ __launch_bounds__(64) void tensor_stats_finish_kernel(const double* partial, int nblocks, double* stats) {
  const int k = threadIdx.x;
  if (k >= 6) return;
  double v = partial[k];
  for (int b = 1; b < nblocks; ++b) {
    const double o = partial[(long)b * 6 + k];
    v = (k == 3) ? (o < v ? o : v) : (k == 4 ? (o > v ? o : v) : v + o);
  }
  stats[k] = v;
}
LIC_EXPORT size_t lic_tensor_stats_workspace_bytes(void) { return (size_t)ST_BLOCKS * 6 * sizeof(double); }
LIC_EXPORT int lic_tensor_stats(const float* x, int64_t n, int32_t nbins, float lo, float hi, double* stats,
                                uint64_t* hist, void* workspace, size_t workspace_bytes, lic_stream_t stream) {
  if (!x || !stats || !hist || !workspace || n <= 0 || nbins < 1 || nbins > 4096 || !(hi > lo)) return LIC_ERR_INVALID;
  if (workspace_bytes < lic_tensor_stats_workspace_bytes()) return LIC_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int nb = (int)cdiv64(n, 256 * 8);
  if (nb > ST_BLOCKS) nb = ST_BLOCKS;
  if (nb < 1) nb = 1;
  hipError_t e = hipMemsetAsync(hist, 0, (size_t)nbins * sizeof(uint64_t), s);
  if (e != hipSuccess) {
    g_lic_last_hip_error = (int)e;
    return LIC_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(tensor_stats_kernel, dim3(nb), dim3(256), nbins * sizeof(unsigned), s, x, (long)n, nbins, lo,
                     (float)nbins / (hi - lo), (double*)workspace, (unsigned long long*)hist);
  int rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  hipLaunchKernelGGL(tensor_stats_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, nb, stats);
  return lic_check_launch();
}
