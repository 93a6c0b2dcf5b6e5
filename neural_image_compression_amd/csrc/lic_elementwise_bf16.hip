// HBM-bound helpers of the bf16-storage path (BASELINE config 3): patch<->column conversion
// with bf16 columns, column sums and the GDN dL/dnorm map on bf16 tensors.  fp32 arithmetic inside.
#include "lic_common.h"
#include <stdlib.h>
#include "lic_patch.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// col[(b,oh,ow)][(r*kw+s)*C + c] (bf16) = x[b, oh*stride-pad+r, ow*stride-pad+s, c] (fp32), 0 outside / pad
__global__ __launch_bounds__(256) void im2col_bf16_kernel(const float* x, bf16_t* col, int B, int H, int W, int C,
                                                          int Ho, int Wo, int kh, int kw, int stride, int pad,
                                                          int Kpad) {
  const long total = (long)B * Ho * Wo * Kpad;
  const int K = kh * kw * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
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
    col[i] = (bf16_t)v;
  }
}
// The RGB geometry (C = 3, 5x5, stride 2, padding 2, Kpad = 80), one lane per output pixel: the 15 values a pixel takes
// from image row 2 oy - 2 + r are 15 CONSECUTIVE floats of the NHWC image (the stem kernel's observation), so a column
// row is five runs of 15 -- four 16-byte loads each, no lookup table, no per-element index arithmetic -- and leaves the
// lane as ten 16-byte stores; neighbouring lanes write neighbouring 160-byte rows.  (The generic kernel gathers every
// element through a table: 66 us for the 84 MB column matrix of a 256^2 x 32 batch, twice per bf16 step.)
__global__ __launch_bounds__(256) void im2col_rgb5_bf16_kernel(const float* x, bf16_t* col, long npix, int H, int W, int Ho,
                                                               int Wo) {
  typedef float i2_f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef unsigned i2_u32x4 __attribute__((ext_vector_type(4)));
  // a wave's 64 rows are 10 KB of CONSECUTIVE column-matrix bytes: the lanes park their 160-byte rows in LDS (16-byte slot
  // 10 i + q of lane i: the eight lanes of a write group hit eight different slots) and the wave stores the block in linear
  // order, ten fully coalesced 1 KB stores instead of ten stores of 64 16-byte pieces 160 bytes apart
  __shared__ __attribute__((aligned(16))) i2_u32x4 park[4][640];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowf = 3 * W;
  for (long base = (long)blockIdx.x * 256 + wave * 64; base < npix; base += (long)gridDim.x * 256) {
    const long pix = base + lane;
    if (pix < npix) {
      const long t = pix / Wo;
      const int ow = (int)(pix - t * Wo);
      const long b = t / Ho;
      const int oh = (int)(t - b * Ho);
      const float* img = x + b * (long)H * rowf;
      const int cf0 = (2 * ow - 2) * 3;                   // first of the 15 floats within an image row
      const bool colfast = cf0 >= 0 && cf0 + 16 <= rowf;  // (16: the fourth 16-byte load reads one float too many)
      float v[80];
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int ih = 2 * oh - 2 + r;
        const bool rowok = ih >= 0 && ih < H;
        const float* src = img + (long)(rowok ? ih : 0) * rowf;
        if (colfast) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const i2_f32x4u w4 = *reinterpret_cast<const i2_f32x4u*>(src + cf0 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (4 * q + e < 15) v[15 * r + 4 * q + e] = rowok ? w4[e] : 0.0f;
          }
        } else {  // image border: element-wise, zero outside
#pragma unroll
          for (int j = 0; j < 15; ++j) {
            const int c = cf0 + j;
            const bool ok = rowok && c >= 0 && c < rowf;
            const float tv = src[ok ? c : 0];
            v[15 * r + j] = ok ? tv : 0.0f;
          }
        }
      }
#pragma unroll
      for (int j = 75; j < 80; ++j) v[j] = 0.0f;
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        i2_u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          typedef float i2_f32x2 __attribute__((ext_vector_type(2)));
          typedef bf16_t i2_bf16x2 __attribute__((ext_vector_type(2)));
          const i2_f32x2 pr = {v[8 * q + 2 * e], v[8 * q + 2 * e + 1]};
          o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, i2_bf16x2));
        }
        park[wave][10 * lane + q] = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const long rows = npix - base < 64 ? npix - base : 64;   // rows of this block that exist
    i2_u32x4* dst = reinterpret_cast<i2_u32x4*>(col + base * 80);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const int slot = 64 * q + lane;
      if (slot < 10 * rows) dst[slot] = park[wave][slot];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

LIC_EXPORT int lic_im2col_bf16(const float* x, void* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho,
                               int32_t Wo, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t Kpad,
                               lic_stream_t stream) {
  if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || Kpad < kh * kw * C)
    return LIC_ERR_INVALID;
  const long total = (long)B * Ho * Wo * Kpad;
  if (C == 3 && kh == 5 && kw == 5 && stride == 2 && pad == 2 && Kpad == 80 && Ho == (H + 1) / 2 && Wo == (W + 1) / 2 &&
      (reinterpret_cast<uintptr_t>(col) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0 &&
      getenv("LIC_IM2COL_GENERIC") == nullptr) {
    const long npix = (long)B * Ho * Wo;
    hipLaunchKernelGGL(im2col_rgb5_bf16_kernel, dim3(ew_grid(npix, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (bf16_t*)col, npix, H, W, Ho, Wo);
    return lic_check_launch();
  }
  if (Kpad <= LIC_PATCH_MAXK && Kpad % 8 == 0 && total / 8 < 0x7FFFFFFFL &&
      (reinterpret_cast<uintptr_t>(col) & 15) == 0 && kh < 256 && kw < 256 && C < 256 && (long)H * W * C < 0x7FFFFFFFL)
    hipLaunchKernelGGL((im2col_vec_kernel<bf16_t, 8>), dim3(ew_grid(total / 8, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, (bf16_t*)col, (unsigned)((long)B * Ho * Wo), H, W, C, Ho, Wo, kh, kw,
                       stride, pad, Kpad);
  else
    hipLaunchKernelGGL(im2col_bf16_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (bf16_t*)col, B, H, W, C, Ho, Wo, kh, kw, stride, pad, Kpad);
  return lic_check_launch();
}

// out[b,oy,ox,c] (fp32) = bias[c] + sum_{r,s} col[(b,ih,iw)][(r*kw+s)*C + c] (bf16), oy = ih*stride-pad+r
__global__ __launch_bounds__(256) void col2im_bf16_kernel(const bf16_t* col, const float* bias, float* out, int B,
                                                          int Hi, int Wi, int C, int Ho, int Wo, int kh, int kw,
                                                          int stride, int pad, int Kpad) {
  const long total = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
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
        v += (float)col[(((long)b * Hi + ih) * Wi + iw) * Kpad + (r * kw + s) * C + c];
      }
    }
    out[i] = v;
  }
}
LIC_EXPORT int lic_col2im_bf16(const void* col, const float* bias, float* out, int32_t B, int32_t Hi, int32_t Wi,
                               int32_t C, int32_t Ho, int32_t Wo, int32_t kh, int32_t kw, int32_t stride,
                               int32_t pad, int32_t Kpad, lic_stream_t stream) {
  if (!col || !out || B <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || stride < 1 ||
      Kpad < kh * kw * C)
    return LIC_ERR_INVALID;
  const long total = (long)B * Ho * Wo * C;
  if (total < 0x7FFFFFFFL && (long)B * Hi * Wi < 0x7FFFFFFFL)
    hipLaunchKernelGGL((col2im_fast_kernel<bf16_t>), dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)col, bias, out, (unsigned)total, Hi, Wi, C, Ho, Wo, kh, kw, stride, pad, Kpad);
  else
    hipLaunchKernelGGL(col2im_bf16_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)col, bias, out, B, Hi, Wi, C, Ho, Wo, kh, kw, stride, pad, Kpad);
  return lic_check_launch();
}

// column sums of a bf16 [P][ld] matrix -> fp32, two deterministic stages (C % 8 == 0)
// (blockIdx.z = 1: the second matrix of lic_colsum2_bf16 -- same shape, its partials behind the first's)
// MASKED (lic_leaky_bwd_colsum_bf16): `in` = dy goes through the LeakyReLU's backward first -- v = y > 0 ? dy : bf16(dy *
// slope), leaky_bwd_bf16_kernel's arithmetic -- is written to `gout` and summed: the bias gradient of a conv -> LeakyReLU
// layer in the pass that masks its gradient (every element is visited exactly once), instead of a pass of its own.
template <bool MASKED>
__global__ __launch_bounds__(256) void colsum_bf16_stage1(const bf16_t* in, const bf16_t* in2, long ld, long P, int C,
                                                          float* part, int nchunk, const bf16_t* ymask, bf16_t* gout,
                                                          float slope) {
  __shared__ float red[32][64 + 4];
  if (blockIdx.z) {
    in = in2;
    part += (long)nchunk * C;
  }
  const int cg = threadIdx.x & 7, ry = threadIdx.x >> 3;  // 8 column octets x 32 row lanes
  const int c = blockIdx.x * 64 + cg * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C)
    for (long pr = (long)blockIdx.y * 32 + ry; pr < P; pr += (long)nchunk * 32) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(in + pr * ld + c);
      if constexpr (MASKED) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ymask + pr * ld + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)a[e] > 0.0f ? v[e] : (bf16_t)((float)v[e] * slope);
        *reinterpret_cast<bf16x8*>(gout + pr * ld + c) = v;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ry][cg * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
    if (cc < C) part[(long)blockIdx.y * C + cc] = s;
  }
}
__global__ __launch_bounds__(256) void colsum_bf16_stage2(const float* part, int C, int nchunk, float scale,
                                                          float* out, float* out2) {
  __shared__ double red[16][17];
  if (blockIdx.z) {
    out = out2;
    part += (long)nchunk * C;
  }
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
static int csh_chunks(int64_t P) {
  int64_t n = cdiv64(P, 32 * 8);
  if (n > 256) n = 256;
  if (n < 1) n = 1;
  return (int)n;
}
LIC_EXPORT size_t lic_colsum_bf16_workspace_bytes(int64_t P, int32_t C) {
  if (P <= 0 || C <= 0) return 0;
  return (size_t)csh_chunks(P) * C * sizeof(float);
}
LIC_EXPORT int lic_colsum_bf16(const void* in, int64_t ld, int64_t P, int32_t C, float scale, float* out,
                               void* workspace, size_t workspace_bytes, lic_stream_t stream) {
  if (!in || !out || !workspace || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  if (C % 8 || ld % 8 || (reinterpret_cast<uintptr_t>(in) & 15)) return LIC_ERR_UNSUPPORTED;
  const int nchunk = csh_chunks(P);
  if (workspace_bytes < (size_t)nchunk * C * sizeof(float)) return LIC_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_bf16_stage1<false>, dim3((C + 63) / 64, nchunk), dim3(256), 0, s, (const bf16_t*)in,
                     (const bf16_t*)nullptr, (long)ld, (long)P, C, (float*)workspace, nchunk, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0.0f);
  int rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  hipLaunchKernelGGL(colsum_bf16_stage2, dim3((C + 15) / 16), dim3(256), 0, s, (const float*)workspace, C, nchunk,
                     scale, out, (float*)nullptr);
  return lic_check_launch();
}
// The quantisation surrogate (Models.py:55-64) with the bf16 copies its consumers need written by the same launch:
// out = v + (u - 0.5) (training) or rint(v); v16 = bf16(v) for the hyper-encoder, out16 = bf16(out) for the decoder, the
// context model and the hyper-decoder (either may be NULL) -- each was a cast launch of its own.  n % 4 == 0.
__global__ __launch_bounds__(256) void quantize_bf16_kernel(const float* v, const float* u, float* out, bf16_t* v16,
                                                            bf16_t* out16, long n4, int training) {
  typedef float q_f32x4 __attribute__((ext_vector_type(4)));
  typedef bf16_t q_bf16x4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const q_f32x4 a = reinterpret_cast<const q_f32x4*>(v)[i];
    q_f32x4 o;
    if (training) {
      const q_f32x4 r = reinterpret_cast<const q_f32x4*>(u)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = a[e] + (r[e] - 0.5f);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rintf(a[e]);
    }
    reinterpret_cast<q_f32x4*>(out)[i] = o;
    if (v16) reinterpret_cast<q_bf16x4*>(v16)[i] = __builtin_convertvector(a, q_bf16x4);
    if (out16) reinterpret_cast<q_bf16x4*>(out16)[i] = __builtin_convertvector(o, q_bf16x4);
  }
}
LIC_EXPORT int lic_quantize_bf16(const float* v, const float* u, float* out, void* v_bf16, void* out_bf16, int64_t n,
                                 int32_t training, lic_stream_t stream) {
  if (!v || !out || n < 0 || (training && !u)) return LIC_ERR_INVALID;
  if (n % 4 || (reinterpret_cast<uintptr_t>(v) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (u && (reinterpret_cast<uintptr_t>(u) & 15)) || (reinterpret_cast<uintptr_t>(v_bf16) & 7) ||
      (reinterpret_cast<uintptr_t>(out_bf16) & 7))
    return LIC_ERR_UNSUPPORTED;
  if (n == 0) return LIC_OK;
  hipLaunchKernelGGL(quantize_bf16_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, v, u, out,
                     (bf16_t*)v_bf16, (bf16_t*)out_bf16, (long)(n / 4), training);
  return lic_check_launch();
}

// stage 1 only; `job` (two of them for the pair variant) receives stage 2 for a later lic_reduce_batch
LIC_EXPORT int lic_colsum_bf16_partial(const void* in, int64_t ld, int64_t P, int32_t C, float scale, float* out,
                                       void* workspace, size_t workspace_bytes, lic_reduce_job* job, lic_stream_t stream) {
  if (!in || !out || !workspace || !job || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  if (C % 8 || ld % 8 || (reinterpret_cast<uintptr_t>(in) & 15)) return LIC_ERR_UNSUPPORTED;
  const int nchunk = csh_chunks(P);
  if (workspace_bytes < (size_t)nchunk * C * sizeof(float)) return LIC_ERR_WORKSPACE;
  hipLaunchKernelGGL(colsum_bf16_stage1<false>, dim3((C + 63) / 64, nchunk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                     (const bf16_t*)nullptr, (long)ld, (long)P, C, (float*)workspace, nchunk, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0.0f);
  *job = lic_reduce_job{};
  job->src = (const float*)workspace;
  job->dst = out;
  job->kind = LIC_REDUCE_COLUMNS;
  job->splitk = nchunk;
  job->Cn = C;
  job->scale = scale;
  return lic_check_launch();
}
LIC_EXPORT int lic_colsum2_bf16_partial(const void* in_a, const void* in_b, int64_t ld, int64_t P, int32_t C, float scale,
                                        float* out_a, float* out_b, void* workspace, size_t workspace_bytes,
                                        lic_reduce_job* jobs2, lic_stream_t stream) {
  if (!in_a || !in_b || !out_a || !out_b || !workspace || !jobs2 || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  if (C % 8 || ld % 8 || (reinterpret_cast<uintptr_t>(in_a) & 15) || (reinterpret_cast<uintptr_t>(in_b) & 15))
    return LIC_ERR_UNSUPPORTED;
  const int nchunk = csh_chunks(P);
  if (workspace_bytes < 2 * (size_t)nchunk * C * sizeof(float)) return LIC_ERR_WORKSPACE;
  hipLaunchKernelGGL(colsum_bf16_stage1<false>, dim3((C + 63) / 64, nchunk, 2), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in_a, (const bf16_t*)in_b, (long)ld, (long)P, C, (float*)workspace, nchunk, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0.0f);
  for (int k = 0; k < 2; ++k) {
    jobs2[k] = lic_reduce_job{};
    jobs2[k].src = (const float*)workspace + (k ? (long)nchunk * C : 0L);
    jobs2[k].dst = k ? out_b : out_a;
    jobs2[k].kind = LIC_REDUCE_COLUMNS;
    jobs2[k].splitk = nchunk;
    jobs2[k].Cn = C;
    jobs2[k].scale = scale;
  }
  return lic_check_launch();
}
// column sums of TWO bf16 [P][ld] matrices of one shape in one launch pair (the d-beta and d-bias sums of a
// conv -> GDN pair's backward: t = dL/dnorm and dL/d(conv output)); workspace: 2 x lic_colsum_bf16_workspace_bytes
LIC_EXPORT int lic_colsum2_bf16(const void* in_a, const void* in_b, int64_t ld, int64_t P, int32_t C, float scale,
                                float* out_a, float* out_b, void* workspace, size_t workspace_bytes, lic_stream_t stream) {
  if (!in_a || !in_b || !out_a || !out_b || !workspace || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  if (C % 8 || ld % 8 || (reinterpret_cast<uintptr_t>(in_a) & 15) || (reinterpret_cast<uintptr_t>(in_b) & 15))
    return LIC_ERR_UNSUPPORTED;
  const int nchunk = csh_chunks(P);
  if (workspace_bytes < 2 * (size_t)nchunk * C * sizeof(float)) return LIC_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_bf16_stage1<false>, dim3((C + 63) / 64, nchunk, 2), dim3(256), 0, s, (const bf16_t*)in_a,
                     (const bf16_t*)in_b, (long)ld, (long)P, C, (float*)workspace, nchunk, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0.0f);
  int rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  hipLaunchKernelGGL(colsum_bf16_stage2, dim3((C + 15) / 16, 1, 2), dim3(256), 0, s, (const float*)workspace, C, nchunk,
                     scale, out_a, out_b);
  return lic_check_launch();
}

// t = dL/dnorm on bf16 tensors (n elements, n % 8 == 0): inverse ? 0.5*g*x*rsqrt(n) : -0.5*g*x*n^-3/2
__global__ __launch_bounds__(256) void gdn_dnorm_bf16_kernel(const bf16_t* g, const bf16_t* x, const bf16_t* nrm,
                                                             bf16_t* t, long n8, int inverse) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const bf16x8 gv = reinterpret_cast<const bf16x8*>(g)[i], xv = reinterpret_cast<const bf16x8*>(x)[i],
                 nv = reinterpret_cast<const bf16x8*>(nrm)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float rs = __builtin_amdgcn_rsqf((float)nv[e]);
      const float gx = (float)gv[e] * (float)xv[e];
      o[e] = (bf16_t)(inverse ? 0.5f * gx * rs : -0.5f * gx * rs * (rs * rs));
    }
    reinterpret_cast<bf16x8*>(t)[i] = o;
  }
}
LIC_EXPORT int lic_gdn_dnorm_bf16(const void* g, const void* x, const void* norm, void* t, int64_t n,
                                  int32_t inverse, lic_stream_t stream) {
  if (!g || !x || !norm || !t || n < 0) return LIC_ERR_INVALID;
  if (n % 8) return LIC_ERR_UNSUPPORTED;
  if (n == 0) return LIC_OK;
  hipLaunchKernelGGL(gdn_dnorm_bf16_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)norm, (bf16_t*)t, (long)(n / 8), inverse);
  return lic_check_launch();
}

// dx = dy * (y > 0 ? 1 : slope) on bf16 tensors (y = the LeakyReLU OUTPUT: its sign is the input's);
// 8 elements per lane.  Backward of the fused LIC_EPI_LEAKY epilogue of lic_igemm_bf16.
__global__ __launch_bounds__(256) void leaky_bwd_bf16_kernel(const bf16_t* y, const bf16_t* dy, bf16_t* dx, long n8,
                                                             float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const bf16x8 a = reinterpret_cast<const bf16x8*>(y)[i], g = reinterpret_cast<const bf16x8*>(dy)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)a[e] > 0.0f ? g[e] : (bf16_t)((float)g[e] * slope);
    reinterpret_cast<bf16x8*>(dx)[i] = o;
  }
}
LIC_EXPORT int lic_leaky_bwd_bf16(const void* y, const void* dy, void* dx, int64_t n, float slope, lic_stream_t stream) {
  if (!y || !dy || !dx || n < 0) return LIC_ERR_INVALID;
  if (n % 8 || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15))
    return LIC_ERR_UNSUPPORTED;
  if (n == 0) return LIC_OK;
  hipLaunchKernelGGL(leaky_bwd_bf16_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, (long)(n / 8), slope);
  return lic_check_launch();
}
// lic_leaky_bwd_bf16 on [P][C] matrices AND the column sums of its result (the bias gradient of the layer in front of the
// LeakyReLU) in one pass.  `job` NULL: the sums' second stage follows at once (lic_colsum_bf16's); else it is left in *job for
// lic_reduce_batch (lic_colsum_bf16_partial's).  Same bits as lic_leaky_bwd_bf16 followed by lic_colsum_bf16.
LIC_EXPORT int lic_leaky_bwd_colsum_bf16(const void* y, const void* dy, void* dx, int64_t P, int32_t C, float slope, float* out,
                                         void* workspace, size_t workspace_bytes, lic_reduce_job* job, lic_stream_t stream) {
  if (!y || !dy || !dx || !out || !workspace || P <= 0 || C <= 0) return LIC_ERR_INVALID;
  if (C % 8 || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15))
    return LIC_ERR_UNSUPPORTED;
  const int nchunk = csh_chunks(P);
  if (workspace_bytes < (size_t)nchunk * C * sizeof(float)) return LIC_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_bf16_stage1<true>, dim3((C + 63) / 64, nchunk), dim3(256), 0, s, (const bf16_t*)dy,
                     (const bf16_t*)nullptr, (long)C, (long)P, C, (float*)workspace, nchunk, (const bf16_t*)y, (bf16_t*)dx,
                     slope);
  int rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  if (job) {
    *job = lic_reduce_job{};
    job->src = (const float*)workspace;
    job->dst = out;
    job->kind = LIC_REDUCE_COLUMNS;
    job->splitk = nchunk;
    job->Cn = C;
    job->scale = 1.0f;
    return LIC_OK;
  }
  hipLaunchKernelGGL(colsum_bf16_stage2, dim3((C + 15) / 16), dim3(256), 0, s, (const float*)workspace, C, nchunk, 1.0f, out,
                     (float*)nullptr);
  return lic_check_launch();
}
