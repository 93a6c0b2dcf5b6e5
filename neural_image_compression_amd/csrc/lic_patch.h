// Patch <-> column conversion for the two RGB-side layers (3 -> M stem, M -> 3 head), shared by the
// fp32 and bf16-storage builds: 16-byte column stores, 32-bit index arithmetic, the k -> (r, s, c)
// decomposition looked up from LDS instead of divided out per element.
#pragma once
#include "lic_common.h"

#define LIC_PATCH_MAXK 512

template <typename T, int V>
struct lic_patch_vec {
  typedef T type __attribute__((ext_vector_type(V)));
};

// col[(b,oh,ow)][(r*kw+s)*C + c] = x[b, oh*stride-pad+r, ow*stride-pad+s, c], 0 outside / in the K padding.
// One lane per V consecutive k of one output pixel (V * sizeof(T) == 16).
template <typename T, int V>
__global__ __launch_bounds__(256) void im2col_vec_kernel(const float* x, T* col, unsigned npix, int H, int W, int C,
                                                         int Ho, int Wo, int kh, int kw, int stride, int pad,
                                                         int Kpad) {
  __shared__ int lut[LIC_PATCH_MAXK];
  const int K = kh * kw * C;
  for (int k = threadIdx.x; k < Kpad; k += 256) {
    int code = -1;
    if (k < K) {
      const int tap = k / C, c = k - tap * C;
      const int r = tap / kw, s = tap - r * kw;
      code = (r << 16) | (s << 8) | c;
    }
    lut[k] = code;
  }
  __syncthreads();
  typedef typename lic_patch_vec<T, V>::type vec_t;
  const unsigned KV = (unsigned)Kpad / V;
  const unsigned total = npix * KV;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const unsigned pix = i / KV, kv = i - pix * KV;
    const unsigned t = pix / (unsigned)Wo, ow = pix - t * Wo;
    const unsigned b = t / (unsigned)Ho, oh = t - b * Ho;
    const int ih0 = (int)oh * stride - pad, iw0 = (int)ow * stride - pad;
    const float* xb = x + (long)b * H * W * C;
    vec_t o;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int code = lut[kv * V + e];
      float v = 0.0f;
      if (code >= 0) {
        const int ih = ih0 + (code >> 16), iw = iw0 + ((code >> 8) & 255);
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = xb[(ih * W + iw) * C + (code & 255)];
      }
      o[e] = (T)v;
    }
    *reinterpret_cast<vec_t*>(col + (long)i * V) = o;
  }
}

// out[b,oy,ox,c] = bias[c] + sum_{r,s} col[(b,ih,iw)][(r*kw+s)*C + c] with oy = ih*stride-pad+r (taps summed
// in ascending (r, s) order).  Only the taps whose parity matches are visited.
template <typename T>
__global__ __launch_bounds__(256) void col2im_fast_kernel(const T* col, const float* bias, float* out,
                                                          unsigned total, int Hi, int Wi, int C, int Ho, int Wo,
                                                          int kh, int kw, int stride, int pad, int Kpad) {
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const unsigned pix = i / (unsigned)C, c = i - pix * C;
    const unsigned t = pix / (unsigned)Wo, ox = pix - t * Wo;
    const unsigned b = t / (unsigned)Ho, oy = t - b * Ho;
    float v = bias ? bias[c] : 0.0f;
    const int ay = (int)oy + pad, ax = (int)ox + pad;
    const int ph = ay % stride, pw = ax % stride;
    for (int r = ph; r < kh && r <= ay; r += stride) {
      const int ih = (ay - r) / stride;
      if (ih >= Hi) continue;
      const T* row = col + ((long)(b * Hi + ih) * Wi) * Kpad + r * kw * C + c;
      for (int s = pw; s < kw && s <= ax; s += stride) {
        const int iw = (ax - s) / stride;
        if (iw >= Wi) continue;
        v += (float)row[(long)iw * Kpad + s * C];
      }
    }
    out[i] = v;
  }
}
