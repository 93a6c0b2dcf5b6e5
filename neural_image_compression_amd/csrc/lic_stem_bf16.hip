// RGB stem of the analysis transform + the GDN behind it (Components.py:10-11), bf16 storage, ONE launch that
// reads the fp32 image and writes the normalised bf16 feature map -- no column matrix in between.
//
// The layer is 3 -> C channels, 5x5, stride 2: 75 MACs per output and a 2^26-element output at config 3, i.e.
// pure streaming (25 MB in, 134 MB out) with a little matrix work.  What made the generic path (im2col 63 us +
// LDS-DMA GEMM with the fused pool 130 us) slow was not bandwidth but the latency chain of a workgroup that
// lives for one tile: parameters, weights, gamma, bias and beta fetched anew 4096 times, two workgroups per CU
// in flight.  Here:
//   * persistent workgroups: the packed weights (K = 5 filter rows x 16), gamma_eff^T, bias and beta_eff are
//     copied to LDS once and a workgroup then walks over its 128-pixel tiles;
//   * one MFMA K step (16) == one filter row: the 5 taps x 3 channels a pixel needs from image row 2*oy+r-2 are
//     15 CONSECUTIVE floats of the NHWC image, so a lane's B operand (8 consecutive K) is two 16-byte loads
//     straight from the image (the 6x overlap between neighbouring windows is L1/L2 traffic), converted to bf16
//     in registers; slot 15 of a row multiplies a zero weight;
//   * the accumulators hold the transposed tile (lane = pixel; channels 4*lh + 8*g + j of every 32-channel
//     tile), as in igemm_bf16_kernel's FUSE variant: x -> bf16, x^2 -> bf16 and the pool's operand stay in
//     registers, gamma_eff^T comes from LDS in lic_pack_weight_bf16_kperm's order, y = x * rsqrt(norm) is
//     element-wise, and v_permlane32_swap turns a lane's 4+4 channels into 8 consecutive ones for 16-byte stores.
// Rounding points are those of conv2d_bf16 -> gdn_bf16 (x and x^2 to bf16, fp32 norm).
#include "lic_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // image rows are only 4-byte aligned

struct StemParams {
  const float* x;       // [B][H][W][3] fp32
  const bf16_t* w;      // packed [3 chunks][C/32][2][64][8]: K = 16*r + 3*s + c (slot 15 of a row and rows 5 = 0)
  const float* bias;    // [C] or null
  const bf16_t* gamma;  // gamma_eff^T, lic_pack_weight_bf16_kperm(taps 1, K = N = C)
  const float* beta;    // beta_eff [C]
  bf16_t* y;            // [B][Ho][Wo][C]
  bf16_t* conv_out;     // or null
  bf16_t* norm;         // or null
  int B, H, W, Ho, Wo, inverse;
  long P;
  int ntiles;
};

__device__ __forceinline__ unsigned pack2(f32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// TW: 32-channel tiles (C = 32*TW); NW: waves per workgroup (tile = 32*NW pixels).
// (C = 128 compiles to 184 VGPR + 64 AGPR: two waves per SIMD.  Forced to three -- 168 registers, 60 bytes of
// scratch -- it ran 100 us instead of 72 at config 3.  Issuing the NEXT tile's window loads under the current
// tile's epilogue keeps 40 more registers live: one wave per SIMD, 94 us.  Neither kept.)
// PLAIN: the convolution only (+ bias), written to p.y -- the data gradient of the RGB head (Components.py:45: the
// gradient of ConvTranspose2d(C, 3, 5, 2, 2, 1) w.r.t. its input IS this convolution of the image gradient with the same
// [C][3][5][5] weight), which ran as im2col + GEMM before.
template <int TW, int NW, bool PLAIN = false>
__global__ __launch_bounds__(64 * NW) void stem_gdn_bf16_kernel(const StemParams p) {
  constexpr int C = 32 * TW;
  constexpr int WEL = 5 * TW * 512;   // weight elements kept in LDS: k steps 0..4 of [step][tile][lane][8]
  constexpr int GEL = PLAIN ? 8 : TW * TW * 1024;  // gamma elements [t][tile][2][lane][8]
  __shared__ __attribute__((aligned(16))) bf16_t s_w[WEL];
  __shared__ __attribute__((aligned(16))) bf16_t s_g[GEL];
  __shared__ __attribute__((aligned(16))) float s_bias[C];
  __shared__ __attribute__((aligned(16))) float s_beta[C];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the packed weight is [chunk cb][tile][q][lane][8] with k step r = 2*cb + q: re-index to [r][tile][lane][8]
  for (int i = tid * 8; i < WEL; i += 64 * NW * 8) {
    const int r = i / (TW * 512), rem = i - r * (TW * 512), tile = rem >> 9, o = rem & 511;
    *reinterpret_cast<bf16x8*>(s_w + i) =
        *reinterpret_cast<const bf16x8*>(p.w + ((long)((r >> 1) * TW + tile) * 2 + (r & 1)) * 512 + o);
  }
  if (!PLAIN)
    for (int i = tid * 8; i < GEL; i += 64 * NW * 8)
      *reinterpret_cast<bf16x8*>(s_g + i) = *reinterpret_cast<const bf16x8*>(p.gamma + i);
  for (int i = tid; i < C; i += 64 * NW) {
    s_bias[i] = p.bias ? p.bias[i] : 0.0f;
    s_beta[i] = PLAIN ? 0.0f : p.beta[i];
  }
  __syncthreads();

  const int HW = p.Ho * p.Wo;
  const int rowf = 3 * p.W;  // floats per image row
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    // (compiler barrier: without it the loop-invariant LDS reads of all weight and gamma fragments are hoisted
    // out of the tile loop into ~200 registers)
    asm volatile("" ::: "memory");
    const long prow = (long)tile * (32 * NW) + wave * 32 + li;
    const bool rok = prow < p.P;
    const long pr = rok ? prow : p.P - 1;  // (a real pixel: the loads stay inside the image)
    const unsigned b = (unsigned)pr / (unsigned)HW;
    const unsigned rem = (unsigned)pr - b * (unsigned)HW;
    const int oy = (int)(rem / (unsigned)p.Wo), ox = (int)rem - oy * p.Wo;
    const int cf0 = (2 * ox - 2) * 3 + 8 * lh;  // first of this lane's 8 floats within an image row
    const bool colfast = cf0 >= 0 && cf0 + 8 <= rowf;
    const float* img = p.x + (long)b * p.H * rowf;

    f32x16 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int iy = 2 * oy - 2 + r;
      const bool rowok = iy >= 0 && iy < p.H;
      const float* src = img + (long)(rowok ? iy : 0) * rowf;
      f32x4 v0 = {0.0f, 0.0f, 0.0f, 0.0f}, v1 = {0.0f, 0.0f, 0.0f, 0.0f};
      if (colfast) {
        if (rowok) {
          v0 = *reinterpret_cast<const f32x4u*>(src + cf0);
          v1 = *reinterpret_cast<const f32x4u*>(src + cf0 + 4);
        }
      } else {  // image border: element-wise, zero outside (padding = 2).  Branch-free -- clamped index, then a
                // select -- so the eight loads are in flight together: as conditional loads each one was a
                // full memory round trip (18 us per tile in every wave that holds a border pixel)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c0 = cf0 + e, c1 = cf0 + 4 + e;
          const bool ok0 = rowok && c0 >= 0 && c0 < rowf, ok1 = rowok && c1 >= 0 && c1 < rowf;
          const float t0 = src[ok0 ? c0 : 0], t1 = src[ok1 ? c1 : 0];
          v0[e] = ok0 ? t0 : 0.0f;
          v1[e] = ok1 ? t1 : 0.0f;
        }
      }
      const u32x4 bq = {pack2(f32x2{v0[0], v0[1]}), pack2(f32x2{v0[2], v0[3]}), pack2(f32x2{v1[0], v1[1]}),
                        pack2(f32x2{v1[2], v1[3]})};
      const bf16x8 bop = __builtin_bit_cast(bf16x8, bq);
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const bf16x8 aop = *reinterpret_cast<const bf16x8*>(s_w + (r * TW + t) * 512 + lane * 8);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aop, bop, acc[t], 0, 0, 0);
      }
    }

    auto store_tile = [&](bf16_t* base, const unsigned (&pk)[8], int t) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x2 r0 = __builtin_amdgcn_permlane32_swap(pk[4 * s], pk[4 * s + 2], false, false);
        const u32x2 r1 = __builtin_amdgcn_permlane32_swap(pk[4 * s + 1], pk[4 * s + 3], false, false);
        if (rok) {
          const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
          *reinterpret_cast<u32x4*>(base + prow * C + t * 32 + 16 * s + 8 * lh) = v;
        }
      }
    };
    // x = conv + bias -> bf16; x^2 -> bf16 (the pool's B operand)
    unsigned sqpk[TW][8];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      unsigned xpk[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bs = *reinterpret_cast<const f32x4*>(s_bias + t * 32 + 4 * lh + 8 * g);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 v = {acc[t][4 * g + 2 * h] + bs[2 * h], acc[t][4 * g + 2 * h + 1] + bs[2 * h + 1]};
          const unsigned pk = pack2(v);
          xpk[2 * g + h] = pk;
          const f32x2 xb = {__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
          acc[t][4 * g + 2 * h] = xb[0];
          acc[t][4 * g + 2 * h + 1] = xb[1];
          sqpk[t][2 * g + h] = pack2(xb * xb);
        }
      }
      if (PLAIN) store_tile(p.y, xpk, t);
      else if (p.conv_out) store_tile(p.conv_out, xpk, t);
    }
    if (PLAIN) continue;
    asm volatile("" ::: "memory");
    // per output-channel tile: norm^T = gamma_eff . (x^2)^T + beta, y = x * norm^-1/2 (or ^1/2)
    auto finish = [&](auto inv) {
      constexpr bool INV = decltype(inv)::value;
#pragma unroll
      for (int bo = 0; bo < TW; ++bo) {
        f32x16 nacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) nacc[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const u32x4 bq = {sqpk[t][4 * s], sqpk[t][4 * s + 1], sqpk[t][4 * s + 2], sqpk[t][4 * s + 3]};
            const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(s_g + ((t * TW + bo) * 2 + s) * 512 + lane * 8);
            nacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, __builtin_bit_cast(bf16x8, bq), nacc, 0, 0, 0);
          }
        unsigned npk[8], ypk[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 be = *reinterpret_cast<const f32x4*>(s_beta + bo * 32 + 4 * lh + 8 * g);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 nv = {nacc[4 * g + 2 * h] + be[2 * h], nacc[4 * g + 2 * h + 1] + be[2 * h + 1]};
            npk[2 * g + h] = pack2(nv);
            const f32x2 f = {INV ? __builtin_amdgcn_sqrtf(nv[0]) : __builtin_amdgcn_rsqf(nv[0]),
                             INV ? __builtin_amdgcn_sqrtf(nv[1]) : __builtin_amdgcn_rsqf(nv[1])};
            const f32x2 xv = {acc[bo][4 * g + 2 * h], acc[bo][4 * g + 2 * h + 1]};
            ypk[2 * g + h] = pack2(xv * f);
          }
        }
        if (p.norm) store_tile(p.norm, npk, bo);
        store_tile(p.y, ypk, bo);
      }
    };
    if (p.inverse) finish(std::true_type{});
    else finish(std::false_type{});
  }
}

// stem weight [C][3][5][5] fp32 -> bf16 MFMA A operand, K = 16*r + 3*s + c (see StemParams::w)
__global__ __launch_bounds__(256) void pack_stem_weight_bf16_kernel(const float* src, bf16_t* dst, int C, int Npad) {
  const int ntile = Npad >> 5;
  const long total = 3L * Npad * 32;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), q = (int)((i >> 9) & 1);
    const long t = i >> 10;
    const int tile = (int)(t % ntile), cb = (int)(t / ntile);
    const int n = tile * 32 + (lane & 31);
    const int k = cb * 32 + q * 16 + (lane >> 5) * 8 + e;
    const int r = k >> 4, j = k & 15;
    float v = 0.0f;
    if (r < 5 && j < 15 && n < C) {
      const int s = j / 3, c = j - 3 * s;
      v = src[n * 75 + c * 25 + r * 5 + s];
    }
    dst[i] = (bf16_t)v;
  }
}

bool st_al16(const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

LIC_EXPORT int lic_stem_gdn_bf16_supported(int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                                           int32_t pad) {
  return Cin == 3 && kh == 5 && kw == 5 && stride == 2 && pad == 2 && (Cout == 64 || Cout == 128 || Cout == 192);
}

LIC_EXPORT int64_t lic_stem_weight_bf16_elems(int32_t Cout) {
  return Cout > 0 ? 3LL * (((Cout + 63) / 64) * 64) * 32 : 0;
}

LIC_EXPORT int lic_pack_stem_weight_bf16(const float* w, void* dst, int32_t Cout, lic_stream_t stream) {
  if (!w || !dst || Cout <= 0 || !st_al16(dst)) return LIC_ERR_INVALID;
  const int Npad = ((Cout + 63) / 64) * 64;
  hipLaunchKernelGGL(pack_stem_weight_bf16_kernel, dim3(ew_grid(3L * Npad * 32, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, (bf16_t*)dst, Cout, Npad);
  return lic_check_launch();
}

// The convolution alone: y[B][ceil(H/2)][ceil(W/2)][Cout] (bf16) = conv2d(x fp32 [B][H][W][3], w, stride 2, padding 2)
// + bias.  Used as the data gradient of the RGB head: x = dL/d(image), w = the ConvTranspose2d weight [Cout][3][5][5].
LIC_EXPORT int lic_stem_conv_bf16(const float* x, const void* w_packed, const float* bias, void* y, int32_t B, int32_t H,
                                  int32_t W, int32_t Cout, lic_stream_t stream) {
  if (!x || !w_packed || !y || B <= 0 || H <= 0 || W <= 0) return LIC_ERR_INVALID;
  if (!(Cout == 64 || Cout == 128 || Cout == 192)) return LIC_ERR_UNSUPPORTED;
  if (!st_al16(w_packed) || !st_al16(y) || (reinterpret_cast<uintptr_t>(x) & 3)) return LIC_ERR_INVALID;
  StemParams p = {};
  p.x = x;
  p.w = (const bf16_t*)w_packed;
  p.bias = bias;
  p.y = (bf16_t*)y;
  p.B = B;
  p.H = H;
  p.W = W;
  p.Ho = (H + 1) / 2;
  p.Wo = (W + 1) / 2;
  p.P = (long)B * p.Ho * p.Wo;
  if (p.P > 0x7FFFFFFFL / 2 || (long)B * H * W * 3 > 0x7FFFFFFFFFL) return LIC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  auto resident = [](const void* fn, int threads) {
    int per_cu = 0, cus = 0, devid = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&devid) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || cus < 1)
      cus = 256;
    return per_cu * cus;
  };
  if (Cout == 192) {
    static const int slots = resident((const void*)stem_gdn_bf16_kernel<6, 8, true>, 512);
    p.ntiles = (int)((p.P + 255) / 256);
    hipLaunchKernelGGL((stem_gdn_bf16_kernel<6, 8, true>), dim3(p.ntiles < slots ? p.ntiles : slots), dim3(512), 0, s, p);
  } else if (Cout == 128) {
    static const int slots = resident((const void*)stem_gdn_bf16_kernel<4, 4, true>, 256);
    p.ntiles = (int)((p.P + 127) / 128);
    hipLaunchKernelGGL((stem_gdn_bf16_kernel<4, 4, true>), dim3(p.ntiles < slots ? p.ntiles : slots), dim3(256), 0, s, p);
  } else {
    static const int slots = resident((const void*)stem_gdn_bf16_kernel<2, 4, true>, 256);
    p.ntiles = (int)((p.P + 127) / 128);
    hipLaunchKernelGGL((stem_gdn_bf16_kernel<2, 4, true>), dim3(p.ntiles < slots ? p.ntiles : slots), dim3(256), 0, s, p);
  }
  return lic_check_launch();
}

LIC_EXPORT int lic_stem_gdn_bf16(const float* x, const void* w_packed, const float* bias, const void* gamma_packed,
                                 const float* beta_eff, void* y, void* conv_out, void* norm, int32_t B, int32_t H,
                                 int32_t W, int32_t Cout, int32_t inverse, lic_stream_t stream) {
  if (!x || !w_packed || !gamma_packed || !beta_eff || !y || B <= 0 || H <= 0 || W <= 0) return LIC_ERR_INVALID;
  if (!(Cout == 64 || Cout == 128 || Cout == 192)) return LIC_ERR_UNSUPPORTED;
  if (!st_al16(w_packed) || !st_al16(gamma_packed) || !st_al16(y) || !st_al16(conv_out) || !st_al16(norm) ||
      (reinterpret_cast<uintptr_t>(x) & 3))
    return LIC_ERR_INVALID;
  StemParams p;
  p.x = x;
  p.w = (const bf16_t*)w_packed;
  p.bias = bias;
  p.gamma = (const bf16_t*)gamma_packed;
  p.beta = beta_eff;
  p.y = (bf16_t*)y;
  p.conv_out = (bf16_t*)conv_out;
  p.norm = (bf16_t*)norm;
  p.B = B;
  p.H = H;
  p.W = W;
  p.Ho = (H + 1) / 2;  // k 5, stride 2, pad 2
  p.Wo = (W + 1) / 2;
  p.inverse = inverse ? 1 : 0;
  p.P = (long)B * p.Ho * p.Wo;
  if (p.P > 0x7FFFFFFFL / 2 || (long)B * H * W * 3 > 0x7FFFFFFFFFL) return LIC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  // persistent grid: exactly the workgroups that are resident at once (registers and LDS decide; asked once)
  auto resident = [](const void* fn, int threads) {
    int per_cu = 0, cus = 0, devid = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&devid) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || cus < 1)
      cus = 256;
    return per_cu * cus;
  };
  if (Cout == 192) {
    static const int slots = resident((const void*)stem_gdn_bf16_kernel<6, 8>, 512);
    p.ntiles = (int)((p.P + 255) / 256);
    hipLaunchKernelGGL((stem_gdn_bf16_kernel<6, 8>), dim3(p.ntiles < slots ? p.ntiles : slots), dim3(512), 0, s, p);
  } else if (Cout == 128) {
    static const int slots = resident((const void*)stem_gdn_bf16_kernel<4, 4>, 256);
    p.ntiles = (int)((p.P + 127) / 128);
    hipLaunchKernelGGL((stem_gdn_bf16_kernel<4, 4>), dim3(p.ntiles < slots ? p.ntiles : slots), dim3(256), 0, s, p);
  } else {
    static const int slots = resident((const void*)stem_gdn_bf16_kernel<2, 4>, 256);
    p.ntiles = (int)((p.P + 127) / 128);
    hipLaunchKernelGGL((stem_gdn_bf16_kernel<2, 4>), dim3(p.ntiles < slots ? p.ntiles : slots), dim3(256), 0, s, p);
  }
  return lic_check_launch();
}
