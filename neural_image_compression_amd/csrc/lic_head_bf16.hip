// RGB head of the synthesis transform (Components.py:45: ConvTranspose2d(C, 3, 5, stride 2, padding 2, output_padding 1)),
// bf16 features in, fp32 image out, ONE launch -- no per-tap column matrix in HBM.
//
// The layer is C -> 3 channels: 75 MACs per INPUT value, 25 MB of image out of 134 MB of features at config 3, i.e.
// one streaming read.  The generic route wrote the [P][80] bf16 column matrix (features x the [C][25*3] weight; 84 MB)
// and gathered it back (col2im): 131 + 96 us per forward where the read alone is ~30 us.  Here a workgroup owns a
// 4 x 32 block of feature pixels plus a one-pixel halo (6 x 34: every output pixel of the 8 x 64 image block gathers
// from the feature rows q-1..q+1, columns likewise), computes the halo block's columns with MFMA -- operands swapped
// as in the fused GDN epilogues: A = the packed weight (rows = the 80 (tap, colour) columns, from registers, loaded once
// per persistent workgroup), B = features straight from global memory (lane = pixel, 16 bytes per K step), so a lane
// ends up with its pixel's columns -- parks them in LDS as fp32 ([224][84]: 75 KB, two workgroups per CU), and after one
// barrier every thread sums the <= 9 taps of two output pixels from LDS and stores 12 bytes each.
// Rounding: fp32 accumulation of bf16 x bf16 products in both steps, nothing rounded in between (the column-matrix
// route rounds the columns to bf16 first): results agree with it to bf16 rounding of the columns, and with the oracle
// to the tolerance of the bf16 operands.
#include "lic_common.h"

namespace {

typedef __bf16 hd_bf16;
typedef __bf16 hd_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned hd_u32x4 __attribute__((ext_vector_type(4)));

constexpr int HD_TH = 4, HD_TW = 32;                 // feature pixels per workgroup tile
constexpr int HD_HR = HD_TH + 2, HD_HC = HD_TW + 2;  // with the halo
constexpr int HD_HP = HD_HR * HD_HC;                 // 204 halo pixels
constexpr int HD_MT = (HD_HP + 31) / 32;             // 7 MFMA row tiles
constexpr int HD_LD = 84;                            // floats per LDS row (80 columns + 4: conflict-free 16-byte writes)

struct HeadParams {
  const hd_bf16* x;   // [B][Hi][Wi][C] bf16
  const hd_bf16* w;   // lic_pack_weight_bf16(taps 1, K = C, N = 80): element (k = ci, n = 3 * (5 ky + kx) + colour)
  const float* bias;  // [3] or null
  float* out;         // [B][2 Hi][2 Wi][3] fp32
  int B, Hi, Wi, tiles_x, tiles_y;
  long ntiles;
};

template <int NK>  // C / 16
__global__ __launch_bounds__(256, NK <= 8 ? 2 : 1) void head_convt_bf16_kernel(const HeadParams p) {
  constexpr int C = 16 * NK;
  __shared__ __attribute__((aligned(16))) float col[HD_MT * 32 * HD_LD];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // weight fragments: [k step][column tile 0..2], packed as [chunk = k step / 2][tile (of 4: N pads to 128)][k step & 1][lane][8]
  hd_bf16x8 wf[NK][3];
#pragma unroll
  for (int s = 0; s < NK; ++s)
#pragma unroll
    for (int t = 0; t < 3; ++t)
      wf[s][t] = *reinterpret_cast<const hd_bf16x8*>(p.w + ((long)((s >> 1) * 4 + t) * 2 + (s & 1)) * 512 + lane * 8);
  float bs[3] = {0.f, 0.f, 0.f};
  if (p.bias)
#pragma unroll
    for (int c = 0; c < 3; ++c) bs[c] = p.bias[c];
  const int Ho = 2 * p.Hi, Wo = 2 * p.Wi;

  for (long tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int tx = (int)(tile % p.tiles_x);
    const long rest = tile / p.tiles_x;
    const int ty = (int)(rest % p.tiles_y), b = (int)(rest / p.tiles_y);
    const int y0 = ty * HD_TH, x0 = tx * HD_TW;
    // ---- columns of the halo block: wave w takes row tiles w, w + 4
    for (int m = wave; m < HD_MT; m += 4) {
      const int hp = m * 32 + li;
      const int hr = hp / HD_HC, hc = hp - hr * HD_HC;
      const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
      const bool ok = hp < HD_HP && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
      const hd_bf16* src = p.x + (((long)b * p.Hi + (ok ? iy : 0)) * p.Wi + (ok ? ix : 0)) * C + 8 * lh;
      hd_u32x4 xq[NK];
#pragma unroll
      for (int s = 0; s < NK; ++s) xq[s] = *reinterpret_cast<const hd_u32x4*>(src + 16 * s);
      if (!ok)
#pragma unroll
        for (int s = 0; s < NK; ++s) xq[s] = hd_u32x4{0u, 0u, 0u, 0u};
      f32x16 acc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < NK; ++s) {
        const hd_bf16x8 xb = __builtin_bit_cast(hd_bf16x8, xq[s]);
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][t], xb, acc[t], 0, 0, 0);
      }
      // lane = pixel; register 4 g + j of tile t = column 32 t + 8 g + 4 lh + j
      float* row = col + hp * HD_LD + 4 * lh;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (32 * t + 8 * g < 80) {
            const f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            *reinterpret_cast<f32x4*>(row + 32 * t + 8 * g) = v;
          }
    }
    __syncthreads();
    // ---- gather: output pixel (oy, ox) = sum over ky = py + 2 a, kx = px + 2 b' of the column (ky, kx) of feature pixel
    //      (q_y + 1 - a, q_x + 1 - b'), q = o >> 1, p = o & 1
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int oyl = (tid >> 6) + 4 * h, oxl = tid & 63;
      const int qy = oyl >> 1, py = oyl & 1, qx = oxl >> 1, px = oxl & 1;
      float o0 = bs[0], o1 = bs[1], o2 = bs[2];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int ky = py + 2 * a;
        if (ky > 4) continue;
        const int hr = qy + 2 - a;   // halo row of feature row q_y + 1 - a (the halo starts one row above the tile)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
          const int kx = px + 2 * bb;
          if (kx > 4) continue;
          const int hc = qx + 2 - bb;
          const float* c3 = col + (hr * HD_HC + hc) * HD_LD + 3 * (5 * ky + kx);
          o0 += c3[0];
          o1 += c3[1];
          o2 += c3[2];
        }
      }
      const int oy = 2 * y0 + oyl, ox = 2 * x0 + oxl;
      if (oy < Ho && ox < Wo) {
        float* dst = p.out + (((long)b * Ho + oy) * Wo + ox) * 3;
        dst[0] = o0;
        dst[1] = o1;
        dst[2] = o2;
      }
    }
    __syncthreads();
  }
}

bool hd_al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int NK>
int head_launch(const HeadParams& p, hipStream_t s) {
  static const int slots = [] {
    int per_cu = 0, cus = 0, devid = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)head_convt_bf16_kernel<NK>, 256, 0) != hipSuccess || per_cu < 1)
      per_cu = 1;
    if (hipGetDevice(&devid) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || cus < 1)
      cus = 256;
    return per_cu * cus;
  }();
  const long grid = p.ntiles < slots ? p.ntiles : slots;
  hipLaunchKernelGGL((head_convt_bf16_kernel<NK>), dim3((unsigned)grid), dim3(256), 0, s, p);
  return lic_check_launch();
}

}  // namespace

LIC_EXPORT int lic_head_convt_bf16_supported(int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                                             int32_t out_pad) {
  return (Cin == 64 || Cin == 128 || Cin == 192) && Cout == 3 && kh == 5 && kw == 5 && stride == 2 && pad == 2 && out_pad == 1;
}

// x: bf16 [B][Hi][Wi][Cin]; w_packed: lic_pack_weight_bf16(taps 1, K = Cin, N = 80) of the [Cin][80] matrix
// w[ci][3 * (5 ky + kx) + colour] (the operand of the column-matrix route); out: fp32 [B][2 Hi][2 Wi][3]
LIC_EXPORT int lic_head_convt_bf16(const void* x, const void* w_packed, const float* bias, float* out, int32_t B, int32_t Hi,
                                   int32_t Wi, int32_t Cin, lic_stream_t stream) {
  if (!x || !w_packed || !out || B <= 0 || Hi <= 0 || Wi <= 0) return LIC_ERR_INVALID;
  if (!(Cin == 64 || Cin == 128 || Cin == 192)) return LIC_ERR_UNSUPPORTED;
  if (!hd_al16(x) || !hd_al16(w_packed) || (reinterpret_cast<uintptr_t>(out) & 3)) return LIC_ERR_INVALID;
  HeadParams p;
  p.x = (const hd_bf16*)x;
  p.w = (const hd_bf16*)w_packed;
  p.bias = bias;
  p.out = out;
  p.B = B;
  p.Hi = Hi;
  p.Wi = Wi;
  p.tiles_x = (Wi + HD_TW - 1) / HD_TW;
  p.tiles_y = (Hi + HD_TH - 1) / HD_TH;
  p.ntiles = (long)B * p.tiles_x * p.tiles_y;
  if ((long)B * Hi * Wi * Cin > 0x7FFFFFFFFFL) return LIC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 64) return head_launch<4>(p, s);
  if (Cin == 128) return head_launch<8>(p, s);
  return head_launch<12>(p, s);
}
