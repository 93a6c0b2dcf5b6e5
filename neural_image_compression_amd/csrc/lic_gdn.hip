// Stand-alone GDN / IGDN kernels (third-party compressai GDN as used at Components.py:11-44 and
// Layers.py:41,75; definition: SURVEY.md Appendix B):
//   forward   norm = beta + x^2 . gamma^T,  y = x * rsqrt(norm)   (IGDN: x * sqrt(norm))
//   backward  t = dL/dnorm(g, x, norm),     dx = g * rsqrt(norm) + 2 x (t . gamma)   (IGDN: sqrt)
// The C x C pool is only 2C FLOP per activation byte-quad, so these launches live between the MFMA
// and the HBM roof: what matters is touching each activation stream ONCE with whole-row loads.
// A workgroup owns 64 pixels x all C channels (C in {64,128,192}): it loads its tile in one
// coalesced sweep into LDS ([64][C+4]: the +4 shifts consecutive rows by one 16-byte bank slot, so the
// 32-row ds_read_b128 fragments are conflict-free), contracts it against the packed gamma panel
// (read from L2 like a conv weight, same chunk / k order as lic_igemm's 1x1 path => bitwise the same
// sums), and finishes in the 16-byte layout.  The generic lic_igemm route (prologue 1 / 2 / 3) remains
// for other channel counts.
#include "lic_common.h"
#include <stdlib.h>

namespace {

constexpr int GD_BM = 64;
constexpr int GD_BK = 16;

struct GdnParams {
  const float* x;      // fwd: input; bwd: saved input
  const float* g;      // bwd: output gradient
  const float* norm;   // bwd: saved pool
  const float* w;      // packed gamma panel (lic_pack_weight, taps = 1, K = N = C)
  const float* beta;   // fwd: beta_eff [C]
  const float* res;    // fwd: optional residual added to y
  float* out;          // fwd: y; bwd: dx
  float* out2;         // fwd: norm; bwd: t
  float* cs_t;         // bwd, optional: per-workgroup column sums of t   [gridDim.x][C] (-> d beta)
  float* cs_dx;        // bwd, optional: per-workgroup column sums of dx  [gridDim.x][C] (-> the conv's d bias)
  long P;              // pixels
  int inverse;
};

// MODE 0 = forward, 1 = backward
template <int TN, int MODE>
__global__ __launch_bounds__(256) void gdn_kernel(const GdnParams p) {
  constexpr int C = 64 * TN;
  constexpr int LDX = C + 4;
  constexpr int NCH = C / GD_BK;
  constexpr int SLOTS = GD_BM * C / 4 / 256;  // float4 slots per thread of the tile sweep (4 TN)
  __shared__ __attribute__((aligned(16))) float smem[GD_BM * LDX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * (32 * TN);
  const int li = lane & 31, lh = lane >> 5;
  const long m0 = (long)blockIdx.x * GD_BM;
  const bool inv = p.inverse != 0;

  // ---- one sweep over the tile: whole rows, 16 bytes per lane ------------------------------------
  if (MODE == 0) {
    // forward: all SLOTS loads in flight at once (one HBM round trip per tile instead of SLOTS / 4: the
    // sweep was 28 % of the waves' time in s_waitcnt, SQ_WAIT_ANY)
    f32x4 a[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
      const int idx = u * 256 + tid;
      const int row = idx / (C / 4), c4 = (idx - row * (C / 4)) * 4;
      const bool ok = m0 + row < p.P;
      a[u] = *reinterpret_cast<const f32x4*>(p.x + (ok ? (m0 + row) * C + c4 : 0L));
      if (!ok) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
      const int idx = u * 256 + tid;
      const int row = idx / (C / 4), c4 = (idx - row * (C / 4)) * 4;
      *reinterpret_cast<f32x4*>(&smem[row * LDX + c4]) = a[u];
    }
  } else {
#pragma unroll
  for (int s0 = 0; s0 < SLOTS; s0 += 4) {
    f32x4 a[4], b[4], c[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = (s0 + u) * 256 + tid;
      const int row = idx / (C / 4), c4 = (idx - row * (C / 4)) * 4;
      ok[u] = m0 + row < p.P;
      const long off = ok[u] ? (m0 + row) * C + c4 : 0L;
      a[u] = *reinterpret_cast<const f32x4*>(p.x + off);
      b[u] = *reinterpret_cast<const f32x4*>(p.g + off);
      c[u] = *reinterpret_cast<const f32x4*>(p.norm + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = (s0 + u) * 256 + tid;
      const int row = idx / (C / 4), c4 = (idx - row * (C / 4)) * 4;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float rs = __builtin_amdgcn_rsqf(c[u][e]);  // as lic_gdn_dnorm
        const float gx = b[u][e] * a[u][e];
        v[e] = inv ? 0.5f * gx * rs : -0.5f * gx * rs * (rs * rs);
      }
      if (ok[u]) *reinterpret_cast<f32x4*>(p.out2 + (m0 + row) * C + c4) = v;
      if (!ok[u]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&smem[row * LDX + c4]) = v;
    }
  }
  }
  __syncthreads();
  if (MODE == 1 && p.cs_t && tid < C) {  // column sums of the t tile (rows past P are zero), fixed order
    float a = 0.0f;
#pragma unroll 8
    for (int r = 0; r < GD_BM; ++r) a += smem[r * LDX + tid];
    p.cs_t[(long)blockIdx.x * C + tid] = a;
  }

  // ---- pool: acc[32 x 32*TN per wave] = A . panel, A = x^2 (forward) or t (backward) --------------
  f32x16 acc[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
  const float* glane = p.w + ((long)(wn0 >> 5) * 512 + lane * 4);
  const float* xrow = smem + (wm0 + li) * LDX + lh * 8;
  f32x4 g0[TN][2], g1[TN][2];
  auto load_g = [&](f32x4 (&rg)[TN][2], int c) {
    const float* src = glane + (long)(c < NCH ? c : NCH - 1) * C * GD_BK;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      rg[b][0] = *reinterpret_cast<const f32x4*>(src + b * 512);
      rg[b][1] = *reinterpret_cast<const f32x4*>(src + b * 512 + 256);
    }
  };
  auto pool = [&](int c, const f32x4 (&rg)[TN][2]) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(xrow + c * GD_BK);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(xrow + c * GD_BK + 4);
    if (MODE == 0) {
      a0 = a0 * a0;
      a1 = a1 * a1;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32((t < 4 ? a0 : a1)[t & 3], rg[b][t >> 2][t & 3], acc[b], 0, 0, 0);
  };
  load_g(g0, 0);
#pragma unroll 1
  for (int c = 0; c < NCH; c += 2) {
    load_g(g1, c + 1);
    pool(c, g0);
    load_g(g0, c + 2);
    pool(c + 1, g1);
  }

  // ---- finish in the 16-byte layout: each 32x32 accumulator tile through a wave-private LDS patch --
  // (forward keeps x in the tile; the patches live behind it only when there is room, else the tile's
  // own rows of OTHER waves must not be overwritten: use a separate barrier-protected region)
  // backward: the t tile is dead after the pool, so the patches alias it (50 KiB per workgroup -> three
  // resident per CU); forward still needs x from the tile and keeps separate patches (two per CU)
  __shared__ __attribute__((aligned(16))) float patch[MODE == 0 ? 4 : 1][MODE == 0 ? 1024 : 4];
  if (MODE == 1) __syncthreads();
  float* stg = MODE == 0 ? patch[wave] : smem + wave * 1024;
  const int c4 = (lane & 7) * 4, r8 = lane >> 3;
  f32x4 csum[TN];  // backward: this lane's share of the column sums of dx
#pragma unroll
  for (int b = 0; b < TN; ++b) csum[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < TN; ++b) {
#pragma unroll
    for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[b][r];
    __builtin_amdgcn_wave_barrier();
    const int col = wn0 + b * 32 + c4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = it * 8 + r8;
      const long pix = m0 + wm0 + rr;
      if (pix >= p.P) continue;
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(&stg[rr * 32 + c4]);
      const long off = pix * C + col;
      f32x4 o;
      if (MODE == 0) {
        const f32x4 n4 = s4 + *reinterpret_cast<const f32x4*>(p.beta + col);
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(&smem[(wm0 + rr) * LDX + col]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = x4[e] * (inv ? __builtin_amdgcn_sqrtf(n4[e]) : __builtin_amdgcn_rsqf(n4[e]));
        if (p.out2) *reinterpret_cast<f32x4*>(p.out2 + off) = n4;
        if (p.res) o += *reinterpret_cast<const f32x4*>(p.res + off);
      } else {
        const f32x4 n4 = *reinterpret_cast<const f32x4*>(p.norm + off);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.g + off);
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(p.x + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f = inv ? __builtin_amdgcn_sqrtf(n4[e]) : __builtin_amdgcn_rsqf(n4[e]);
          o[e] = g4[e] * f + 2.0f * x4[e] * s4[e];
        }
        csum[b] += o;
      }
      *reinterpret_cast<f32x4*>(p.out + off) = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (MODE == 1 && p.cs_dx) {
    // lanes with equal lane%8 own the same 4 columns (8 row groups), the two wave rows the same columns:
    // park the 16 partials per column in LDS and add them in a fixed order
    __syncthreads();  // every wave is done with its patch
    float* part = smem;  // [2 wave rows][8 row groups][C]
#pragma unroll
    for (int b = 0; b < TN; ++b)
      *reinterpret_cast<f32x4*>(&part[(((wave >> 1) * 8 + r8) * C) + wn0 + b * 32 + c4]) = csum[b];
    __syncthreads();
    if (tid < C) {
      float a = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) a += part[k * C + tid];
      p.cs_dx[(long)blockIdx.x * C + tid] = a;
    }
  }
}

// Backward with the epilogue operands kept in registers: the tile sweep uses the SAME (row, 4-channel)
// ownership as the 16-byte epilogue, so each thread still holds g * f(norm) and x for exactly the
// elements it finishes -- the epilogue issues no loads at all and every stream crosses HBM once
// (3 reads, 2 writes).  96 more registers: two workgroups per CU instead of three.
template <int TN>
__global__ __launch_bounds__(256, 2) void gdn_bwd_reg_kernel(const GdnParams p) {
  constexpr int C = 64 * TN;
  constexpr int LDX = C + 4;
  constexpr int NCH = C / GD_BK;
  __shared__ __attribute__((aligned(16))) float smem[GD_BM * LDX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * (32 * TN);
  const int li = lane & 31, lh = lane >> 5;
  const int c4 = (lane & 7) * 4, r8 = lane >> 3;
  const long m0 = (long)blockIdx.x * GD_BM;
  const bool inv = p.inverse != 0;

  f32x4 u[TN][4], xx[TN][4];  // g * f(norm) and x of this thread's epilogue elements
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    f32x4 gg[4], nn[4];
    bool ok[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long pix = m0 + wm0 + it * 8 + r8;
      ok[it] = pix < p.P;
      const long off = ok[it] ? pix * C + wn0 + b * 32 + c4 : 0L;
      xx[b][it] = *reinterpret_cast<const f32x4*>(p.x + off);
      gg[it] = *reinterpret_cast<const f32x4*>(p.g + off);
      nn[it] = *reinterpret_cast<const f32x4*>(p.norm + off);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = wm0 + it * 8 + r8, col = wn0 + b * 32 + c4;
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float rs = __builtin_amdgcn_rsqf(nn[it][e]);  // as lic_gdn_dnorm
        const float gx = gg[it][e] * xx[b][it][e];
        t[e] = inv ? 0.5f * gx * rs : -0.5f * gx * rs * (rs * rs);
        u[b][it][e] = gg[it][e] * (inv ? __builtin_amdgcn_sqrtf(nn[it][e]) : rs);
      }
      if (ok[it]) {
        *reinterpret_cast<f32x4*>(p.out2 + (m0 + row) * C + col) = t;
      } else {
        t = f32x4{0.f, 0.f, 0.f, 0.f};
        u[b][it] = t;
        xx[b][it] = t;
      }
      *reinterpret_cast<f32x4*>(&smem[row * LDX + col]) = t;
    }
  }
  __syncthreads();
  if (p.cs_t && tid < C) {
    float a = 0.0f;
#pragma unroll 8
    for (int r = 0; r < GD_BM; ++r) a += smem[r * LDX + tid];
    p.cs_t[(long)blockIdx.x * C + tid] = a;
  }

  f32x16 acc[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
  const float* glane = p.w + ((long)(wn0 >> 5) * 512 + lane * 4);
  const float* xrow = smem + (wm0 + li) * LDX + lh * 8;
  f32x4 g0[TN][2], g1[TN][2];
  auto load_g = [&](f32x4 (&rg)[TN][2], int c) {
    const float* src = glane + (long)(c < NCH ? c : NCH - 1) * C * GD_BK;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      rg[b][0] = *reinterpret_cast<const f32x4*>(src + b * 512);
      rg[b][1] = *reinterpret_cast<const f32x4*>(src + b * 512 + 256);
    }
  };
  auto pool = [&](int c, const f32x4 (&rg)[TN][2]) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(xrow + c * GD_BK);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(xrow + c * GD_BK + 4);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32((t < 4 ? a0 : a1)[t & 3], rg[b][t >> 2][t & 3], acc[b], 0, 0, 0);
  };
  load_g(g0, 0);
#pragma unroll 1
  for (int c = 0; c < NCH; c += 2) {
    load_g(g1, c + 1);
    pool(c, g0);
    load_g(g0, c + 2);
    pool(c + 1, g1);
  }
  __syncthreads();  // the t tile is dead: the patches alias it
  float* stg = smem + wave * 1024;
  f32x4 csum[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    csum[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[b][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = it * 8 + r8;
      const long pix = m0 + wm0 + rr;
      if (pix >= p.P) continue;
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(&stg[rr * 32 + c4]);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = u[b][it][e] + 2.0f * xx[b][it][e] * s4[e];
      csum[b] += o;
      *reinterpret_cast<f32x4*>(p.out + pix * C + wn0 + b * 32 + c4) = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (p.cs_dx) {
    __syncthreads();
    float* part = smem;  // [2 wave rows][8 row groups][C]
#pragma unroll
    for (int b = 0; b < TN; ++b)
      *reinterpret_cast<f32x4*>(&part[(((wave >> 1) * 8 + r8) * C) + wn0 + b * 32 + c4]) = csum[b];
    __syncthreads();
    if (tid < C) {
      float a = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) a += part[k * C + tid];
      p.cs_dx[(long)blockIdx.x * C + tid] = a;
    }
  }
}

bool gd_al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int MODE>
int gdn_launch(const GdnParams& p, int C, hipStream_t s) {
  const unsigned grid = (unsigned)cdiv64(p.P, GD_BM);
  if (C == 192)
    hipLaunchKernelGGL((gdn_kernel<3, MODE>), dim3(grid), dim3(256), 0, s, p);
  else if (C == 128)
    hipLaunchKernelGGL((gdn_kernel<2, MODE>), dim3(grid), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((gdn_kernel<1, MODE>), dim3(grid), dim3(256), 0, s, p);
  return lic_check_launch();
}

}  // namespace

LIC_EXPORT int lic_gdn_supported(int32_t C) { return C == 64 || C == 128 || C == 192; }

LIC_EXPORT int lic_gdn_fwd(const float* x, const float* gammaT_packed, const float* beta_eff, const float* res,
                           float* y, float* norm, int64_t P, int32_t C, int32_t inverse, lic_stream_t stream) {
  if (!x || !gammaT_packed || !beta_eff || !y || P <= 0) return LIC_ERR_INVALID;  // (norm may be NULL: inference)
  if (!lic_gdn_supported(C)) return LIC_ERR_UNSUPPORTED;
  if (!gd_al16(x) || !gd_al16(gammaT_packed) || !gd_al16(beta_eff) || !gd_al16(y) || !gd_al16(norm) || !gd_al16(res))
    return LIC_ERR_INVALID;
  if (P > 0x7FFFFFFFL * 32) return LIC_ERR_UNSUPPORTED;
  GdnParams p{x, nullptr, nullptr, gammaT_packed, beta_eff, res, y, norm, nullptr, nullptr, (long)P, inverse};
  return gdn_launch<0>(p, C, (hipStream_t)stream);
}

LIC_EXPORT int64_t lic_gdn_bwd_partial_rows(int64_t P) { return P <= 0 ? 0 : cdiv64(P, GD_BM); }

LIC_EXPORT int lic_gdn_bwd(const float* g, const float* x, const float* norm, const float* gamma_packed, float* dx,
                           float* t, float* colsum_t_partial, float* colsum_dx_partial, int64_t P, int32_t C,
                           int32_t inverse, lic_stream_t stream) {
  if (!g || !x || !norm || !gamma_packed || !dx || !t || P <= 0) return LIC_ERR_INVALID;
  if (!lic_gdn_supported(C)) return LIC_ERR_UNSUPPORTED;
  if (!gd_al16(g) || !gd_al16(x) || !gd_al16(norm) || !gd_al16(gamma_packed) || !gd_al16(dx) || !gd_al16(t))
    return LIC_ERR_INVALID;
  if (P > 0x7FFFFFFFL * 32) return LIC_ERR_UNSUPPORTED;
  GdnParams p{x, g, norm, gamma_packed, nullptr, nullptr, dx, t, colsum_t_partial, colsum_dx_partial, (long)P, inverse};
  const char* e = getenv("LIC_GDN_BWD_REG");
  if (e && e[0] == '0') return gdn_launch<1>(p, C, (hipStream_t)stream);  // A/B: epilogue re-reads g, x, norm
  const unsigned grid = (unsigned)cdiv64(p.P, GD_BM);
  hipStream_t s = (hipStream_t)stream;
  if (C == 192)
    hipLaunchKernelGGL((gdn_bwd_reg_kernel<3>), dim3(grid), dim3(256), 0, s, p);
  else if (C == 128)
    hipLaunchKernelGGL((gdn_bwd_reg_kernel<2>), dim3(grid), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((gdn_bwd_reg_kernel<1>), dim3(grid), dim3(256), 0, s, p);
  return lic_check_launch();
}
