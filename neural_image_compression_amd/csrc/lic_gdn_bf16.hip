// GDN / IGDN backward in bf16 storage as ONE sweep (compressai GDN at Components.py:11-44; definition SURVEY.md
// Appendix B):   t = dL/dnorm(g, x, norm),   dx = g * norm^-1/2 + 2 x (t . gamma_eff)      (IGDN: norm^+1/2)
//
// The two-launch route (lic_gdn_dnorm_bf16, then lic_igemm_bf16 with the GDN_BWD epilogue) moves g, x and norm
// through HBM twice and t three times: 1.2 GB at 128^2 x 32 x 128 channels for a C x C contraction of 17 GFLOP -- the
// launches are memory-bound, not matrix-bound.  Here a wave owns 32 pixels x all C channels, reads g, x, norm ONCE
// (16 bytes per lane: pixel = lane & 31, 8 consecutive channels per 16-channel group and lane half), forms t in
// registers, turns its packed pairs into the MFMA B operand with one v_permlane32_swap per dword pair (the K order
// of lic_pack_weight_bf16_kperm: the inverse of the fused conv+GDN kernels' store swap), contracts it against
// gamma_eff (staged once per workgroup in LDS, kperm-packed) with the operands swapped -- the accumulators hold the
// transposed tile -- swaps the pool back to the loaded layout and finishes element-wise: 0.67 GB, no LDS round trip
// of the activations, t and dx written with 16-byte stores.  Rounding points are the two-launch route's (t to bf16,
// fp32 pool of bf16 operands, fp32 epilogue on the bf16 inputs); the pool sums each group of 16 channels in another
// order, so dx agrees to fp32 / one-bf16-ulp rounding, not bitwise.
#include "lic_common.h"

typedef __bf16 gb_bf16;
typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gb_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gb_u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct GdnBwdHParams {
  const gb_bf16* g;
  const gb_bf16* x;
  const gb_bf16* norm;
  const gb_bf16* gamma;  // gamma_eff packed by lic_pack_weight_bf16_kperm(taps 1, K = C (norm index), N = C, s_k = C, s_n = 1)
  gb_bf16* dx;
  gb_bf16* t;
  const gb_bf16* gammaT;  // RN: gamma_eff^T packed by lic_pack_weight_bf16_kperm(K = C (x index), N = C, s_k = 1, s_n = C)
  const float* beta;      // RN: beta_eff [C]
  float* cs_t;   // optional: [gridDim.x][C] per-workgroup column sums of t (-> d beta) ...
  float* cs_dx;  // ... and of dx (-> the d bias of the convolution in front); both or neither
  long P;
  long cs_rows;  // rows of cs_t / cs_dx
  int inverse;
};

// CS: also the column sums of t and dx.  d beta and the convolution's d bias are the column sums of two tensors this kernel
// has just produced; a separate launch re-read both (2 x 134 MB at 128^2 x 32 x 128 channels, 58-64 us) for 256 numbers.
// Here a wave parks its 32 x C tile of t (then of dx) in LDS as written -- rows 16 bytes apart from a multiple of 256 so
// that the 16-byte row writes spread over the banks -- and every lane sums one channel PAIR down the 32 rows; the sums
// stay in two registers per tensor across the workgroup's tiles and leave as one [C] row per workgroup for the pass's
// batched reduction (lic_reduce_batch, COLUMNS).
// RN: the pool norm = beta_eff + x^2 . gamma_eff^T is RECOMPUTED here instead of read (p.norm is not touched): the forward
// pass then writes two tensors per GDN layer (conv output, y) instead of three and this kernel reads two (g, x) instead of
// three -- the second C x C contraction costs a memory-bound kernel nothing.  The recomputed norm is rounded to bf16 like
// the stored one was, so everything downstream is the same arithmetic.
template <int NT4, bool CS = false, bool RN = false>  // C / 32
__global__ __launch_bounds__(256, NT4 <= 2 ? 2 : 1) void gdn_bwd_bf16_kernel(const GdnBwdHParams p) {
  constexpr int C = 32 * NT4, NG = C / 16;
  constexpr int CSLD = C / 2 + 4;   // dwords per parked row
  __shared__ __attribute__((aligned(16))) gb_bf16 gam[C * C];  // [chunk = C/32][tile = C/32][2][64 lanes][8]
  __shared__ __attribute__((aligned(16))) gb_bf16 gamT[RN ? C * C : 8];
  __shared__ __attribute__((aligned(16))) float s_beta[RN ? C : 4];
  __shared__ __attribute__((aligned(16))) unsigned park[CS ? 4 * 32 * CSLD : 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  float cst[2] = {0.f, 0.f}, csd[2] = {0.f, 0.f};   // column sums of channels 2 lane, 2 lane + 1 (lanes < C / 2)
  unsigned* mypark = park + (CS ? wave * 32 * CSLD : 0);
  auto colsum_tile = [&](float (&accum)[2]) {   // after the wave's lanes wrote their rows
    __builtin_amdgcn_wave_barrier();
    if (lane < C / 2) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const unsigned u = mypark[r * CSLD + lane];
        a0 += __builtin_bit_cast(float, u << 16);
        a1 += __builtin_bit_cast(float, u & 0xffff0000u);
      }
      accum[0] += a0;
      accum[1] += a1;
    }
    __builtin_amdgcn_wave_barrier();
  };
  // gamma panel: C*C*2 bytes, 16 bytes per thread per pass
#pragma unroll
  for (int i = 0; i < C * C / 8 / 256; ++i)
    reinterpret_cast<gb_u32x4*>(gam)[i * 256 + tid] = reinterpret_cast<const gb_u32x4*>(p.gamma)[i * 256 + tid];
  if (RN) {
#pragma unroll
    for (int i = 0; i < C * C / 8 / 256; ++i)
      reinterpret_cast<gb_u32x4*>(gamT)[i * 256 + tid] = reinterpret_cast<const gb_u32x4*>(p.gammaT)[i * 256 + tid];
    if (tid < C) s_beta[tid] = p.beta[tid];
  }
  __syncthreads();

  const long ntile = (p.P + 127) / 128;
  // RN: the NEXT tile's g and x are in flight while the current one is computed (a second register set, copied over at
  // the top of the loop) -- with the pool recomputed the sweep has more arithmetic between its loads and its stores, and a
  // wave that computes has nothing in flight
  gb_u32x4 gN[RN ? NG : 1], xN[RN ? NG : 1];
  auto prefetch = [&](long tile) {
    const long row = tile * 128 + wave * 32 + li;
    const long off = (row < p.P ? row : 0) * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      gN[RN ? s : 0] = *reinterpret_cast<const gb_u32x4*>(p.g + off + 16 * s);
      xN[RN ? s : 0] = *reinterpret_cast<const gb_u32x4*>(p.x + off + 16 * s);
    }
  };
  if (RN && (long)blockIdx.x < ntile) prefetch(blockIdx.x);
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    // (without the column sums' wave barriers nothing stops hipcc from hoisting BOTH panels' fragments out of the loop)
    if (RN && !CS) asm volatile("" ::: "memory");
    const long row = tile * 128 + wave * 32 + li;
    const bool rok = row < p.P;
    const long off = (rok ? row : 0) * C + 8 * lh;
    // ---- one sweep: 8 consecutive channels of group s for this lane's pixel, all three streams in flight at once
    gb_u32x4 gq[NG], xq[NG], nq[NG];
    if (RN) {
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        gq[s] = gN[RN ? s : 0];
        xq[s] = xN[RN ? s : 0];
      }
      if (tile + gridDim.x < ntile) prefetch(tile + gridDim.x);
    } else {
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        gq[s] = *reinterpret_cast<const gb_u32x4*>(p.g + off + 16 * s);
        xq[s] = *reinterpret_cast<const gb_u32x4*>(p.x + off + 16 * s);
        nq[s] = *reinterpret_cast<const gb_u32x4*>(p.norm + off + 16 * s);
      }
    }
    auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
    auto hi = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
    auto pack2 = [](float a, float b) {
      const gb_f32x2 v = {a, b};
      return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gb_bf16x2));
    };
    // a transposed accumulator tile back in the loaded layout: the accumulator of channel tile bo holds, for this lane's
    // pixel, channels 8 gg + 4 lh + {0..3} (gg = 0..3); group s = 2 bo + (gg >> 1) wants channels 8 lh + {0..7}
    auto unswap = [&](const f32x16& a16, int s, float (&pl)[8]) {
      const int g0 = 2 * (s & 1);   // registers 4 g0 .. 4 g0 + 7
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // (by value first: __builtin_bit_cast on a vector-element lvalue reads element 0 of the vector with this clang)
        const float q0 = a16[4 * g0 + e], q1 = a16[4 * g0 + 4 + e];
        const gb_u32x2 sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, q0),
                                                             __builtin_bit_cast(unsigned, q1), false, false);
        // low lanes: (own first quad, partner's first quad) = channels e, 4 + e; high lanes: (partner's second quad,
        // own second quad) = channels 8 + e, 12 + e
        const unsigned w0 = sw[0], w1 = sw[1];
        pl[e] = __builtin_bit_cast(float, w0);
        pl[4 + e] = __builtin_bit_cast(float, w1);
      }
    };
    if (RN) {
      // ---- norm = beta_eff + x^2 . gamma_eff^T, as the forward pass forms it: x^2 rounded to bf16, fp32 pool, bf16 result
      f32x16 nacc[NT4];
#pragma unroll
      for (int bo = 0; bo < NT4; ++bo)
#pragma unroll
        for (int r = 0; r < 16; ++r) nacc[bo][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        gb_u32x4 sq;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float x0 = lo(xq[s][d]), x1 = hi(xq[s][d]);
          sq[d] = pack2(x0 * x0, x1 * x1);
        }
        const gb_u32x2 a = __builtin_amdgcn_permlane32_swap(sq[0], sq[2], false, false);
        const gb_u32x2 b = __builtin_amdgcn_permlane32_swap(sq[1], sq[3], false, false);
        const gb_bf16x8 b2 = __builtin_bit_cast(gb_bf16x8, gb_u32x4{a[0], b[0], a[1], b[1]});
#pragma unroll
        for (int bo = 0; bo < NT4; ++bo) {
          const gb_bf16x8 a2 = *reinterpret_cast<const gb_bf16x8*>(gamT + (((s >> 1) * NT4 + bo) * 2 + (s & 1)) * 512 + lane * 8);
          nacc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, nacc[bo], 0, 0, 0);
        }
      }
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        float pl[8];
        unswap(nacc[s >> 1], s, pl);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_beta + 16 * s + 8 * lh);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(s_beta + 16 * s + 8 * lh + 4);
        nq[s] = gb_u32x4{pack2(pl[0] + b0[0], pl[1] + b0[1]), pack2(pl[2] + b0[2], pl[3] + b0[3]),
                         pack2(pl[4] + b1[0], pl[5] + b1[1]), pack2(pl[6] + b1[2], pl[7] + b1[3])};
      }
    }
    // ---- t = dL/dnorm, rounded to bf16 (what the d-gamma / d-beta launches read); its kperm B operand
    gb_u32x4 tb[NG];
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      gb_u32x4 tq;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float n0 = lo(nq[s][d]), n1 = hi(nq[s][d]);
        const float r0 = __builtin_amdgcn_rsqf(n0), r1 = __builtin_amdgcn_rsqf(n1);
        const float gx0 = lo(gq[s][d]) * lo(xq[s][d]), gx1 = hi(gq[s][d]) * hi(xq[s][d]);
        const float t0 = p.inverse ? 0.5f * gx0 * r0 : -0.5f * gx0 * r0 * (r0 * r0);
        const float t1 = p.inverse ? 0.5f * gx1 * r1 : -0.5f * gx1 * r1 * (r1 * r1);
        tq[d] = pack2(t0, t1);
      }
      if (rok) *reinterpret_cast<gb_u32x4*>(p.t + off + 16 * s) = tq;
      if (CS) *reinterpret_cast<gb_u32x4*>(mypark + li * CSLD + 8 * s + 4 * lh) = rok ? tq : gb_u32x4{0u, 0u, 0u, 0u};
      // lanes li / li + 32 hold channels 0..7 / 8..15 of the group; the kperm operand wants {0..3, 8..11} / {4..7, 12..15}
      const gb_u32x2 a = __builtin_amdgcn_permlane32_swap(tq[0], tq[2], false, false);
      const gb_u32x2 b = __builtin_amdgcn_permlane32_swap(tq[1], tq[3], false, false);
      tb[s] = gb_u32x4{a[0], b[0], a[1], b[1]};
    }
    if (CS) colsum_tile(cst);
    // ---- pool^T[co][pixel] = sum_ci gamma_eff[ci][co] t[pixel][ci]   (A = gamma from LDS, B = t from registers)
    f32x16 acc[NT4];
#pragma unroll
    for (int bo = 0; bo < NT4; ++bo)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[bo][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const gb_bf16x8 b2 = __builtin_bit_cast(gb_bf16x8, tb[s]);
#pragma unroll
      for (int bo = 0; bo < NT4; ++bo) {
        const gb_bf16x8 a2 = *reinterpret_cast<const gb_bf16x8*>(gam + (((s >> 1) * NT4 + bo) * 2 + (s & 1)) * 512 + lane * 8);
        acc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[bo], 0, 0, 0);
      }
    }
    // ---- dx = g f + 2 x pool, f = norm^-1/2 (IGDN: ^+1/2); the pool goes back to the loaded layout first (unswap)
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      float pl[8];
      unswap(acc[s >> 1], s, pl);
      gb_u32x4 dq;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float n0 = lo(nq[s][d]), n1 = hi(nq[s][d]);
        const float f0 = p.inverse ? __builtin_amdgcn_sqrtf(n0) : __builtin_amdgcn_rsqf(n0);
        const float f1 = p.inverse ? __builtin_amdgcn_sqrtf(n1) : __builtin_amdgcn_rsqf(n1);
        const float d0 = __builtin_fmaf(2.0f * lo(xq[s][d]), pl[2 * d], lo(gq[s][d]) * f0);
        const float d1 = __builtin_fmaf(2.0f * hi(xq[s][d]), pl[2 * d + 1], hi(gq[s][d]) * f1);
        dq[d] = pack2(d0, d1);
      }
      if (rok) *reinterpret_cast<gb_u32x4*>(p.dx + off + 16 * s) = dq;
      if (CS) *reinterpret_cast<gb_u32x4*>(mypark + li * CSLD + 8 * s + 4 * lh) = rok ? dq : gb_u32x4{0u, 0u, 0u, 0u};
    }
    if (CS) colsum_tile(csd);
  }
  if (CS) {   // the four waves' sums -> one row per workgroup (fixed order)
    __syncthreads();
    float* red = reinterpret_cast<float*>(park);   // [2][4][C]
    if (lane < C / 2) {
      red[(0 * 4 + wave) * C + 2 * lane] = cst[0];
      red[(0 * 4 + wave) * C + 2 * lane + 1] = cst[1];
      red[(1 * 4 + wave) * C + 2 * lane] = csd[0];
      red[(1 * 4 + wave) * C + 2 * lane + 1] = csd[1];
    }
    __syncthreads();
    if (tid < 2 * C) {
      const int which = tid / C, c = tid - which * C;
      const float* q = red + which * 4 * C + c;
      const float v = (q[0] + q[C]) + (q[2 * C] + q[3 * C]);
      float* dstp = which ? p.cs_dx : p.cs_t;
      dstp[(long)blockIdx.x * C + c] = v;
      // (the caller sized the buffers for lic_gdn_bwd_bf16_partial_rows(P) rows: a smaller grid zeroes the rest)
      for (long r = (long)blockIdx.x + gridDim.x; r < p.cs_rows; r += gridDim.x) dstp[r * C + c] = 0.0f;
    }
  }
}

}  // namespace

LIC_EXPORT int lic_gdn_bwd_bf16_supported(int32_t C) { return C == 64 || C == 128; }

// g, x, norm, dx, t: dense bf16 [P][C]; gamma_packed: lic_pack_weight_bf16_kperm(gamma_eff, taps 1, K = C, N = C,
// s_k = C, s_n = 1)
static unsigned gdn_bwd_bf16_grid(int64_t P) {
  const long ntile = (P + 127) / 128;
  return (unsigned)(ntile < 2048 ? ntile : 2048);
}
// rows of the optional per-workgroup column-sum outputs of lic_gdn_bwd_bf16
LIC_EXPORT int64_t lic_gdn_bwd_bf16_partial_rows(int64_t P) { return P <= 0 ? 0 : (int64_t)gdn_bwd_bf16_grid(P); }

static int gdn_bwd_bf16_run(const GdnBwdHParams& p, int C, bool rn, hipStream_t s) {
  unsigned grid = gdn_bwd_bf16_grid(p.P);
  if (rn) {
    // Fewer, longer-lived workgroups let the prefetch work between more tiles: alone the 128^2 launch runs 186 / 167 / 155 /
    // 143 us at 2048 / 1024 / 512 / 256 workgroups -- but the config-3 step does not move (9777-9814 / 9691-9739 / 9732-9763 /
    // 9768-9770 img/s): beside the other stream's launches a 256-workgroup kernel with 100 KB of LDS owns its CUs for its
    // whole life.  The default stays at the full grid; LIC_GDN_BWD_RN_GRID caps it.
    static const long cap = [] {
      const char* e = getenv("LIC_GDN_BWD_RN_GRID");
      const long v = e ? atol(e) : 2048;
      return v >= 64 ? v : 2048;
    }();
    if ((long)grid > cap) grid = (unsigned)cap;
  }
#define LIC_GB(nt, cs, rnv) hipLaunchKernelGGL((gdn_bwd_bf16_kernel<nt, cs, rnv>), dim3(grid), dim3(256), 0, s, p)
  const bool cs = p.cs_t != nullptr;
  if (C == 64) {
    if (rn) { if (cs) LIC_GB(2, true, true); else LIC_GB(2, false, true); }
    else { if (cs) LIC_GB(2, true, false); else LIC_GB(2, false, false); }
  } else {
    if (rn) { if (cs) LIC_GB(4, true, true); else LIC_GB(4, false, true); }
    else { if (cs) LIC_GB(4, true, false); else LIC_GB(4, false, false); }
  }
#undef LIC_GB
  return lic_check_launch();
}

LIC_EXPORT int lic_gdn_bwd_bf16(const void* g, const void* x, const void* norm, const void* gamma_packed, void* dx,
                                void* t, float* colsum_t_partial, float* colsum_dx_partial, int64_t P, int32_t C,
                                int32_t inverse, lic_stream_t stream) {
  if (!g || !x || !norm || !gamma_packed || !dx || !t || P <= 0) return LIC_ERR_INVALID;
  if ((colsum_t_partial == nullptr) != (colsum_dx_partial == nullptr)) return LIC_ERR_INVALID;
  if (!lic_gdn_bwd_bf16_supported(C)) return LIC_ERR_UNSUPPORTED;
  for (const void* q : {g, x, norm, gamma_packed, (const void*)dx, (const void*)t})
    if (reinterpret_cast<uintptr_t>(q) & 15) return LIC_ERR_INVALID;
  GdnBwdHParams p{(const gb_bf16*)g, (const gb_bf16*)x, (const gb_bf16*)norm, (const gb_bf16*)gamma_packed,
                  (gb_bf16*)dx, (gb_bf16*)t, nullptr, nullptr, colsum_t_partial, colsum_dx_partial, (long)P,
                  (long)gdn_bwd_bf16_grid(P), inverse ? 1 : 0};
  return gdn_bwd_bf16_run(p, C, false, (hipStream_t)stream);
}

// ... with the pool recomputed instead of read: norm = beta_eff + x^2 . gamma_eff^T is formed here as the forward pass
// forms it (x^2 and the result rounded to bf16), so the forward pass need not store it.  gammaT_packed =
// lic_pack_weight_bf16_kperm(gamma_eff, taps 1, K = C, N = C, s_k = 1, s_n = C) (the fused forward kernels' operand).
LIC_EXPORT int lic_gdn_bwd_bf16_recompute(const void* g, const void* x, const void* gamma_packed, const void* gammaT_packed,
                                          const float* beta_eff, void* dx, void* t, float* colsum_t_partial,
                                          float* colsum_dx_partial, int64_t P, int32_t C, int32_t inverse,
                                          lic_stream_t stream) {
  if (!g || !x || !gamma_packed || !gammaT_packed || !beta_eff || !dx || !t || P <= 0) return LIC_ERR_INVALID;
  if ((colsum_t_partial == nullptr) != (colsum_dx_partial == nullptr)) return LIC_ERR_INVALID;
  if (!lic_gdn_bwd_bf16_supported(C)) return LIC_ERR_UNSUPPORTED;
  for (const void* q : {g, x, gamma_packed, gammaT_packed, (const void*)beta_eff, (const void*)dx, (const void*)t})
    if (reinterpret_cast<uintptr_t>(q) & 15) return LIC_ERR_INVALID;
  GdnBwdHParams p{(const gb_bf16*)g, (const gb_bf16*)x, nullptr, (const gb_bf16*)gamma_packed, (gb_bf16*)dx, (gb_bf16*)t,
                  (const gb_bf16*)gammaT_packed, beta_eff, colsum_t_partial, colsum_dx_partial, (long)P,
                  (long)gdn_bwd_bf16_grid(P), inverse ? 1 : 0};
  return gdn_bwd_bf16_run(p, C, true, (hipStream_t)stream);
}
