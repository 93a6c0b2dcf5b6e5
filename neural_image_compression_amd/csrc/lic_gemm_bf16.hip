// bf16-storage / fp32-accumulate variants of the implicit-GEMM kernels (BASELINE config 3).
// Same structure as lic_gemm.hip -- A gathered through a double-buffered LDS tile, B (packed
// weights) read straight from L2 into MFMA operand registers, 2x2 waves, LDS-staged epilogue --
// on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate): a K chunk is 32 bf16 = the same 64 bytes
// per row as the fp32 kernel's 16 floats, so the memory-side instruction stream is identical and
// only the matrix work per chunk shrinks (2 MFMAs of 32 cycles per tile instead of 8 of 64).
#include "lic_common.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct FastDivB {
  unsigned m, s;
};
static FastDivB make_fastdivb(unsigned d) {
  FastDivB f;
  if (d == 0) d = 1;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.s = s;
  f.m = (unsigned)(((1ull << (31 + s)) + d - 1) / d);
  return f;
}
__device__ __forceinline__ int fdivb(int n, FastDivB f) {
  return (int)(((unsigned long long)(unsigned)n * f.m) >> (31 + f.s));
}

constexpr int HB_BK = 32;            // bf16 elements per K chunk
constexpr int HB_LDA = HB_BK + 8;    // LDS row pitch in bf16 (80 bytes)

struct IgemmHParams {
  const bf16_t* in;
  const bf16_t* w;  // packed [tap][cpt][Npad][32] bf16
  const float* bias;
  void* out;        // bf16 or fp32 (out_f32)
  bf16_t* out2;     // GDN norm (bf16)
  const bf16_t* aux;
  const bf16_t* aux2;
  const bf16_t* aux3;
  long in_ld, out_ld, out2_ld, aux_ld, aux2_ld, aux3_ld;
  int B, Hi, Wi, Cin, Ho, Wo, Cout;
  int kw, stride, pad, transposed, prologue, epilogue, out_f32;
  int cpt, Npad, nphase, MT, NT;
  int ntaps[4];
  int Hq[4], Wq[4];
  FastDivB dHW[4], dW[4];
  unsigned char taps[4][28];
};

__device__ __forceinline__ bf16x8 sq8(bf16x8 v) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = (float)v[e];
    o[e] = (bf16_t)(f * f);
  }
  return o;
}

template <int BM, int TN, bool FULLN>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IgemmHParams p) {
  constexpr int BN = 64 * TN;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32;
  constexpr int APASS = BM / 64;
  // A double buffer (bf16), later reused as the fp32 staging area of the epilogue (4 KiB per wave)
  constexpr int SA_BYTES = (2 * BM * HB_LDA * 2 > 4 * 4096) ? 2 * BM * HB_LDA * 2 : 4 * 4096;
  __shared__ __attribute__((aligned(16))) char smem_raw[SA_BYTES];
  bf16_t(*sA)[BM * HB_LDA] = reinterpret_cast<bf16_t(*)[BM * HB_LDA]>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int li = lane & 31, lh = lane >> 5;

  const int nwg = gridDim.x;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nt = wg % p.NT;
  const int kq = wg / p.NT;
  const int phase = (p.nphase == 4) ? ((kq + (kq >> 2) + (kq >> 4) + (kq >> 6) + (kq >> 8)) & 3) : 0;
  const int mt = kq / p.nphase;
  const int Hq = p.Hq[phase], Wq = p.Wq[phase];
  const int P = p.B * Hq * Wq;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= P) return;
  const int sph = (p.nphase > 1) ? p.stride : 1;
  const int py = (p.nphase > 1) ? phase / p.stride : 0;
  const int px = (p.nphase > 1) ? phase % p.stride : 0;

  const int a_c8 = (tid & 3) * 8;  // bf16 channel offset of this thread's 16-byte piece
  int a_base[APASS], a_hy[APASS], a_wx[APASS];
  bool a_ok[APASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int prow = m0 + (tid >> 2) + 64 * j;
    a_ok[j] = prow < P;
    const int pr = a_ok[j] ? prow : 0;
    const int b = fdivb(pr, p.dHW[phase]);
    const int rem = pr - b * Hq * Wq;
    const int i = fdivb(rem, p.dW[phase]), jj = rem - i * Wq;
    const int oy = i * sph + py, ox = jj * sph + px;
    a_base[j] = b * p.Hi * p.Wi;
    if (p.transposed) {
      a_hy[j] = oy + p.pad;
      a_wx[j] = ox + p.pad;
    } else {
      a_hy[j] = oy * p.stride - p.pad;
      a_wx[j] = ox * p.stride - p.pad;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  int n_live = 0;
#pragma unroll
  for (int b = 0; b < TN; ++b) n_live += ((n0 + wn0 + b * 32) < p.Npad) ? 1 : 0;

  const int ntaps = p.ntaps[phase];
  const int nchunks = ntaps * p.cpt;
  bf16x8 ra[APASS];
  bool ra_ok[APASS];
  const bf16_t* wlane = p.w + ((long)(n0 + wn0 + li) * HB_BK + lh * 8);

  const int sgn = p.transposed ? -1 : 1;
  const int sh = (p.transposed && p.stride == 2) ? 1 : 0;
  const int last_tap = ntaps - 1, last_cb = p.cpt - 1;
  auto load_a = [&](int tapi, int cb) {
    const bool past = tapi > last_tap;
    const int tap = p.taps[phase][past ? last_tap : tapi];
    const int r = tap / p.kw, s = tap - r * p.kw;
    const int ci = (past ? last_cb : cb) * HB_BK + a_c8;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int nh = a_hy[j] + sgn * r, nw = a_wx[j] + sgn * s;
      const int ih = nh >> sh, iw = nw >> sh;
      const bool ok = a_ok[j] && nh >= 0 && nw >= 0 && ih < p.Hi && iw < p.Wi && ci < p.Cin;
      const int okm = -(int)ok;
      const int pixi = (a_base[j] + ih * p.Wi + iw) & okm;
      const int cc = ci & okm;
      ra[j] = *reinterpret_cast<const bf16x8*>(p.in + ((long)pixi * p.in_ld + cc));
      ra_ok[j] = ok;
    }
  };
  const bool sq = p.prologue == 1;
  auto store_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      bf16x8 v = ra[j];
      if (!ra_ok[j]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.0f;
      }
      if (sq) v = sq8(v);
      *reinterpret_cast<bf16x8*>(&sA[buf][((tid >> 2) + 64 * j) * HB_LDA + a_c8]) = v;
    }
  };
  int b_off[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) b_off[b] = (b < n_live ? b : (n_live > 0 ? n_live - 1 : 0)) * 32 * HB_BK;
  if (n_live == 0) wlane = p.w + lh * 8;
  auto load_b = [&](bf16x8 (&rb)[TN][2], int tapi, int cb) {
    const bool past = tapi > last_tap;
    const int tap = p.taps[phase][past ? last_tap : tapi];
    const bf16_t* src = wlane + ((long)tap * p.cpt + (past ? last_cb : cb)) * p.Npad * HB_BK;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      rb[b][0] = *reinterpret_cast<const bf16x8*>(src + b_off[b]);
      rb[b][1] = *reinterpret_cast<const bf16x8*>(src + b_off[b] + 16);
    }
  };
  auto compute = [&](int buf, const bf16x8 (&rb)[TN][2]) {
    bf16x8 af[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const bf16_t* src = &sA[buf][(wm0 + a * 32 + li) * HB_LDA + lh * 8];
      af[a][0] = *reinterpret_cast<const bf16x8*>(src);
      af[a][1] = *reinterpret_cast<const bf16x8*>(src + 16);
    }
    if constexpr (FULLN) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int a = 0; a < TM; ++a)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][q], rb[b][q], acc[a][b], 0, 0, 0);
    } else {
#pragma unroll
      for (int b = 0; b < TN; ++b)
        if (b < n_live) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int a = 0; a < TM; ++a)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][q], rb[b][q], acc[a][b], 0, 0, 0);
        }
    }
  };

  int l_tap = 0, l_cb = 0;
  auto advance = [&]() {
    if (++l_cb == p.cpt) {
      l_cb = 0;
      ++l_tap;
    }
  };
  bf16x8 rb0[TN][2], rb1[TN][2];
  if (nchunks > 0) {
    load_a(0, 0);
    load_b(rb0, 0, 0);
    store_a(0);
    __syncthreads();
    advance();
    load_a(l_tap, l_cb);
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
      load_b(rb1, l_tap, l_cb);
      store_a(1);
      advance();
      load_a(l_tap, l_cb);
      compute(0, rb0);
      __syncthreads();
      load_b(rb0, l_tap, l_cb);
      store_a(0);
      advance();
      load_a(l_tap, l_cb);
      compute(1, rb1);
      __syncthreads();
    }
    if (c < nchunks) {
      compute(0, rb0);
      __syncthreads();
    }
  }

  // ---- epilogue: stage each 32x32 fp32 tile through LDS; a lane then owns 8 consecutive
  // channels of a row (16-byte bf16 accesses; fp32 output writes two 16-byte halves) ------------
  const int epi = p.epilogue;
  float* stg = reinterpret_cast<float*>(smem_raw) + wave * 1024;
  const int c8 = (lane & 3) * 8, r16 = lane >> 2;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[a][b][r];
      __syncthreads();
      const int col = n0 + wn0 + b * 32 + c8;
      if (col >= p.Cout) continue;
      float bias8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias8[e] = p.bias ? p.bias[col + e] : 0.0f;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int rr = it * 16 + r16;
        const int prow = m0 + wm0 + a * 32 + rr;
        if (prow >= P) continue;
        long opix = prow;
        if (p.nphase > 1) {
          const int bb = fdivb(prow, p.dHW[phase]);
          const int rem = prow - bb * Hq * Wq;
          const int i = fdivb(rem, p.dW[phase]), jj = rem - i * Wq;
          opix = ((long)bb * p.Ho + i * sph + py) * p.Wo + jj * sph + px;
        }
        float v[8];
        {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(&stg[rr * 32 + c8]);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(&stg[rr * 32 + c8 + 4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = v0[e] + bias8[e];
            v[4 + e] = v1[e] + bias8[4 + e];
          }
        }
        if (epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN) {
          if (p.out2) {
            bf16x8 nb;
#pragma unroll
            for (int e = 0; e < 8; ++e) nb[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x8*>(p.out2 + opix * p.out2_ld + col) = nb;
          }
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(p.aux + opix * p.aux_ld + col);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = (float)x[e] * ((epi == LIC_EPI_GDN) ? __builtin_amdgcn_rsqf(v[e]) : __builtin_amdgcn_sqrtf(v[e]));
        } else if (epi == LIC_EPI_GDN_BWD || epi == LIC_EPI_IGDN_BWD) {
          const bf16x8 n = *reinterpret_cast<const bf16x8*>(p.aux3 + opix * p.aux3_ld + col);
          const bf16x8 g = *reinterpret_cast<const bf16x8*>(p.aux + opix * p.aux_ld + col);
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(p.aux2 + opix * p.aux2_ld + col);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float nf = (float)n[e];
            const float f = (epi == LIC_EPI_GDN_BWD) ? __builtin_amdgcn_rsqf(nf) : __builtin_amdgcn_sqrtf(nf);
            v[e] = (float)g[e] * f + 2.0f * (float)x[e] * v[e];
          }
        }
        if (p.out_f32) {
          float* o = reinterpret_cast<float*>(p.out) + opix * p.out_ld + col;
          f32x4 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o0[e] = v[e];
            o1[e] = v[4 + e];
          }
          *reinterpret_cast<f32x4*>(o) = o0;
          *reinterpret_cast<f32x4*>(o + 4) = o1;
        } else {
          bf16x8 ob;
#pragma unroll
          for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.out) + opix * p.out_ld + col) = ob;
        }
      }
    }
}

static bool al16h(const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// ---- weight packing to bf16: dst[tap][chunk][n][32], zero padded ----------------------------------
__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float* src, bf16_t* dst, int taps, int K,
                                                               int N, int cpt, int Npad, long s_tap, long s_k,
                                                               long s_n) {
  const long total = (long)taps * cpt * Npad * HB_BK;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int kk = (int)(i & (HB_BK - 1));
    long t = i >> 5;
    const int n = (int)(t % Npad);
    t /= Npad;
    const int cb = (int)(t % cpt);
    const int tap = (int)(t / cpt);
    const int k = cb * HB_BK + kk;
    dst[i] = (bf16_t)((k < K && n < N) ? src[tap * s_tap + k * s_k + n * s_n] : 0.0f);
  }
}
LIC_EXPORT int64_t lic_packed_weight_bf16_elems(int32_t taps, int32_t K, int32_t N) {
  if (taps <= 0 || K <= 0 || N <= 0) return 0;
  return (int64_t)taps * ((K + HB_BK - 1) / HB_BK) * (((N + 31) / 32) * 32) * HB_BK;
}
LIC_EXPORT int lic_pack_weight_bf16(const float* src, void* dst, int32_t taps, int32_t K, int32_t N,
                                    int64_t s_tap, int64_t s_k, int64_t s_n, lic_stream_t stream) {
  if (!src || !dst || taps <= 0 || K <= 0 || N <= 0) return LIC_ERR_INVALID;
  const int cpt = (K + HB_BK - 1) / HB_BK, Npad = ((N + 31) / 32) * 32;
  const long total = (long)taps * cpt * Npad * HB_BK;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, taps, K, N, cpt, Npad, (long)s_tap, (long)s_k, (long)s_n);
  return lic_check_launch();
}

// d uses the lic_igemm_desc layout; activation / aux / out2 pointers are bf16, `w` is the bf16
// packed weight, bias is fp32; out is bf16 unless out_f32.
LIC_EXPORT int lic_igemm_bf16(const lic_igemm_desc* d, int32_t out_f32, lic_stream_t stream) {
  if (!d || !d->in || !d->w || !d->out) return LIC_ERR_INVALID;
  if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 ||
      d->kh <= 0 || d->kw <= 0)
    return LIC_ERR_INVALID;
  if (d->kh * d->kw > 28 || d->stride < 1 || d->stride > 2) return LIC_ERR_UNSUPPORTED;
  const int epi = d->epilogue;
  if (!(epi == LIC_EPI_NONE || epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN || epi == LIC_EPI_GDN_BWD ||
        epi == LIC_EPI_IGDN_BWD) || d->res)
    return LIC_ERR_UNSUPPORTED;
  if ((epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN) && !d->aux) return LIC_ERR_INVALID;
  if ((epi == LIC_EPI_GDN_BWD || epi == LIC_EPI_IGDN_BWD) && (!d->aux || !d->aux2 || !d->aux3)) return LIC_ERR_INVALID;
  // bf16 path: 16-byte pieces everywhere -> channel counts and pitches multiples of 8
  if (d->Cin % 8 || d->Cout % 8 || d->in_ld % 8 || d->out_ld % 8 || (d->out2 && d->out2_ld % 8) ||
      (d->aux && d->aux_ld % 8) || (d->aux2 && d->aux2_ld % 8) || (d->aux3 && d->aux3_ld % 8))
    return LIC_ERR_UNSUPPORTED;
  if (!al16h(d->in) || !al16h(d->w) || !al16h(d->out) || !al16h(d->out2) || !al16h(d->aux) || !al16h(d->aux2) ||
      !al16h(d->aux3))
    return LIC_ERR_INVALID;
  IgemmHParams p;
  p.in = (const bf16_t*)d->in;
  p.w = (const bf16_t*)d->w;
  p.bias = d->bias;
  p.out = d->out;
  p.out2 = (bf16_t*)d->out2;
  p.aux = (const bf16_t*)d->aux;
  p.aux2 = (const bf16_t*)d->aux2;
  p.aux3 = (const bf16_t*)d->aux3;
  p.in_ld = d->in_ld;
  p.out_ld = d->out_ld;
  p.out2_ld = d->out2_ld;
  p.aux_ld = d->aux_ld;
  p.aux2_ld = d->aux2_ld;
  p.aux3_ld = d->aux3_ld;
  p.B = d->B;
  p.Hi = d->Hi;
  p.Wi = d->Wi;
  p.Cin = d->Cin;
  p.Ho = d->Ho;
  p.Wo = d->Wo;
  p.Cout = d->Cout;
  p.kw = d->kw;
  p.stride = d->stride;
  p.pad = d->pad;
  p.transposed = d->transposed ? 1 : 0;
  p.prologue = d->prologue;
  p.epilogue = epi;
  p.out_f32 = out_f32 ? 1 : 0;
  p.cpt = (d->Cin + HB_BK - 1) / HB_BK;
  p.Npad = ((d->Cout + 31) / 32) * 32;
  const uint32_t mask = d->tap_mask ? d->tap_mask : 0xFFFFFFFFu;
  p.nphase = (p.transposed && d->stride > 1) ? d->stride * d->stride : 1;
  long maxP = 0;
  for (int ph = 0; ph < 4; ++ph) {
    p.ntaps[ph] = 0;
    p.Hq[ph] = p.Wq[ph] = 0;
    p.dHW[ph] = p.dW[ph] = make_fastdivb(1);
  }
  for (int ph = 0; ph < p.nphase; ++ph) {
    const int py = (p.nphase > 1) ? ph / d->stride : 0, px = (p.nphase > 1) ? ph % d->stride : 0;
    const int st = (p.nphase > 1) ? d->stride : 1;
    p.Hq[ph] = (d->Ho - py + st - 1) / st;
    p.Wq[ph] = (d->Wo - px + st - 1) / st;
    if (p.Hq[ph] < 0) p.Hq[ph] = 0;
    if (p.Wq[ph] < 0) p.Wq[ph] = 0;
    const long Pp = (long)d->B * p.Hq[ph] * p.Wq[ph];
    p.dHW[ph] = make_fastdivb((unsigned)(p.Hq[ph] * p.Wq[ph]));
    p.dW[ph] = make_fastdivb((unsigned)p.Wq[ph]);
    if (Pp > maxP) maxP = Pp;
    int n = 0;
    for (int r = 0; r < d->kh; ++r)
      for (int s = 0; s < d->kw; ++s) {
        const int t = r * d->kw + s;
        if (!((mask >> t) & 1u)) continue;
        if (p.nphase > 1)
          if (((py + d->pad - r) % d->stride) != 0 || ((px + d->pad - s) % d->stride) != 0) continue;
        p.taps[ph][n++] = (unsigned char)t;
      }
    p.ntaps[ph] = n;
  }
  if (maxP <= 0) return LIC_OK;
  if (maxP > 0x7FFFFFFFL / 2) return LIC_ERR_UNSUPPORTED;
  static const int cand[6][2] = {{128, 3}, {64, 3}, {128, 2}, {64, 2}, {128, 1}, {64, 1}};
  int best = 5;
  long best_wg = -1;
  bool found = false;
  for (int pass = 0; pass < 2 && !found; ++pass)
    for (int c = 0; c < 6; ++c) {
      const int bm = cand[c][0], tn = cand[c][1];
      if (p.Npad < 64 * tn && tn > 1 && p.Npad <= 64 * (tn - 1)) continue;
      if (pass == 0 && p.Npad % (64 * tn) != 0) continue;
      const long wgs = ((maxP + bm - 1) / bm) * ((p.Npad + 64 * tn - 1) / (64 * tn)) * p.nphase;
      if (wgs >= 512) {
        best = c;
        best_wg = wgs;
        found = true;
        break;
      }
      if (wgs > best_wg) {
        best = c;
        best_wg = wgs;
      }
    }
  const int BM = cand[best][0], TN = cand[best][1];
  p.NT = (p.Npad + 64 * TN - 1) / (64 * TN);
  p.MT = (int)((maxP + BM - 1) / BM);
  const long nwg = (long)p.MT * p.NT * p.nphase;
  if (nwg > 0x7FFFFFFFL) return LIC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)nwg), block(256);
  const bool full = (p.Npad % (64 * TN)) == 0;
#define LIC_IGEMMH_LAUNCH(bm, tn)                                                   \
  do {                                                                              \
    if (full)                                                                       \
      hipLaunchKernelGGL((igemm_bf16_kernel<bm, tn, true>), grid, block, 0, s, p);  \
    else                                                                            \
      hipLaunchKernelGGL((igemm_bf16_kernel<bm, tn, false>), grid, block, 0, s, p); \
  } while (0)
  if (BM == 128 && TN == 3)
    LIC_IGEMMH_LAUNCH(128, 3);
  else if (BM == 64 && TN == 3)
    LIC_IGEMMH_LAUNCH(64, 3);
  else if (BM == 128 && TN == 2)
    LIC_IGEMMH_LAUNCH(128, 2);
  else if (BM == 64 && TN == 2)
    LIC_IGEMMH_LAUNCH(64, 2);
  else if (BM == 128 && TN == 1)
    LIC_IGEMMH_LAUNCH(128, 1);
  else
    LIC_IGEMMH_LAUNCH(64, 1);
#undef LIC_IGEMMH_LAUNCH
  return lic_check_launch();
}

// ------------------------------------------------------------------------------------------------
// wgrad (bf16 operands, fp32 slabs): R[tap][m][n] = sum_pix A[pix][m] * B[pix][n].
// The MFMA wants 8 consecutive K (= pixels) per lane, but activations are pixel-major, so each
// 32-pixel chunk is transposed on its way into LDS: a thread loads 8 channels of one pixel (16 B)
// and scatters them as 2-byte stores into a channel-major tile [ch][32 pix (+8 pad)], whose
// 8-pixel groups are XOR-swizzled by (ch/8)%4 (8-way -> 2-way store conflicts); fragments are then
// plain ds_read_b128.
// ------------------------------------------------------------------------------------------------
struct WgHOperand {
  const bf16_t* ptr;
  long ld;
  int C;
  int gathered;
  int sq;
};
struct WgradHParams {
  WgHOperand row, col;
  float* slabs;
  int B, Hs, Ws, Hl, Wl;
  int kw, stride, pad, ntaps;
  int MTt, NTt;
  int chunks_per_split, nchunks;
  long Ps;
  FastDivB dHW, dW;
};
constexpr int WH_BK = 32;
constexpr int WH_LD = WH_BK + 8;

template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WgradHParams p) {
  constexpr int BMt = 64 * TM, BNt = 64 * TN;
  constexpr int WM = BMt / 2, WN = BNt / 2;
  __shared__ __attribute__((aligned(16))) bf16_t sA[2][BMt * WH_LD];
  __shared__ __attribute__((aligned(16))) bf16_t sB[2][BNt * WH_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware bijective remap: whole K splits per XCD (see wgrad_kernel in lic_gemm.hip)
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int tiles = p.MTt * p.NTt;
  const int tile = wg % tiles;
  wg /= tiles;
  const int tap = wg % p.ntaps, split = wg / p.ntaps;
  const int mt = tile / p.NTt, nt = tile - mt * p.NTt;
  const int m0 = mt * BMt, n0 = nt * BNt;
  const int r = tap / p.kw, s = tap - r * p.kw;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
  const int nloc = c_end - c_begin;
  int m_live = 0, n_live = 0;
#pragma unroll
  for (int a = 0; a < TM; ++a) m_live += ((m0 + wm0 + a * 32) < p.row.C) ? 1 : 0;
#pragma unroll
  for (int b = 0; b < TN; ++b) n_live += ((n0 + wn0 + b * 32) < p.col.C) ? 1 : 0;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;

  const int kr = tid >> 3, l8 = tid & 7;     // pixel row of the chunk, channel octet
  const int swz = (l8 & 3) * 8;              // store-side swizzle of this thread's channels
  const int kpos = kr ^ swz;                 // where pixel kr lands inside a channel row
  bf16x8 ra[TM], rb[TN];
  bool ra_ok[TM], rb_ok[TN];
  auto load_chunk = [&](int c) {
    const long pk = (long)(c < c_end ? c : c_end - 1) * WH_BK + kr;
    const bool inb = pk < p.Ps;
    const long pix = inb ? pk : 0;
    const int b = fdivb((int)pix, p.dHW);
    const int rem = (int)pix - b * p.Hs * p.Ws;
    const int hs = fdivb(rem, p.dW), ws = rem - hs * p.Ws;
    const int hl = hs * p.stride - p.pad + r, wl = ws * p.stride - p.pad + s;
    const bool gok = inb && hl >= 0 && wl >= 0 && hl < p.Hl && wl < p.Wl;
    const long gpix = ((long)b * p.Hl + hl) * p.Wl + wl;
    auto load_op = [&](const WgHOperand& op, int ch, bf16x8& v, bool& okr) {
      bool ok = (op.gathered ? gok : inb) && ch < op.C;
      const long px = op.gathered ? gpix : pix;
      v = *reinterpret_cast<const bf16x8*>(op.ptr + (ok ? px * op.ld + ch : 0L));
      okr = ok;
    };
#pragma unroll
    for (int j = 0; j < TM; ++j) load_op(p.row, m0 + l8 * 8 + 64 * j, ra[j], ra_ok[j]);
#pragma unroll
    for (int j = 0; j < TN; ++j) load_op(p.col, n0 + l8 * 8 + 64 * j, rb[j], rb_ok[j]);
  };
  const bool sqa = p.row.sq != 0, sqb = p.col.sq != 0;
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      bf16x8 v = ra[j];
      if (sqa) v = sq8(v);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        sA[buf][(l8 * 8 + 64 * j + e) * WH_LD + kpos] = ra_ok[j] ? v[e] : (bf16_t)0.0f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bf16x8 v = rb[j];
      if (sqb) v = sq8(v);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        sB[buf][(l8 * 8 + 64 * j + e) * WH_LD + kpos] = rb_ok[j] ? v[e] : (bf16_t)0.0f;
    }
  };
  auto compute = [&](int buf) {
    bf16x8 af[TM][2], bf[TN][2];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int ch = wm0 + a * 32 + li;
      const int sw = (ch >> 3) & 3;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        af[a][q] = *reinterpret_cast<const bf16x8*>(&sA[buf][ch * WH_LD + (((2 * q + lh) ^ sw) * 8)]);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int ch = wn0 + b * 32 + li;
      const int sw = (ch >> 3) & 3;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        bf[b][q] = *reinterpret_cast<const bf16x8*>(&sB[buf][ch * WH_LD + (((2 * q + lh) ^ sw) * 8)]);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
      if (a < m_live) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
          if (b < n_live) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][q], bf[b][q], acc[a][b], 0, 0, 0);
          }
      }
  };
  if (nloc > 0) {
    load_chunk(c_begin);
    store_chunk(0);
    __syncthreads();
    load_chunk(c_begin + 1);
    int c = 0;
    for (; c + 1 < nloc; c += 2) {
      store_chunk(1);
      load_chunk(c_begin + c + 2);
      compute(0);
      __syncthreads();
      store_chunk(0);
      load_chunk(c_begin + c + 3);
      compute(1);
      __syncthreads();
    }
    if (c < nloc) {
      compute(0);
      __syncthreads();
    }
  }
  float* slab = p.slabs + ((long)split * p.ntaps + tap) * p.row.C * p.col.C;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + wm0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
      if (m >= p.row.C) continue;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn0 + b * 32 + li;
        if (n < p.col.C) slab[(long)m * p.col.C + n] = acc[a][b][q];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* slabs, float* dst, int splitk,
                                                                int ntaps, int Cm, int Cn, long sm, long sn,
                                                                long stap, float scale) {
  const long total = (long)ntaps * Cm * Cn;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float acc = 0.0f;
    for (int z = 0; z < splitk; ++z) acc += slabs[(long)z * total + i];
    const int n = (int)(i % Cn);
    const long t2 = i / Cn;
    const int m = (int)(t2 % Cm);
    const int tap = (int)(t2 / Cm);
    dst[m * sm + n * sn + tap * stap] = acc * scale;
  }
}

struct WgHPlan {
  int TM, TN, MTt, NTt, ntaps, nchunks, splitk, cps, Cm, Cn;
};
static int wgh_plan(const lic_wgrad_desc* d, WgHPlan* pl) {
  if (!d || d->B <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Cp <= 0 || d->Cg <= 0 || d->kh <= 0 || d->kw <= 0 ||
      d->Hl <= 0 || d->Wl <= 0)
    return LIC_ERR_INVALID;
  if (d->Cp % 8 || d->Cg % 8 || d->p_ld % 8 || d->g_ld % 8) return LIC_ERR_UNSUPPORTED;
  pl->Cm = d->g_is_row ? d->Cg : d->Cp;
  pl->Cn = d->g_is_row ? d->Cp : d->Cg;
  pl->ntaps = d->kh * d->kw;
  pl->TM = (pl->Cm > 64 && pl->Cm % 128 == 0) ? 2 : 1;
  pl->TN = pl->Cn > 128 ? 3 : (pl->Cn > 64 ? 2 : 1);
  pl->MTt = (pl->Cm + 64 * pl->TM - 1) / (64 * pl->TM);
  pl->NTt = (pl->Cn + 64 * pl->TN - 1) / (64 * pl->TN);
  const long Ps = (long)d->B * d->Hs * d->Ws;
  pl->nchunks = (int)((Ps + WH_BK - 1) / WH_BK);
  const long base = (long)pl->MTt * pl->NTt * pl->ntaps;
  long sk = (1024 + base - 1) / base;
  const long max_sk = (pl->nchunks + 15) / 16;
  if (sk > max_sk) sk = max_sk;
  if (sk < 1) sk = 1;
  if (sk > 256) sk = 256;
  if (sk > 8) sk = (sk + 7) & ~7L;
  if (sk > max_sk) sk = max_sk;
  pl->cps = (int)((pl->nchunks + sk - 1) / sk);
  pl->splitk = (pl->nchunks + pl->cps - 1) / pl->cps;
  return LIC_OK;
}
LIC_EXPORT size_t lic_wgrad_bf16_workspace_bytes(const lic_wgrad_desc* d) {
  WgHPlan pl;
  if (wgh_plan(d, &pl) != LIC_OK) return 0;
  return (size_t)pl.splitk * pl.ntaps * pl.Cm * pl.Cn * sizeof(float);
}
// p / g are bf16 activations; dst and the workspace are fp32
LIC_EXPORT int lic_wgrad_bf16(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes,
                              lic_stream_t stream) {
  WgHPlan pl;
  int rc = wgh_plan(d, &pl);
  if (rc != LIC_OK) return rc;
  if (!d->p || !d->g || !d->dst || !workspace) return LIC_ERR_INVALID;
  if (!al16h(d->p) || !al16h(d->g)) return LIC_ERR_INVALID;
  const size_t need = (size_t)pl.splitk * pl.ntaps * pl.Cm * pl.Cn * sizeof(float);
  if (workspace_bytes < need) return LIC_ERR_WORKSPACE;
  if ((long)d->B * d->Hs * d->Ws > 0x7FFFFFFFL) return LIC_ERR_UNSUPPORTED;
  WgHOperand P, G;
  P.ptr = (const bf16_t*)d->p;
  P.ld = d->p_ld;
  P.C = d->Cp;
  P.gathered = 0;
  P.sq = d->sq_p;
  G.ptr = (const bf16_t*)d->g;
  G.ld = d->g_ld;
  G.C = d->Cg;
  G.gathered = !(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->Hl == d->Hs && d->Wl == d->Ws);
  G.sq = d->sq_g;
  WgradHParams p;
  p.row = d->g_is_row ? G : P;
  p.col = d->g_is_row ? P : G;
  p.slabs = (float*)workspace;
  p.B = d->B;
  p.Hs = d->Hs;
  p.Ws = d->Ws;
  p.Hl = d->Hl;
  p.Wl = d->Wl;
  p.kw = d->kw;
  p.stride = d->stride;
  p.pad = d->pad;
  p.ntaps = pl.ntaps;
  p.MTt = pl.MTt;
  p.NTt = pl.NTt;
  p.chunks_per_split = pl.cps;
  p.nchunks = pl.nchunks;
  p.Ps = (long)d->B * d->Hs * d->Ws;
  p.dHW = make_fastdivb((unsigned)(d->Hs * d->Ws));
  p.dW = make_fastdivb((unsigned)d->Ws);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(pl.MTt * pl.NTt * pl.ntaps * pl.splitk), block(256);
  if (pl.TM == 2 && pl.TN == 3)
    hipLaunchKernelGGL((wgrad_bf16_kernel<2, 3>), grid, block, 0, s, p);
  else if (pl.TM == 2 && pl.TN == 2)
    hipLaunchKernelGGL((wgrad_bf16_kernel<2, 2>), grid, block, 0, s, p);
  else if (pl.TM == 2 && pl.TN == 1)
    hipLaunchKernelGGL((wgrad_bf16_kernel<2, 1>), grid, block, 0, s, p);
  else if (pl.TM == 1 && pl.TN == 3)
    hipLaunchKernelGGL((wgrad_bf16_kernel<1, 3>), grid, block, 0, s, p);
  else if (pl.TM == 1 && pl.TN == 2)
    hipLaunchKernelGGL((wgrad_bf16_kernel<1, 2>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((wgrad_bf16_kernel<1, 1>), grid, block, 0, s, p);
  rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  const long total = (long)pl.ntaps * pl.Cm * pl.Cn;
  hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s, (const float*)workspace,
                     d->dst, pl.splitk, pl.ntaps, pl.Cm, pl.Cn, (long)d->dst_sm, (long)d->dst_sn, (long)d->dst_stap,
                     d->scale);
  return lic_check_launch();
}
