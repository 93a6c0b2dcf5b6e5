// fp32-MFMA implicit-GEMM kernels for gfx950 (MI355X): convolution family (lic_igemm) and the
// pixel-contraction weight gradient (lic_wgrad).  One wave owns 32x32 accumulator tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 64 cycles/instr/SIMD = the chip's fp32 peak);
// the kernels' job is to keep that pipe issuing: K is consumed in 16-deep chunks staged
// through LDS, the next chunk's global loads are issued before the current chunk's MFMAs, and
// 3-4 workgroups per CU cover each other's load/store phases.
#include "lic_common.h"

thread_local int g_lic_last_hip_error = 0;

// ------------------------------------------------------------------------------------------------
// igemm
// ------------------------------------------------------------------------------------------------
struct IgemmParams {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  float* out2;
  const float* aux;
  const float* aux2;
  const float* aux3;
  const float* res;
  long in_ld, out_ld, out2_ld, aux_ld, aux2_ld, aux3_ld, res_ld;
  int B, Hi, Wi, Cin, Ho, Wo, Cout;
  int kw, stride, pad, transposed, prologue, epilogue;
  float slope;
  int vec;     // float4 global loads legal
  int cpt;     // 16-deep chunks per tap
  int nphase;  // 1, or stride^2 output phases of a transposed conv
  int MT, NT;  // tiles in M (max over phases) and N
  int ntaps[4];
  int Hq[4], Wq[4];
  unsigned char taps[4][28];
};

constexpr int IG_BK = 16;
constexpr int IG_LDA = IG_BK + 4;

template <int BM, int BN, bool VEC>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int WM = BM / 2, WN = BN / 2;    // 2x2 waves
  constexpr int TM = WM / 32, TN = WN / 32;  // MFMA tiles per wave
  constexpr int APASS = BM / 64;             // float4 A loads per thread per chunk
  constexpr int BPASS = BN / 64;             // float4 B loads per thread per chunk
  __shared__ __attribute__((aligned(16))) float sA[BM * IG_LDA];
  __shared__ __attribute__((aligned(16))) float sB[IG_BK * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware block remap (bijective): consecutive ids on one XCD share A rows / the weight panel
  const int nwg = gridDim.x;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nt = wg % p.NT;
  const int mt = (wg / p.NT) % p.MT;
  const int phase = wg / (p.NT * p.MT);
  const int Hq = p.Hq[phase], Wq = p.Wq[phase];
  const int P = p.B * Hq * Wq;
  const int m0 = mt * BM, n0 = nt * BN;
  if (m0 >= P) return;
  const int sph = (p.nphase > 1) ? p.stride : 1;  // output step between rows of this phase
  const int py = (p.nphase > 1) ? phase / p.stride : 0;
  const int px = (p.nphase > 1) ? phase % p.stride : 0;

  // ---- per-thread A row slots -------------------------------------------------------------
  const int a_c4 = (tid & 3) * 4;
  int a_base[APASS], a_hy[APASS], a_wx[APASS];
  bool a_ok[APASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int prow = m0 + (tid >> 2) + 64 * j;
    a_ok[j] = prow < P;
    const int pr = a_ok[j] ? prow : 0;
    const int b = pr / (Hq * Wq);
    const int rem = pr - b * Hq * Wq;
    const int i = rem / Wq, jj = rem - i * Wq;
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
  // B slots
  int b_kr[BPASS], b_c4[BPASS];
#pragma unroll
  for (int j = 0; j < BPASS; ++j) {
    const int idx = tid + 256 * j;
    b_kr[j] = idx / (BN / 4);
    b_c4[j] = (idx % (BN / 4)) * 4;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  int n_live_cnt = 0;  // live 32-column tiles of this wave (wave-uniform)
#pragma unroll
  for (int b = 0; b < TN; ++b) n_live_cnt += ((n0 + wn0 + b * 32) < p.Cout) ? 1 : 0;

  const int ntaps = p.ntaps[phase];
  const int nchunks = ntaps * p.cpt;
  f32x4 ra[APASS], rb[BPASS];
  bool ra_ok[APASS], rb_ok[BPASS];

  // Issue the global loads of one chunk.  On the vector path nothing here consumes a loaded
  // value (out-of-range lanes load from a safe address and are zeroed when the registers are
  // written to LDS), so the loads stay in flight under the MFMAs of the current chunk.
  auto load_chunk = [&](int tapi, int cb) {
    const int tap = p.taps[phase][tapi];
    const int r = tap / p.kw, s = tap - r * p.kw;
    const int ci0 = cb * IG_BK;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      int ih, iw;
      bool ok = a_ok[j];
      if (p.transposed) {
        const int nh = a_hy[j] - r, nw = a_wx[j] - s;
        ok = ok && nh >= 0 && nw >= 0;
        ih = (p.stride == 2) ? (nh >> 1) : nh / p.stride;
        iw = (p.stride == 2) ? (nw >> 1) : nw / p.stride;
      } else {
        ih = a_hy[j] + r;
        iw = a_wx[j] + s;
        ok = ok && ih >= 0 && iw >= 0;
      }
      ok = ok && ih < p.Hi && iw < p.Wi;
      const int ci = ci0 + a_c4;
      if (VEC) {
        ok = ok && ci < p.Cin;
        const long off = ok ? (long)(a_base[j] + ih * p.Wi + iw) * p.in_ld + ci : 0L;
        ra[j] = *reinterpret_cast<const f32x4*>(p.in + off);
        ra_ok[j] = ok;
      } else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const float* src = p.in + (long)(a_base[j] + ih * p.Wi + iw) * p.in_ld + ci;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ci + e < p.Cin) v[e] = src[e];
        }
        ra[j] = v;
        ra_ok[j] = true;
      }
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const int k = ci0 + b_kr[j];
      const int n = n0 + b_c4[j];
      if (VEC) {
        const bool ok = k < p.Cin && n < p.Cout;
        const long off = ok ? ((long)tap * p.Cin + k) * p.Cout + n : 0L;
        rb[j] = *reinterpret_cast<const f32x4*>(p.w + off);
        rb_ok[j] = ok;
      } else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < p.Cin) {
          const float* src = p.w + ((long)tap * p.Cin + k) * p.Cout + n;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) v[e] = src[e];
        }
        rb[j] = v;
        rb_ok[j] = true;
      }
    }
  };

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int tapi = 0, cb = 0;
  if (nchunks > 0) load_chunk(0, 0);
  for (int c = 0; c < nchunks; ++c) {
    if (c > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      f32x4 v = ra_ok[j] ? ra[j] : zero4;
      if (p.prologue == 1) v = v * v;
      *reinterpret_cast<f32x4*>(&sA[((tid >> 2) + 64 * j) * IG_LDA + a_c4]) = v;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j)
      *reinterpret_cast<f32x4*>(&sB[b_kr[j] * BN + b_c4[j]]) = rb_ok[j] ? rb[j] : zero4;
    __syncthreads();
    if (++cb == p.cpt) {
      cb = 0;
      ++tapi;
    }
    if (c + 1 < nchunks) load_chunk(tapi, cb);

    float af[TM][8], bf[TN][8];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const float* src = &sA[(wm0 + a * 32 + li) * IG_LDA + lh * 8];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        af[a][e] = v0[e];
        af[a][4 + e] = v1[e];
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int t = 0; t < 8; ++t) bf[b][t] = sB[(lh * 8 + t) * BN + wn0 + b * 32 + li];
    if (n_live_cnt == TN) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
    } else if (n_live_cnt > 0) {  // only the first 32-column tile of this wave is live
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int a = 0; a < TM; ++a)
          acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[0][t], acc[a][0], 0, 0, 0);
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------
  const int epi = p.epilogue;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int prow = m0 + row;
      if (prow >= P) continue;
      long opix;
      if (p.nphase > 1) {
        const int b = prow / (Hq * Wq);
        const int rem = prow - b * Hq * Wq;
        const int i = rem / Wq, jj = rem - i * Wq;
        opix = ((long)b * p.Ho + i * sph + py) * p.Wo + jj * sph + px;
      } else {
        opix = prow;
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn0 + b * 32 + li;
        if (col >= p.Cout) continue;
        float v = acc[a][b][r];
        if (p.bias) v += p.bias[col];
        if (epi == LIC_EPI_LEAKY) {
          v = v > 0.0f ? v : v * p.slope;
          if (p.res) p.out2[opix * p.out2_ld + col] = v + p.res[opix * p.res_ld + col];
        } else {
          if (epi == LIC_EPI_MUL_LEAKY_MASK) {
            v = p.aux[opix * p.aux_ld + col] > 0.0f ? v : v * p.slope;
          } else if (epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN) {
            if (p.out2) p.out2[opix * p.out2_ld + col] = v;
            const float f = (epi == LIC_EPI_GDN) ? 1.0f / sqrtf(v) : sqrtf(v);
            v = p.aux[opix * p.aux_ld + col] * f;
          } else if (epi == LIC_EPI_GDN_BWD || epi == LIC_EPI_IGDN_BWD) {
            const float n = p.aux3[opix * p.aux3_ld + col];
            const float f = (epi == LIC_EPI_GDN_BWD) ? 1.0f / sqrtf(n) : sqrtf(n);
            v = p.aux[opix * p.aux_ld + col] * f + 2.0f * p.aux2[opix * p.aux2_ld + col] * v;
          }
          if (p.res) v += p.res[opix * p.res_ld + col];
        }
        p.out[opix * p.out_ld + col] = v;
      }
    }
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// fills the kernel parameter block; returns LIC_OK, or 1 when there is nothing to launch
static int igemm_prepare(const lic_igemm_desc* d, IgemmParams& p, int& BM, int& BN, long& nwg,
                         int64_t& live_macs) {
  if (!d || !d->in || !d->w || !d->out) return LIC_ERR_INVALID;
  if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 ||
      d->Cout <= 0 || d->kh <= 0 || d->kw <= 0)
    return LIC_ERR_INVALID;
  if (d->kh * d->kw > 28 || d->stride < 1 || d->stride > 2) return LIC_ERR_UNSUPPORTED;
  const int epi = d->epilogue;
  if ((epi == LIC_EPI_MUL_LEAKY_MASK || epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN) && !d->aux)
    return LIC_ERR_INVALID;
  if ((epi == LIC_EPI_GDN_BWD || epi == LIC_EPI_IGDN_BWD) && (!d->aux || !d->aux2 || !d->aux3))
    return LIC_ERR_INVALID;
  if (epi == LIC_EPI_LEAKY && d->res && !d->out2) return LIC_ERR_INVALID;

  p.in = d->in;
  p.w = d->w;
  p.bias = d->bias;
  p.out = d->out;
  p.out2 = d->out2;
  p.aux = d->aux;
  p.aux2 = d->aux2;
  p.aux3 = d->aux3;
  p.res = d->res;
  p.in_ld = d->in_ld;
  p.out_ld = d->out_ld;
  p.out2_ld = d->out2_ld;
  p.aux_ld = d->aux_ld;
  p.aux2_ld = d->aux2_ld;
  p.aux3_ld = d->aux3_ld;
  p.res_ld = d->res_ld;
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
  p.slope = d->slope;
  p.vec = (d->Cin % 4 == 0) && (d->in_ld % 4 == 0) && (d->Cout % 4 == 0) && aligned16(d->in) &&
          aligned16(d->w);
  p.cpt = (d->Cin + IG_BK - 1) / IG_BK;
  const uint32_t mask = d->tap_mask ? d->tap_mask : 0xFFFFFFFFu;
  p.nphase = (p.transposed && d->stride > 1) ? d->stride * d->stride : 1;
  long maxP = 0;
  live_macs = 0;
  for (int ph = 0; ph < 4; ++ph) {
    p.ntaps[ph] = 0;
    p.Hq[ph] = p.Wq[ph] = 0;
  }
  for (int ph = 0; ph < p.nphase; ++ph) {
    const int py = (p.nphase > 1) ? ph / d->stride : 0, px = (p.nphase > 1) ? ph % d->stride : 0;
    const int st = (p.nphase > 1) ? d->stride : 1;
    p.Hq[ph] = (d->Ho - py + st - 1) / st;
    p.Wq[ph] = (d->Wo - px + st - 1) / st;
    if (p.Hq[ph] < 0) p.Hq[ph] = 0;
    if (p.Wq[ph] < 0) p.Wq[ph] = 0;
    const long Pp = (long)d->B * p.Hq[ph] * p.Wq[ph];
    if (Pp > maxP) maxP = Pp;
    int n = 0;
    for (int r = 0; r < d->kh; ++r)
      for (int s = 0; s < d->kw; ++s) {
        const int t = r * d->kw + s;
        if (!((mask >> t) & 1u)) continue;
        if (p.nphase > 1) {
          if (((py + d->pad - r) % d->stride) != 0 || ((px + d->pad - s) % d->stride) != 0) continue;
        }
        p.taps[ph][n++] = (unsigned char)t;
      }
    p.ntaps[ph] = n;
    live_macs += (int64_t)Pp * n * d->Cin * d->Cout;
  }
  if (maxP <= 0) return 1;
  if (maxP > 0x7FFFFFFFL / 2) return LIC_ERR_UNSUPPORTED;

  // tile selection: minimise padded N (a 64-wide tile whose second 32-column half is dead costs
  // nothing: waves skip it); keep >= ~768 workgroups when the layer allows it
  const int pad128 = ((d->Cout + 127) / 128) * 128;
  const int eff64 = ((d->Cout + 31) / 32) * 32;
  BN = (pad128 <= eff64 && d->Cout > 64) ? 128 : 64;
  if (!p.vec) BN = 64;
  p.NT = (d->Cout + BN - 1) / BN;
  const long wg128 = ((maxP + 127) / 128) * p.NT * p.nphase;
  BM = (wg128 >= 768 && p.vec) ? 128 : 64;
  p.MT = (int)((maxP + BM - 1) / BM);
  nwg = (long)p.MT * p.NT * p.nphase;
  if (nwg > 0x7FFFFFFFL) return LIC_ERR_UNSUPPORTED;
  return LIC_OK;
}

LIC_EXPORT int lic_igemm_plan(const lic_igemm_desc* d, int32_t* BM, int32_t* BN, int64_t* live_macs) {
  IgemmParams p;
  int bm = 0, bn = 0;
  long nwg = 0;
  int64_t macs = 0;
  const int rc = igemm_prepare(d, p, bm, bn, nwg, macs);
  if (rc < 0) return rc;
  if (BM) *BM = bm;
  if (BN) *BN = bn;
  if (live_macs) *live_macs = macs;
  return LIC_OK;
}

LIC_EXPORT int lic_igemm(const lic_igemm_desc* d, lic_stream_t stream) {
  IgemmParams p;
  int BM = 0, BN = 0;
  long nwg = 0;
  int64_t macs = 0;
  const int rc = igemm_prepare(d, p, BM, BN, nwg, macs);
  if (rc < 0) return rc;
  if (rc == 1) return LIC_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)nwg), block(256);
  if (!p.vec)  // odd channel counts / unaligned views: scalar-load variant, one tile shape
    hipLaunchKernelGGL((igemm_kernel<64, 64, false>), grid, block, 0, s, p);
  else if (BM == 128 && BN == 128)
    hipLaunchKernelGGL((igemm_kernel<128, 128, true>), grid, block, 0, s, p);
  else if (BM == 128 && BN == 64)
    hipLaunchKernelGGL((igemm_kernel<128, 64, true>), grid, block, 0, s, p);
  else if (BM == 64 && BN == 128)
    hipLaunchKernelGGL((igemm_kernel<64, 128, true>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((igemm_kernel<64, 64, true>), grid, block, 0, s, p);
  return lic_check_launch();
}

// ------------------------------------------------------------------------------------------------
// wgrad: R[tap][m][n] = sum over small-grid pixels of row_operand[m] * col_operand[n]
// ------------------------------------------------------------------------------------------------
struct WgOperand {
  const float* ptr;
  long ld;
  int C;
  int gathered;  // sample the large grid at (hs*stride-pad+r, ws*stride-pad+s)
  int sq;
  int vec;
};
struct WgradParams {
  WgOperand row, col;
  float* slabs;  // [splitk][ntaps][Cm][Cn]
  int B, Hs, Ws, Hl, Wl;
  int kw, stride, pad, ntaps;
  int MTt, NTt;  // tiles
  int chunks_per_split, nchunks;
  long Ps;
};

constexpr int WG_BK = 16;

template <int TMt, int TNt>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int WM = TMt / 2, WN = TNt / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int APASS = TMt / 64, BPASS = TNt / 64;
  __shared__ __attribute__((aligned(16))) float sA[WG_BK * TMt];
  __shared__ __attribute__((aligned(16))) float sB[WG_BK * TNt];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int li = lane & 31, lh = lane >> 5;
  const int mt = blockIdx.x / p.NTt, nt = blockIdx.x % p.NTt;
  const int tap = blockIdx.y, split = blockIdx.z;
  const int m0 = mt * TMt, n0 = nt * TNt;
  const int r = tap / p.kw, s = tap - r * p.kw;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);

  int a_kr[APASS], a_c4[APASS], b_kr[BPASS], b_c4[BPASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int idx = tid + 256 * j;
    a_kr[j] = idx / (TMt / 4);
    a_c4[j] = (idx % (TMt / 4)) * 4;
  }
#pragma unroll
  for (int j = 0; j < BPASS; ++j) {
    const int idx = tid + 256 * j;
    b_kr[j] = idx / (TNt / 4);
    b_c4[j] = (idx % (TNt / 4)) * 4;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;
  bool m_live[TM], n_live[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) m_live[a] = (m0 + wm0 + a * 32) < p.row.C;
#pragma unroll
  for (int b = 0; b < TN; ++b) n_live[b] = (n0 + wn0 + b * 32) < p.col.C;

  auto load_op = [&](const WgOperand& op, long pk, int ch) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (pk >= p.Ps) return v;
    long pix = pk;
    if (op.gathered) {
      const int hw = p.Hs * p.Ws;
      const int b = (int)(pk / hw);
      const int rem = (int)(pk - (long)b * hw);
      const int hs = rem / p.Ws, ws = rem - hs * p.Ws;
      const int hl = hs * p.stride - p.pad + r, wl = ws * p.stride - p.pad + s;
      if (hl < 0 || wl < 0 || hl >= p.Hl || wl >= p.Wl) return v;
      pix = ((long)b * p.Hl + hl) * p.Wl + wl;
    }
    const float* src = op.ptr + pix * op.ld + ch;
    if (op.vec) {
      if (ch < op.C) v = *reinterpret_cast<const f32x4*>(src);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ch + e < op.C) v[e] = src[e];
    }
    return v;
  };

  f32x4 ra[APASS], rb[BPASS];
  auto load_chunk = [&](int c) {
    const long k0 = (long)c * WG_BK;
#pragma unroll
    for (int j = 0; j < APASS; ++j) ra[j] = load_op(p.row, k0 + a_kr[j], m0 + a_c4[j]);
#pragma unroll
    for (int j = 0; j < BPASS; ++j) rb[j] = load_op(p.col, k0 + b_kr[j], n0 + b_c4[j]);
  };

  if (c_begin < c_end) load_chunk(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    if (c > c_begin) __syncthreads();
#pragma unroll
    for (int j = 0; j < APASS; ++j)
      *reinterpret_cast<f32x4*>(&sA[a_kr[j] * TMt + a_c4[j]]) = p.row.sq ? ra[j] * ra[j] : ra[j];
#pragma unroll
    for (int j = 0; j < BPASS; ++j)
      *reinterpret_cast<f32x4*>(&sB[b_kr[j] * TNt + b_c4[j]]) = p.col.sq ? rb[j] * rb[j] : rb[j];
    __syncthreads();
    if (c + 1 < c_end) load_chunk(c + 1);
    float af[TM][8], bf[TN][8];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int t = 0; t < 8; ++t) af[a][t] = sA[(lh * 8 + t) * TMt + wm0 + a * 32 + li];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int t = 0; t < 8; ++t) bf[b][t] = sB[(lh * 8 + t) * TNt + wn0 + b * 32 + li];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          if (m_live[a] && n_live[b])
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
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

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slabs, float* dst, int splitk,
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

struct WgPlan {
  int TMt, TNt, MTt, NTt, ntaps, nchunks, splitk, cps;
  int Cm, Cn;
};
static int wg_plan(const lic_wgrad_desc* d, WgPlan* pl) {
  if (!d || d->B <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Cp <= 0 || d->Cg <= 0 || d->kh <= 0 ||
      d->kw <= 0 || d->Hl <= 0 || d->Wl <= 0)
    return LIC_ERR_INVALID;
  pl->Cm = d->g_is_row ? d->Cg : d->Cp;
  pl->Cn = d->g_is_row ? d->Cp : d->Cg;
  pl->ntaps = d->kh * d->kw;
  pl->TMt = 64;
  pl->TNt = 64;
  pl->MTt = (pl->Cm + 63) / 64;
  pl->NTt = (pl->Cn + 63) / 64;
  const long Ps = (long)d->B * d->Hs * d->Ws;
  pl->nchunks = (int)((Ps + WG_BK - 1) / WG_BK);
  const long base = (long)pl->MTt * pl->NTt * pl->ntaps;
  long sk = (2048 + base - 1) / base;
  const long max_sk = (pl->nchunks + 15) / 16;  // at least 16 chunks (256 pixels) per split
  if (sk > max_sk) sk = max_sk;
  if (sk < 1) sk = 1;
  if (sk > 256) sk = 256;
  pl->cps = (int)((pl->nchunks + sk - 1) / sk);
  pl->splitk = (pl->nchunks + pl->cps - 1) / pl->cps;
  return LIC_OK;
}

LIC_EXPORT size_t lic_wgrad_workspace_bytes(const lic_wgrad_desc* d) {
  WgPlan pl;
  if (wg_plan(d, &pl) != LIC_OK) return 0;
  return (size_t)pl.splitk * pl.ntaps * pl.Cm * pl.Cn * sizeof(float);
}

LIC_EXPORT int lic_wgrad(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes,
                         lic_stream_t stream) {
  WgPlan pl;
  int rc = wg_plan(d, &pl);
  if (rc != LIC_OK) return rc;
  if (!d->p || !d->g || !d->dst || !workspace) return LIC_ERR_INVALID;
  if (d->stride < 1) return LIC_ERR_UNSUPPORTED;
  const size_t need = (size_t)pl.splitk * pl.ntaps * pl.Cm * pl.Cn * sizeof(float);
  if (workspace_bytes < need) return LIC_ERR_WORKSPACE;
  WgOperand P, G;
  P.ptr = d->p;
  P.ld = d->p_ld;
  P.C = d->Cp;
  P.gathered = 0;
  P.sq = d->sq_p;
  P.vec = (d->Cp % 4 == 0) && (d->p_ld % 4 == 0) && aligned16(d->p);
  G.ptr = d->g;
  G.ld = d->g_ld;
  G.C = d->Cg;
  G.gathered = !(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->Hl == d->Hs &&
                 d->Wl == d->Ws);
  G.sq = d->sq_g;
  G.vec = (d->Cg % 4 == 0) && (d->g_ld % 4 == 0) && aligned16(d->g);
  WgradParams p;
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
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(pl.MTt * pl.NTt, pl.ntaps, pl.splitk), block(256);
  hipLaunchKernelGGL((wgrad_kernel<64, 64>), grid, block, 0, s, p);
  rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  const long total = (long)pl.ntaps * pl.Cm * pl.Cn;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s,
                     (const float*)workspace, d->dst, pl.splitk, pl.ntaps, pl.Cm, pl.Cn, (long)d->dst_sm,
                     (long)d->dst_sn, (long)d->dst_stap, d->scale);
  return lic_check_launch();
}

LIC_EXPORT int lic_version(void) { return LIC_ABI_VERSION; }
LIC_EXPORT int lic_last_hip_error(void) { return g_lic_last_hip_error; }
LIC_EXPORT const char* lic_arch(void) { return "gfx950"; }
