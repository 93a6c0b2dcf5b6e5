// lic_prep: ONE launch per optimizer step that refreshes every buffer derived from the parameters.
//
// A training step of the hot path used to spend ~90 of its ~300 launches on parameter-sized helpers:
// lic_pack_weight for the forward and the data-gradient layout of every conv / convT weight
// (Components.py:10-16,39-45,69-73,99-103; ContextModels.py:19-20; ParametersModels.py:22-34), the GDN
// re-parametrisation (compressai GDN at Components.py:11-44: beta_eff, gamma_eff) plus two packed gamma
// panels per GDN, the column-matrix forms of the RGB stem / head weights, and the in-place masking of
// the context model's weight (ContextModels.py:19).  Each is a few microseconds of work behind ~6 us of
// dispatch latency on the stream that also carries the MFMA launches.  The parameters change exactly
// once per step (optimizer.step()), so all of it is one table-driven launch at the top of forward():
// block b looks up its job in a device-resident table (built once per model) and runs the same
// per-element / per-tile code as the stand-alone entry points (lic_pack_weight, lic_pack_weight_bf16,
// lic_gdn_reparam, lic_mul_inplace) -- same values, bit for bit.
#include "lic_common.h"
#include <hip/hip_bf16.h>

namespace {

typedef __hip_bfloat16 bf16_t;
constexpr int PP_BK = 16;     // fp32 packed chunk depth (lic_gemm.hip IG_BK)
constexpr int PP_HBK = 32;    // bf16 packed chunk depth (lic_gemm_bf16.hip HB_BK)
constexpr int PP_NS = 8;      // columns per tiled block
constexpr int PP_MAXTAPS = 28;
constexpr int PP_ITEMS = 2048;  // elements per block of the element-wise job kinds

__device__ __forceinline__ float pp_value(const lic_prep_job& j, long off) {
  float v = j.src[off];
  if (j.mask) v *= j.mask[off];
  if (j.transform == 1) v = lic_reparam(v, j.bound, j.pedestal);  // as lic_gdn_reparam
  return v;
}
__device__ __forceinline__ long pp_offset(const lic_prep_job& j, int tap, int k, int n) {
  long off = (long)tap * j.s_tap;
  if (j.kdiv > 0) {
    const int q = k / j.kdiv;
    off += (long)q * j.s_kq + (long)(k - q * j.kdiv) * j.s_kr;
  } else {
    off += (long)k * j.s_kq;
  }
  if (j.ndiv > 0) {
    const int q = n / j.ndiv;
    off += (long)q * j.s_nq + (long)(n - q * j.ndiv) * j.s_nr;
  } else {
    off += (long)n * j.s_nq;
  }
  return off;
}

// fp32 MFMA-operand layout dst[tap][chunk][n/32][q][lane][4] (lic_gemm.hip pack_weight_kernel)
__device__ void pp_pack_f32(const lic_prep_job& j, int blk) {
  float* dst = (float*)j.dst;
  const int ntile = j.npad >> 5;
  const long end = min(j.total, (long)(blk + 1) * PP_ITEMS);
  for (long i = (long)blk * PP_ITEMS + threadIdx.x; i < end; i += 256) {
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63), q = (int)((i >> 8) & 1);
    long t = i >> 9;
    const int tile = (int)(t % ntile);
    t /= ntile;
    const int cb = (int)(t % j.cpt);
    const int tap = (int)(t / j.cpt);
    const int n = tile * 32 + (lane & 31);
    const int k = cb * PP_BK + (lane >> 5) * 8 + q * 4 + e;
    dst[i] = (k < j.K && n < j.N) ? pp_value(j, pp_offset(j, tap, k, n)) : 0.0f;
  }
}
// bf16 layout dst[tap][chunk][n/32][kstep][lane][8] (lic_gemm_bf16.hip pack_weight_bf16_kernel)
__device__ void pp_pack_bf16(const lic_prep_job& j, int blk) {
  bf16_t* dst = (bf16_t*)j.dst;
  const int ntile = j.npad >> 5;
  const long end = min(j.total, (long)(blk + 1) * PP_ITEMS);
  for (long i = (long)blk * PP_ITEMS + threadIdx.x; i < end; i += 256) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), q = (int)((i >> 9) & 1);
    long t = i >> 10;
    const int tile = (int)(t % ntile);
    t /= ntile;
    const int cb = (int)(t % j.cpt);
    const int tap = (int)(t / j.cpt);
    const int n = tile * 32 + (lane & 31);
    // (KPERM: the K order of lic_igemm_bf16's fused GDN pool, whose operand comes straight from accumulator registers)
    const int kin = (j.kind == LIC_PREP_PACK_BF16_KPERM) ? ((e >> 2) * 8 + (lane >> 5) * 4 + (e & 3)) : ((lane >> 5) * 8 + e);
    const int k = cb * PP_HBK + q * 16 + kin;
    if (j.kind == LIC_PREP_PACK_BF16_STEM) {  // lic_pack_stem_weight_bf16: K = 16*r + 3*s + c of a [N][3][5][5] weight
      const int r = k >> 4, jj = k & 15, s = jj / 3;
      dst[i] = (bf16_t)((r < 5 && jj < 15 && n < j.N) ? j.src[n * 75 + (jj - 3 * s) * 25 + r * 5 + s] : 0.0f);
      continue;
    }
    dst[i] = (bf16_t)((k < j.K && n < j.N) ? pp_value(j, pp_offset(j, tap, k, n)) : 0.0f);
  }
}
// element-wise map dst[i] = value(src[i]) (beta_eff) and in-place masking src[i] *= mask[i]
__device__ void pp_map(const lic_prep_job& j, int blk) {
  const long end = min(j.total, (long)(blk + 1) * PP_ITEMS);
  if (j.kind == LIC_PREP_MASK_INPLACE) {
    float* w = const_cast<float*>(j.src);
    for (long i = (long)blk * PP_ITEMS + threadIdx.x; i < end; i += 256) w[i] *= j.mask[i];
  } else {
    float* dst = (float*)j.dst;
    for (long i = (long)blk * PP_ITEMS + threadIdx.x; i < end; i += 256) dst[i] = pp_value(j, i);
  }
}

// The tiled packer of lic_gemm.hip for conv-shaped sources (taps innermost and contiguous): one block per
// (16-k chunk, 8 columns) stages its source block through LDS with 16-byte loads along the contiguous
// direction and writes 128-byte pieces of the packed layout.  RUN_K: for a fixed n the (k, tap) run is
// contiguous (s_k == taps); else for a fixed k the (n, tap) run is (s_n == taps).
// HALF: the bf16 layout (32-k chunks: dst[tap][chunk][n/32][kstep][lane][8]) from the same staged block -- the element-wise
// bf16 packer read its source 4 bytes at a time, 100 bytes apart (57 us for the 29 MB of config 3's weights, in front of
// the first launch of every step).
template <bool RUN_K, bool HALF = false>
__device__ void pp_pack_tiled(const lic_prep_job& j, int blk, float* st) {
  constexpr int TBK = HALF ? PP_HBK : PP_BK;   // k depth of the staged block = one packed chunk
  const float* src = j.src;
  float* dst = (float*)j.dst;
  const int taps = j.taps, K = j.K, N = j.N, cpt = j.cpt, ntile = j.npad >> 5;
  const long s_k = j.s_kq, s_n = j.s_nq;
  const int cb = blk % cpt, nb = blk / cpt;
  const int k0 = cb * TBK, n0 = nb * PP_NS;
  const int run = (RUN_K ? TBK : PP_NS) * taps, total = TBK * PP_NS * taps;
  const bool v4 = j.v4 != 0;
  if (v4 && n0 >= N) {
    for (int i = threadIdx.x; i < taps * TBK * (PP_NS + 1); i += 256) st[i] = 0.0f;
  } else if (v4) {
    for (int base = threadIdx.x; base < total / 4; base += 4 * 256) {
      f32x4 v[4];
      int jj[4], oo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = (base + u * 256) * 4;
        oo[u] = -1;
        if (idx < total) {
          const int o = idx / run, jx = idx - o * run;
          oo[u] = o;
          jj[u] = jx;
          v[u] = *reinterpret_cast<const f32x4*>(RUN_K ? src + (long)(n0 + o) * s_n + (long)k0 * taps + jx
                                                       : src + (long)(k0 + o) * s_k + (long)n0 * taps + jx);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (oo[u] >= 0) {
          int i = jj[u] / taps, tap = jj[u] - i * taps;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kl = RUN_K ? i : oo[u], nl = RUN_K ? oo[u] : i;
            st[(tap * TBK + kl) * (PP_NS + 1) + nl] = v[u][e];
            if (++tap == taps) {
              tap = 0;
              ++i;
            }
          }
        }
    }
  } else {
    for (int base = threadIdx.x; base < total; base += 4 * 256) {
      float v[4];
      int slot[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * 256;
        v[u] = 0.0f;
        slot[u] = -1;
        if (idx < total) {
          const int o = idx / run, jx = idx - o * run;
          const int i = jx / taps, tap = jx - i * taps;
          const int kl = RUN_K ? i : o, nl = RUN_K ? o : i;
          slot[u] = (tap * TBK + kl) * (PP_NS + 1) + nl;
          if (k0 + kl < K && n0 + nl < N)
            v[u] = RUN_K ? src[(long)(n0 + nl) * s_n + (long)k0 * taps + jx] : src[(long)(k0 + kl) * s_k + (long)n0 * taps + jx];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (slot[u] >= 0) st[slot[u]] = v[u];
    }
  }
  __syncthreads();
  const int tile = n0 >> 5, nin = n0 & 31;
  if constexpr (HALF) {
    typedef __bf16 pp_bf16x8 __attribute__((ext_vector_type(8)));
    typedef float pp_f32x8 __attribute__((ext_vector_type(8)));
    bf16_t* dh = (bf16_t*)j.dst;
    for (int idx = threadIdx.x; idx < taps * 32; idx += 256) {   // (tap, kstep q, lane half lh, column nl): 8 k each
      const int tap = idx >> 5, r = idx & 31, q = r >> 4, lh = (r >> 3) & 1, nl = r & 7;
      const int kl = q * 16 + lh * 8, lane = lh * 32 + nin + nl;
      pp_f32x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = st[(tap * TBK + kl + e) * (PP_NS + 1) + nl];
      *reinterpret_cast<pp_bf16x8*>(dh + ((((long)tap * cpt + cb) * ntile + tile) * 2 + q) * 512 + lane * 8) =
          __builtin_convertvector(o, pp_bf16x8);
    }
    return;
  }
  for (int idx = threadIdx.x; idx < taps * 32; idx += 256) {
    const int tap = idx >> 5, r = idx & 31, q = r >> 4, lh = (r >> 3) & 1, nl = r & 7;
    const int kl = lh * 8 + q * 4, lane = lh * 32 + nin + nl;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = st[(tap * TBK + kl + e) * (PP_NS + 1) + nl];
    *reinterpret_cast<f32x4*>(dst + (((long)tap * cpt + cb) * ntile + tile) * 512 + q * 256 + lane * 4) = o;
  }
}

__global__ __launch_bounds__(256) void prep_kernel(const lic_prep_job* jobs, int njobs) {
  __shared__ __attribute__((aligned(16))) float st[PP_MAXTAPS * PP_HBK * (PP_NS + 1)];
  __shared__ lic_prep_job job;
  __shared__ int s_blk;
  if (threadIdx.x == 0) {
    // binary search: last job whose first block is <= this block
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].block0 <= b) lo = mid;
      else hi = mid - 1;
    }
    job = jobs[lo];
    s_blk = b - jobs[lo].block0;
  }
  __syncthreads();
  const int blk = s_blk;
  if (blk >= job.nblocks) return;
  if (job.kind == LIC_PREP_PACK_F32) {
    if (job.tiled == 1) pp_pack_tiled<true>(job, blk, st);
    else if (job.tiled == 2) pp_pack_tiled<false>(job, blk, st);
    else pp_pack_f32(job, blk);
  } else if (job.kind == LIC_PREP_PACK_BF16 || job.kind == LIC_PREP_PACK_BF16_KPERM || job.kind == LIC_PREP_PACK_BF16_STEM) {
    if (job.tiled == 1) pp_pack_tiled<true, true>(job, blk, st);
    else if (job.tiled == 2) pp_pack_tiled<false, true>(job, blk, st);
    else pp_pack_bf16(job, blk);
  } else {
    pp_map(job, blk);
  }
}

bool pp_al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// Fills the derived fields (cpt, npad, tiled, v4, total, block0, nblocks) of a host-side job array in
// place and returns the grid size of lic_prep_run, or a negative lic_status.
LIC_EXPORT int64_t lic_prep_plan(lic_prep_job* jobs, int32_t njobs) {
  if (!jobs || njobs <= 0) return LIC_ERR_INVALID;
  long blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    lic_prep_job& j = jobs[i];
    if (!j.src) return LIC_ERR_INVALID;
    j.tiled = j.v4 = 0;
    j.cpt = j.npad = 0;
    if (j.kind == LIC_PREP_PACK_BF16_STEM) {
      j.taps = 1;
      j.K = 80;
    }
    if (j.kind == LIC_PREP_PACK_F32 || j.kind == LIC_PREP_PACK_BF16 || j.kind == LIC_PREP_PACK_BF16_KPERM ||
        j.kind == LIC_PREP_PACK_BF16_STEM) {
      if (!j.dst || j.taps <= 0 || j.K <= 0 || j.N <= 0 || !pp_al16(j.dst)) return LIC_ERR_INVALID;
      const bool h = j.kind != LIC_PREP_PACK_F32;
      const int bk = h ? PP_HBK : PP_BK;
      j.cpt = (j.K + bk - 1) / bk;
      j.npad = h ? ((j.N + 63) / 64) * 64 : lic_npad_f32(j.N);
      j.total = (long)j.taps * j.cpt * j.npad * bk;
      const long s_k = j.s_kq, s_n = j.s_nq;
      const bool plain = j.kdiv == 0 && j.ndiv == 0 && j.transform == 0 && j.mask == nullptr;
      static const bool tiled_h = [] {
        const char* e = getenv("LIC_PREP_TILED_BF16");
        return !(e && e[0] == '0');
      }();
      if ((!h || (j.kind == LIC_PREP_PACK_BF16 && tiled_h)) && plain && j.s_tap == 1 && j.taps >= 4 && j.taps <= PP_MAXTAPS &&
          (s_k == j.taps || s_n == j.taps)) {
        j.tiled = (s_k == j.taps) ? 1 : 2;
        j.v4 = (j.K % bk == 0 && j.N % PP_NS == 0 && pp_al16(j.src) && ((s_k == j.taps ? s_n : s_k) % 4 == 0)) ? 1 : 0;
        j.nblocks = j.cpt * (j.npad / PP_NS);
      } else {
        j.nblocks = (int)((j.total + PP_ITEMS - 1) / PP_ITEMS);
      }
    } else if (j.kind == LIC_PREP_MAP || j.kind == LIC_PREP_MASK_INPLACE) {
      if (j.N <= 0 || (j.kind == LIC_PREP_MAP && !j.dst) || (j.kind == LIC_PREP_MASK_INPLACE && !j.mask)) return LIC_ERR_INVALID;
      j.total = j.N;
      j.nblocks = (int)((j.total + PP_ITEMS - 1) / PP_ITEMS);
    } else {
      return LIC_ERR_UNSUPPORTED;
    }
    if (blocks + j.nblocks > 0x7FFFFFFFL) return LIC_ERR_UNSUPPORTED;
    j.block0 = (int)blocks;
    blocks += j.nblocks;
  }
  return blocks;
}

// `jobs_device`: the planned array copied to device memory (it stays valid as long as the parameter and
// destination pointers do); `total_blocks`: lic_prep_plan's return value.
LIC_EXPORT int lic_prep_run(const lic_prep_job* jobs_device, int32_t njobs, int64_t total_blocks, lic_stream_t stream) {
  if (!jobs_device || njobs <= 0 || total_blocks <= 0 || total_blocks > 0x7FFFFFFFL) return LIC_ERR_INVALID;
  hipLaunchKernelGGL(prep_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs);
  return lic_check_launch();
}
