// lic_adam_run: the Adam update of EVERY parameter of a model in one launch.
//
// The reference trains with torch.optim.Adam (Main.ipynb; Trainer.py:81-86 calls optimizer.step()); on the GPU
// that is nine multi-tensor launches per step that read / write each of the four per-parameter streams several
// times (measured 0.25 ms for the 57 MB of config 2: 7 streams x 57 MB would take 0.07 ms at HBM speed).  This
// kernel does torch's arithmetic (non-amsgrad, L2 weight decay) for all parameters in one pass: block b looks
// up its job (parameter, gradient, exp_avg, exp_avg_sq, length) in a device-resident table, as lic_prep_run does.
//   g' = g + weight_decay * p;   m += (1 - beta1) * (g' - m);   v = beta2 * v + (1 - beta2) * g' * g';
//   p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps)
// The bias corrections are computed on the host from the step count, as torch's default (non-capturable) path does.
#include "lic_common.h"
#include <math.h>

namespace {
constexpr int AD_ITEMS = 4096;  // elements per block

// Gradient tensors are new allocations every step, so their addresses travel as kernel arguments (copied at
// launch time: no host -> device table copy that a host running several steps ahead could overwrite too early).
template <int NG>
struct AdamGrads {
  const float* g[NG];
};

template <int NG>
__global__ __launch_bounds__(256) void adam_kernel(const lic_adam_job* jobs, int njobs, const AdamGrads<NG> grads,
                                                   float lerp_w, float beta2, float one_minus_beta2, float eps,
                                                   float weight_decay, float step_size, float bc2_sqrt) {
  __shared__ lic_adam_job job;
  __shared__ int s_blk;
  __shared__ const float* s_g;
  if (threadIdx.x == 0) {
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].block0 <= b) lo = mid;
      else hi = mid - 1;
    }
    job = jobs[lo];
    s_blk = b - jobs[lo].block0;
    s_g = grads.g[lo];
  }
  __syncthreads();
  float* __restrict__ p = job.p;
  const float* __restrict__ g = s_g;
  float* __restrict__ m = job.m;
  float* __restrict__ v = job.v;
  const long n = job.n;
  const long begin = (long)s_blk * AD_ITEMS, end = begin + AD_ITEMS < n ? begin + AD_ITEMS : n;
  auto upd = [&](float& pw, float gw, float& mw, float& vw) {
    if (weight_decay != 0.0f) gw += weight_decay * pw;
    mw = lerp_w < 0.5f ? mw + lerp_w * (gw - mw) : gw - (gw - mw) * (1.0f - lerp_w);  // torch.lerp
    vw = vw * beta2;
    vw = vw + (one_minus_beta2 * gw) * gw;
    const float denom = sqrtf(vw) / bc2_sqrt + eps;
    pw = pw - step_size * (mw / denom);
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (vec) {
    const long e4 = begin + ((end - begin) & ~3L);
    for (long i = begin + threadIdx.x * 4L; i < e4; i += 256 * 4) {
      f32x4 pw = *reinterpret_cast<const f32x4*>(p + i), mw = *reinterpret_cast<const f32x4*>(m + i);
      f32x4 vw = *reinterpret_cast<const f32x4*>(v + i);
      const f32x4 gw = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pw[e], me = mw[e], ve = vw[e];
        upd(pe, gw[e], me, ve);
        pw[e] = pe;
        mw[e] = me;
        vw[e] = ve;
      }
      *reinterpret_cast<f32x4*>(p + i) = pw;
      *reinterpret_cast<f32x4*>(m + i) = mw;
      *reinterpret_cast<f32x4*>(v + i) = vw;
    }
    for (long i = e4 + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
  } else {
    for (long i = begin + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
  }
}
}  // namespace

// fills block0 / nblocks of a host-side job array; returns the grid size of lic_adam_run (or a negative status)
LIC_EXPORT int64_t lic_adam_plan(lic_adam_job* jobs, int32_t njobs) {
  if (!jobs || njobs <= 0) return LIC_ERR_INVALID;
  long blocks = 0;
  for (int i = 0; i < njobs; ++i) {
    if (!jobs[i].p || !jobs[i].m || !jobs[i].v || jobs[i].n <= 0) return LIC_ERR_INVALID;  // (g is passed per launch)
    jobs[i].nblocks = (int)((jobs[i].n + AD_ITEMS - 1) / AD_ITEMS);
    if (blocks + jobs[i].nblocks > 0x7FFFFFFFL) return LIC_ERR_UNSUPPORTED;
    jobs[i].block0 = (int)blocks;
    blocks += jobs[i].nblocks;
  }
  return blocks;
}

template <int NG>
static void adam_launch(const lic_adam_job* jobs_device, int njobs, long blocks, const float* const* grads, float lerp_w,
                        float beta2, float one_minus_beta2, float eps, float wd, float step_size, float bc2_sqrt,
                        hipStream_t s) {
  AdamGrads<NG> a;
  for (int i = 0; i < NG; ++i) a.g[i] = i < njobs ? grads[i] : nullptr;
  hipLaunchKernelGGL((adam_kernel<NG>), dim3((unsigned)blocks), dim3(256), 0, s, jobs_device, njobs, a, lerp_w, beta2,
                     one_minus_beta2, eps, wd, step_size, bc2_sqrt);
}

// grads_host: njobs gradient pointers (HOST array of DEVICE pointers), in job order; njobs <= 448
// (scalars are doubles: torch derives 1 - beta, lr / bias_correction1 and sqrt(bias_correction2) in Python doubles
// and rounds once to fp32; 1.0f - 0.999f is 1.7e-5 away from float(0.001))
LIC_EXPORT int lic_adam_run(const lic_adam_job* jobs_device, int32_t njobs, int64_t total_blocks,
                            const float* const* grads_host, double lr, double beta1, double beta2, double eps,
                            double weight_decay, double bias_correction1, double bias_correction2, lic_stream_t stream) {
  if (!jobs_device || !grads_host || njobs <= 0 || total_blocks <= 0 || total_blocks > 0x7FFFFFFFL) return LIC_ERR_INVALID;
  if (!(bias_correction1 > 0.0) || !(bias_correction2 > 0.0)) return LIC_ERR_INVALID;
  if (njobs > 448) return LIC_ERR_UNSUPPORTED;
  for (int i = 0; i < njobs; ++i)
    if (!grads_host[i]) return LIC_ERR_INVALID;
  const float lw = (float)(1.0 - beta1), ss = (float)(lr / bias_correction1), bs = (float)sqrt(bias_correction2);
  const float b2 = (float)beta2, omb2 = (float)(1.0 - beta2), ep = (float)eps, wd = (float)weight_decay;
  hipStream_t s = (hipStream_t)stream;
  if (njobs <= 96) adam_launch<96>(jobs_device, njobs, total_blocks, grads_host, lw, b2, omb2, ep, wd, ss, bs, s);
  else if (njobs <= 224) adam_launch<224>(jobs_device, njobs, total_blocks, grads_host, lw, b2, omb2, ep, wd, ss, bs, s);
  else adam_launch<448>(jobs_device, njobs, total_blocks, grads_host, lw, b2, omb2, ep, wd, ss, bs, s);
  return lic_check_launch();
}
