// fp32-MFMA implicit-GEMM kernels for gfx950 (MI355X): convolution family (lic_igemm) and the
// pixel-contraction weight gradient (lic_wgrad).  One wave owns 32x32 accumulator tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 64 cycles/instr/SIMD = the chip's fp32 peak);
// the kernels' job is to keep that pipe issuing: K is consumed in 16-deep chunks staged
// through LDS, the next chunk's global loads are issued before the current chunk's MFMAs, and
// 3-4 workgroups per CU cover each other's load/store phases.
#include "lic_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

thread_local int g_lic_last_hip_error = 0;

// division of 0 <= n < 2^31 by a launch-constant d via multiply-high (host precomputes m, s)
struct FastDiv {
  unsigned m, s;
};
static FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  if (d == 0) d = 1;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.s = s;
  f.m = (unsigned)(((1ull << (31 + s)) + d - 1) / d);
  return f;
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) {
  return (int)(((unsigned long long)(unsigned)n * f.m) >> (31 + f.s));
}

// ------------------------------------------------------------------------------------------------
// igemm: rows = output pixels (gathered from the NHWC input), cols = output channels.
//
// Workgroup = 4 waves as 2(M) x 2(N); wave tile = (BM/2) x (32*TN) in 32x32 MFMA tiles.
//  * A (activations): each 16-deep K chunk is gathered global -> registers -> LDS ([BM][16+4]
//    floats, conflict-free ds_read_b128 fragments), double-buffered so one barrier per chunk
//    suffices; the loads of chunk c+2 are issued before the MFMAs of chunk c.
//  * B (weights): never touches LDS.  lic_pack_weight lays them out
//    [tap][chunk][Npad/32][2][64 lanes][4] so that lane (col j, half h) of a wave reads its 8
//    K-values of a chunk as two 16-byte loads, each a contiguous 1 KiB per wave, straight into
//    the MFMA B-operand registers (L2-resident, zero-padded: no guards).
// Per chunk a wave issues TM*TN*8 MFMAs (up to 48 = 3072 cycles) against ~10 memory
// instructions, which keeps LDS (the limiter of the first version of this kernel) nearly idle.
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
  float* out3;
  long in_ld, out_ld, out2_ld, aux_ld, aux2_ld, aux3_ld, res_ld, out3_ld;
  int B, Hi, Wi, Cin, Ho, Wo, Cout;
  int kw, stride, pad, transposed, prologue, epilogue;
  float slope;
  int vec;     // float4 global loads of A legal
  int vec_out; // 16-byte epilogue accesses legal (Cout, every ld and pointer 16-byte friendly)
  int cpt;     // 16-deep chunks per tap
  int Npad;    // Cout rounded up to 32 (packed weight pitch)
  int nphase;  // 1, or stride^2 output phases of a transposed conv
  int ksplit;  // >1: K (the chunk list) is split across workgroups; raw partial sums go to `slabs`
  int cps;     // chunks per split
  float* slabs;  // [ksplit][B*Ho*Wo][Cout] fp32
  int MT, NT;  // tiles in M (max over phases) and N
  int bm_unfused;  // host-side note: the M tile the same launch would use without a fused GDN
  int pgroup;      // 4-phase launches: M tiles per phase-sorted group (0 = rotate phases tile by tile)
  int porder;      // phase ids by decreasing tap count, 2 bits each
  int ntaps[4];
  int Hq[4], Wq[4];
  FastDiv dHW[4], dW[4];  // divide by Hq*Wq and by Wq
  unsigned char taps[4][28];
};

constexpr int IG_BK = 16;
constexpr int IG_LDA = IG_BK + 4;

// 16 zero bytes: the source of LDS-DMA lanes that fall on padding / past the last pixel
__device__ __attribute__((aligned(16))) float g_lic_zero16[4];
typedef const __attribute__((address_space(1))) void* lic_gptr_t;
typedef __attribute__((address_space(3))) void* lic_lptr_t;

// FULLN: every wave of every workgroup has all its TN column tiles live (Npad % (64*TN) == 0), so
// the MFMA block is branch-free.  (With the scalar branches of the ragged variant hipcc shuffles
// accumulators through v_accvgpr_read/mov at the joins: ~2 extra AGPR moves per MFMA.)
// FUSE (BM = 64, one N tile covering all channels): the GDN / IGDN that follows the convolution
// runs in this kernel's epilogue -- see the block after the K loop.
// GLDS (VEC, FULLN, no prologue): both operands reach LDS by LDS-DMA (`global_load_lds_dwordx4`):
// no staging registers, no ds_write pass and no B registers held across a chunk; see the loop.
template <int BM, int TN, bool VEC, bool FULLN, bool FUSE = false, bool GLDS = false>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int BN = 64 * TN;
  constexpr int WM = BM / 2, WN = BN / 2;  // 2x2 waves
  constexpr int TM = WM / 32;              // MFMA tiles per wave in M
  constexpr int APASS = BM / 64;           // float4 A loads per thread per chunk
  // A double buffer; after the K loop the same memory stages 32x32 output tiles (one per wave)
  constexpr int GL_BUF = (BM + BN) * IG_BK;  // GLDS: floats of one (A tile, B panel) buffer
  constexpr int GL_NBUF = 3;  // ring of three (A tile, B panel) buffers, see the loop
  constexpr int SA_MAIN = GLDS ? GL_NBUF * GL_BUF : ((2 * BM * IG_LDA > 4 * 1024) ? 2 * BM * IG_LDA : 4 * 1024);
  static_assert(!GLDS || (VEC && FULLN && GL_NBUF * GL_BUF >= 4 * 1024), "LDS-DMA variant: float4 gathers, full N");
  // fused GDN epilogue: x tile [64][BN+4] + one 32x32 patch per wave
  constexpr int SA_FLOATS = (FUSE && 64 * (BN + 4) + 4096 > SA_MAIN) ? 64 * (BN + 4) + 4096 : SA_MAIN;
  static_assert(!FUSE || (VEC && FULLN), "fused GDN epilogue: float4 gathers, full N");
  __shared__ __attribute__((aligned(16))) float smem[SA_FLOATS + 32];  // + the decoded tap list
  float(*sA)[BM * IG_LDA] = reinterpret_cast<float(*)[BM * IG_LDA]>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // readfirstlane makes the wave index provably uniform: everything derived from it lives in
  // SGPRs and branches on it are scalar (otherwise hipcc wraps each MFMA group in exec-mask
  // branches and drains vmcnt to 0 around them)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware block remap (bijective): consecutive ids on one XCD share A rows / the weight panel
  const int nwg = gridDim.x;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // phases interleave under one M tile index so that every XCD's contiguous id range holds all
  // phases (their K lengths differ up to 2.25x: 9/6/6/4 taps for 5x5 stride 2)
  const int ks = wg % p.ksplit;  // K split fastest: the splits of a tile share its A rows in L2
  wg /= p.ksplit;
  const int nt = wg % p.NT;
  const int kq = wg / p.NT;
  // Rotate the phase order from one M tile to the next.  Phase durations differ (9/6/6/4 taps)
  // and the hardware deals consecutive workgroups round-robin over its shader engines / CUs:
  // with a fixed period-4 order one engine would receive only 9-tap workgroups and pace all the
  // others (measured: 1.2 instead of 1.9 resident waves per SIMD).
  int phase = 0, mt = kq;
  if (p.nphase == 4) {
    if (p.pgroup > 0) {
      // phases sorted by tap count inside groups of `pgroup` M tiles: an XCD's 64 slots run a round of
      // 9-tap workgroups, then the 6-tap ones, then the 4-tap ones -- homogeneous rounds, short tail
      // (all-equal neighbours also avoid the period-4 pattern described above)
      const int span = 4 * p.pgroup;
      const int grp = kq / span, loc = kq - grp * span;
      const int rank = loc / p.pgroup;
      phase = (p.porder >> (2 * rank)) & 3;
      mt = grp * p.pgroup + (loc - rank * p.pgroup);
    } else {
      phase = (kq + (kq >> 2) + (kq >> 4) + (kq >> 6) + (kq >> 8)) & 3;
      mt = kq / p.nphase;
    }
  }
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
    const int b = fdiv(pr, p.dHW[phase]);
    const int rem = pr - b * Hq * Wq;
    const int i = fdiv(rem, p.dW[phase]), jj = rem - i * Wq;
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
  int n_live = 0;  // live 32-column tiles of this wave (wave-uniform)
#pragma unroll
  for (int b = 0; b < TN; ++b) n_live += ((n0 + wn0 + b * 32) < p.Npad) ? 1 : 0;

  const int ntaps = p.ntaps[phase];
  // Tap list of this phase, decoded once into LDS as tap | r << 8 | s << 16.  (Indexing the kernarg
  // array per chunk compiles to a global_load_ubyte + `s_waitcnt vmcnt(0)` -- a memory round trip in
  // front of every chunk's loads, and a full drain of the DMA queue -- plus a division by kw.)
  // (kept INSIDE the one staging array: a second __shared__ object next to LDS-DMA buffers makes
  // hipcc wait vmcnt(0) before every LDS read)
  int* s_taps = reinterpret_cast<int*>(smem + SA_FLOATS);
  if (tid < 28) {
    const int t = p.taps[phase][tid < ntaps ? tid : 0];
    const int tr = t / p.kw;
    s_taps[tid] = t | (tr << 8) | ((t - tr * p.kw) << 16);
  }
  __syncthreads();
  const int c_first = ks * p.cps;  // this workgroup's slice of the (tap, channel-block) chunk list
  const int nchunks = max(0, min(ntaps * p.cpt, c_first + p.cps) - c_first);
  f32x4 ra[APASS];
  bool ra_ok[APASS];
  // prologue 2 (1x1, VEC): the A operand is t = dL/dnorm of a GDN, computed from three streams as
  // the chunk is staged, and written out once (out2) for the d-gamma / d-beta launches that follow --
  // the stand-alone lic_gdn_dnorm pass (3 reads + 1 write of the activation) disappears
  const bool pro2 = p.prologue == 2 || p.prologue == 3;
  const bool pro_inv = p.prologue == 3;
  f32x4 rx[APASS], rn[APASS];
  long rt_off[APASS];
  // this lane's B-operand address inside a (tap, chunk) panel: column n, K-half lh
  const float* wlane = p.w + ((long)((n0 + wn0) >> 5) * 512 + lane * 4);

  // Issue the global loads of one A chunk.  On the vector path nothing here consumes a loaded
  // value (out-of-range lanes load from a safe address and are zeroed when the registers are
  // written to LDS), so the loads stay in flight under the MFMAs.
  const int sgn = p.transposed ? -1 : 1;
  const int sh = (p.transposed && p.stride == 2) ? 1 : 0;  // stride is 1 or 2
  const int last_tap = ntaps - 1, last_cb = p.cpt - 1;
  auto load_a = [&](int tapi, int cb) {
    const bool past = tapi > last_tap;  // cursor ran past the end: harmless duplicate load
    const int code = __builtin_amdgcn_readfirstlane(s_taps[past ? last_tap : tapi]);
    const int r = (code >> 8) & 0xFF, s = code >> 16;
    const int ci = (past ? last_cb : cb) * IG_BK + a_c4;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int nh = a_hy[j] + sgn * r, nw = a_wx[j] + sgn * s;
      const int ih = nh >> sh, iw = nw >> sh;
      bool ok = a_ok[j] && nh >= 0 && nw >= 0 && ih < p.Hi && iw < p.Wi;
      if (VEC) {
        ok = ok && ci < p.Cin;
        // 32-bit selects (not a branch): the load must stay in the MFMA's basic block
        const int okm = -(int)ok;  // all-ones / zero mask: arithmetic keeps hipcc from branching
        const int pixi = (a_base[j] + ih * p.Wi + iw) & okm;
        const int cc = ci & okm;
        ra[j] = *reinterpret_cast<const f32x4*>(p.in + ((long)pixi * p.in_ld + cc));
        ra_ok[j] = ok;
        if (pro2) {  // wave-uniform: GDN backward, A = dL/dnorm built from (g = in, x = aux2, norm = aux3)
          rx[j] = *reinterpret_cast<const f32x4*>(p.aux2 + ((long)pixi * p.aux2_ld + cc));
          rn[j] = *reinterpret_cast<const f32x4*>(p.aux3 + ((long)pixi * p.aux3_ld + cc));
          rt_off[j] = (long)pixi * p.out2_ld + cc;
        }
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
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool sq = p.prologue == 1;
  auto store_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      f32x4 v = ra_ok[j] ? ra[j] : zero4;
      v = sq ? v * v : v;
      if (pro2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float rs = __builtin_amdgcn_rsqf(rn[j][e]);  // as lic_gdn_dnorm
          const float gx = ra[j][e] * rx[j][e];
          v[e] = pro_inv ? 0.5f * gx * rs : -0.5f * gx * rs * (rs * rs);
        }
        if (ra_ok[j]) {
          if (nt == 0) *reinterpret_cast<f32x4*>(p.out2 + rt_off[j]) = v;
        } else {
          v = zero4;
        }
      }
      *reinterpret_cast<f32x4*>(&sA[buf][((tid >> 2) + 64 * j) * IG_LDA + a_c4]) = v;
    }
  };
  // dead 32-column tiles (beyond Npad) re-read the wave's last live tile instead of branching
  int b_off[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) b_off[b] = (b < n_live ? b : (n_live > 0 ? n_live - 1 : 0)) * 512;
  if (n_live == 0) wlane = p.w + lane * 4;
  auto load_b = [&](f32x4 (&rb)[TN][2], int tapi, int cb) {
    const bool past = tapi > last_tap;
    const int tap = __builtin_amdgcn_readfirstlane(s_taps[past ? last_tap : tapi]) & 0xFF;
    const float* src = wlane + ((long)tap * p.cpt + (past ? last_cb : cb)) * p.Npad * IG_BK;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      rb[b][0] = *reinterpret_cast<const f32x4*>(src + b_off[b]);
      rb[b][1] = *reinterpret_cast<const f32x4*>(src + b_off[b] + 256);
    }
  };
  auto compute = [&](int buf, const f32x4 (&rb)[TN][2]) {
    f32x4 af[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const float* src = &sA[buf][(wm0 + a * 32 + li) * IG_LDA + lh * 8];
      af[a][0] = *reinterpret_cast<const f32x4*>(src);
      af[a][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
    // one scalar branch per 32-column tile (n_live is wave-uniform); a single code path keeps
    // every accumulator in one AGPR set
    if constexpr (!FULLN) __builtin_amdgcn_s_setprio(1);
    if constexpr (FULLN) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int a = 0; a < TM; ++a)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t >> 2][t & 3], rb[b][t >> 2][t & 3],
                                                              acc[a][b], 0, 0, 0);
    } else {
#pragma unroll
      for (int b = 0; b < TN; ++b)
        if (b < n_live) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int a = 0; a < TM; ++a)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t >> 2][t & 3], rb[b][t >> 2][t & 3],
                                                                acc[a][b], 0, 0, 0);
        }
    }
    if constexpr (!FULLN) __builtin_amdgcn_s_setprio(0);
  };

  // chunk cursor for the loads (runs ahead of the compute cursor)
  int l_tap = c_first / p.cpt, l_cb = c_first - l_tap * p.cpt;
  auto advance = [&]() {
    if (++l_cb == p.cpt) {
      l_cb = 0;
      ++l_tap;
    }
  };
  if constexpr (GLDS) {
    // LDS image of one buffer: A tile [BM][16] packed (64-byte rows; thread t of pass j owns row
    // t/4 + 64j, 16-byte slot t%4, i.e. byte 16*t + 4096*j -- the wave-linear order the DMA writes),
    // then the B panel of this N tile exactly as packed in memory ([32-col tile][q][lane][4]).
    // Packed 64-byte rows would put rows r, r+4, r+8, r+12 of a ds_read_b128 lane group on the same
    // banks, so slot s of row r holds K-quad s ^ ((r >> 2) & 3): the swizzle is applied to the
    // per-lane SOURCE address here and to the fragment reads below.
    const int gq = ((tid & 3) ^ ((tid >> 4) & 3)) * 4;  // this thread's logical channel offset in a chunk
    // Per-tap gather state: the pixel a row reads changes only with the tap (every cpt chunks); within a
    // tap a chunk moves 16 channels on, so the ~60 VALU of the full address computation run once per
    // tap instead of once per chunk (wave-uniform branch; `ks` splits may start mid-tap: t_tapi).
    long t_off[APASS];
    bool t_ok[APASS];
    int t_tap = 0, t_tapi = -1;
    auto issue = [&](int tapi, int cb, int buf) {
      const bool past = tapi > last_tap;
      const int cbb = past ? last_cb : cb;
      const int tcur = past ? last_tap : tapi;
      if (tcur != t_tapi) {  // a new tap begins
        t_tapi = tcur;
        const int code = __builtin_amdgcn_readfirstlane(s_taps[tcur]);
        const int r = (code >> 8) & 0xFF, s = code >> 16;
        t_tap = code & 0xFF;
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
          const int nh = a_hy[j] + sgn * r, nw = a_wx[j] + sgn * s;
          const int ih = nh >> sh, iw = nw >> sh;
          t_ok[j] = a_ok[j] && nh >= 0 && nw >= 0 && ih < p.Hi && iw < p.Wi;
          t_off[j] = t_ok[j] ? (long)(a_base[j] + ih * p.Wi + iw) * p.in_ld : 0L;
        }
      }
      const int tap = t_tap;
      const int ci = cbb * IG_BK + gq;
      const bool cok = ci < p.Cin;
      float* dstA = smem + buf * GL_BUF;
#pragma unroll
      for (int j = 0; j < APASS; ++j) {
        const bool ok = t_ok[j] && cok;
        const float* src = ok ? p.in + t_off[j] + ci : g_lic_zero16;
        __builtin_amdgcn_global_load_lds((lic_gptr_t)src, (lic_lptr_t)(dstA + j * 1024 + wave * 256), 16, 0, 0);
      }
      const float* wsrc = p.w + ((long)tap * p.cpt + cbb) * p.Npad * IG_BK + (long)n0 * IG_BK + tid * 4;
      float* dstB = dstA + BM * IG_BK;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        __builtin_amdgcn_global_load_lds((lic_gptr_t)(wsrc + j * 1024), (lic_lptr_t)(dstB + j * 1024 + wave * 256),
                                         16, 0, 0);
    };
    // Fragments of half a chunk (k-steps 4h .. 4h+3): TM + TN ds_read_b128 per lane.  The reads are inline
    // asm: left to hipcc, every fragment read is followed by `s_waitcnt lgkmcnt(0)` in front of the next MFMA
    // group whether that group uses it or not, which puts the LDS latency back in front of the matrix pipe.
    // With asm the waits are ours to place (LDS returns in order: lgkmcnt(TM + TN) leaves exactly the
    // youngest half-chunk's reads in flight).
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&smem[0];
    unsigned fa_addr[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int row = wm0 + a * 32 + li;
      const int sw = (row >> 2) & 3;
#pragma unroll
      for (int h = 0; h < 2; ++h) fa_addr[a][h] = lds0 + 4u * (unsigned)(row * IG_BK + (((lh * 2 + h) ^ sw) * 4));
    }
    const unsigned fb_addr = lds0 + 4u * (unsigned)(BM * IG_BK + (wn0 >> 5) * 512 + lane * 4);
    auto read_h = [&](auto bufc, auto hc, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
      constexpr int buf = decltype(bufc)::value, h = decltype(hc)::value;
#pragma unroll
      for (int a = 0; a < TM; ++a)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[a]) : "v"(fa_addr[a][h]), "i"(buf * GL_BUF * 4));
      if constexpr (TN >= 1)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[0]) : "v"(fb_addr), "i"(buf * GL_BUF * 4 + h * 1024));
      if constexpr (TN >= 2)
        asm volatile("ds_read_b128 %0, %1 offset:%2"
                     : "=v"(bf[TN >= 2 ? 1 : 0])
                     : "v"(fb_addr), "i"(buf * GL_BUF * 4 + 2048 + h * 1024));
      if constexpr (TN >= 3)
        asm volatile("ds_read_b128 %0, %1 offset:%2"
                     : "=v"(bf[TN >= 3 ? 2 : 0])
                     : "v"(fb_addr), "i"(buf * GL_BUF * 4 + 4096 + h * 1024));
    };
    // wait until at most `newer` LDS reads are in flight; the fragments pass through, so that the MFMAs that
    // use them cannot be scheduled above the wait
    auto arrive = [&](auto newerc, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(decltype(newerc)::value) : "memory");
#pragma unroll
      for (int a = 0; a < TM; ++a) asm volatile("" : "+v"(af[a]));
#pragma unroll
      for (int b = 0; b < TN; ++b) asm volatile("" : "+v"(bf[b]));
    };
    auto mfma_h = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int a = 0; a < TM; ++a)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
    };
    if (nchunks > 0) {
      // Ring of three buffers, fragments one half-chunk ahead of the MFMAs that use them.  At the top of step c
      // chunk c+1 (DMA issued a whole step earlier) has landed everywhere and everyone is done reading chunk
      // c-1, whose buffer takes chunk c+2.  The wave already holds the first-half fragments of chunk c; it
      // reads the second half, runs the first half's 4*TM*TN MFMAs, re-uses those registers for the first-half
      // fragments of chunk c+1, and runs the second half: no MFMA waits for the ds_read in front of it.
      // (Unrolled by three: buffer numbers are compile-time constants.  Past-the-end chunks are clamped
      // duplicates, so the prefetch past the last chunk reads valid data that is never used.)
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using NF = std::integral_constant<int, TM + TN>;
      f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
      issue(l_tap, l_cb, 0);
      advance();
      issue(l_tap, l_cb, 1);
      advance();
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(APASS + TN) : "memory");  // chunk 0 is in
      read_h(I0{}, I0{}, a0, b0);
      auto step = [&](auto cur, auto nxt, auto fill) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        read_h(cur, I1{}, a1, b1);
        arrive(NF{}, a0, b0);  // the first half is in (read a half-step ago); the second half stays in flight
        __builtin_amdgcn_sched_barrier(0);
        issue(l_tap, l_cb, decltype(fill)::value);  // its address arithmetic threads between the MFMAs below
        advance();
        mfma_h(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        read_h(nxt, I0{}, a0, b0);
        arrive(NF{}, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_h(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      };
      int c = 0;
      for (; c + 2 < nchunks; c += 3) {
        step(I0{}, I1{}, I2{});
        step(I1{}, I2{}, I0{});
        step(I2{}, I0{}, I1{});
      }
      if (c < nchunks) step(I0{}, I1{}, I2{});
      if (c + 1 < nchunks) step(I1{}, I2{}, I0{});
      arrive(I0{}, a0, b0);  // the last prefetch must land before its registers are reused
      __syncthreads();       // drains the duplicate tail DMAs; the epilogue reuses the buffers
    }
  } else {
  f32x4 rb0[TN][2], rb1[TN][2];
  if (nchunks > 0) {
    load_a(l_tap, l_cb);
    load_b(rb0, l_tap, l_cb);
    store_a(0);
    __syncthreads();
    advance();  // cursor -> chunk 1
    load_a(l_tap, l_cb);
    // Invariant at the top of an iteration (c even): sA[0] holds chunk c; `ra` holds chunk c+1
    // (in flight); rb0 holds B of chunk c; the cursor points at chunk c+1.  Loads past the last
    // chunk are clamped duplicates, so the body is branch-free with a single exit (anything else
    // makes hipcc copy the 96 accumulators between register sets every iteration).
    // Ask the scheduler to thread the next chunk's address arithmetic, global loads and LDS store
    // between this chunk's MFMAs (each 32x32x2 f32 MFMA leaves ~15 free issue slots).  Without
    // this the non-MFMA work sits in front of the burst, and the two waves sharing a SIMD -- which
    // interleave MFMA by MFMA and therefore run in lockstep -- idle the matrix pipe together.
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < TM * TN * 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);  // up to 4 VALU/SALU
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // up to 1 VMEM read
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // up to 1 DS write
      }
    };
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
      load_b(rb1, l_tap, l_cb);
      store_a(1);
      advance();
      load_a(l_tap, l_cb);
      compute(0, rb0);
      if constexpr (FULLN) interleave();
      __syncthreads();
      load_b(rb0, l_tap, l_cb);
      store_a(0);
      advance();
      load_a(l_tap, l_cb);
      compute(1, rb1);
      if constexpr (FULLN) interleave();
      __syncthreads();
    }
    if (c < nchunks) {  // odd tail: chunk c is in sA[0] / rb0
      compute(0, rb0);
      __syncthreads();
    }
  }
  }  // !GLDS


  // ---- fused GDN / IGDN (LIC_EPI_CONV_GDN / LIC_EPI_CONV_IGDN) ----------------------------------
  // The tile holds ALL channels of its 64 pixels, so the normalisation pool
  //   norm[p][i] = beta[i] + sum_j gamma[i][j] * x[p][j]^2,   y = x * rsqrt(norm)  (sqrt for IGDN)
  // is a second, short MFMA loop over the tile itself: x = acc + bias goes to LDS ([64][BN+4], the
  // +4 shifts consecutive rows by one 16-byte bank slot), each wave contracts its own 32 rows
  // against the packed gamma^T panel (p.aux, read from L2 exactly like the conv weights) in the
  // same chunk / k order as the stand-alone contraction launch, so the results are bitwise those of
  // conv -> gdn run as two kernels -- minus one full read of x and one launch per layer.
  if constexpr (FUSE) {
    constexpr int LDX = BN + 4;
    constexpr int NCH = BN / IG_BK;
    float* xt = smem;  // [64][LDX]: 32 rows per wave row; 128-row tiles take two passes (a = 0, 1)
    const int xr0 = (wave >> 1) * 32;
    const float* glane = p.aux + ((long)(wn0 >> 5) * 512 + lane * 4);
    const float* xrow = xt + (xr0 + li) * LDX + lh * 8;
    const int c4 = (lane & 7) * 4, r8 = lane >> 3;
    const bool inv = p.epilogue == LIC_EPI_CONV_IGDN;
    // (the K loop ended with a barrier: the A buffers are dead)
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if (a > 0) __syncthreads();  // the previous pass is done with the tile
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = wn0 + b * 32 + li;
        const float bv = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          xt[(xr0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDX + col] = acc[a][b][r] + bv;
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
      f32x4 g0[TN][2], g1[TN][2];
      auto load_g = [&](f32x4 (&rg)[TN][2], int c) {
        const float* src = glane + (long)(c < NCH ? c : NCH - 1) * p.Npad * IG_BK;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          rg[b][0] = *reinterpret_cast<const f32x4*>(src + b * 512);
          rg[b][1] = *reinterpret_cast<const f32x4*>(src + b * 512 + 256);
        }
      };
      auto pool = [&](int c, const f32x4 (&rg)[TN][2]) {
        f32x4 a0 = *reinterpret_cast<const f32x4*>(xrow + c * IG_BK);
        f32x4 a1 = *reinterpret_cast<const f32x4*>(xrow + c * IG_BK + 4);
        a0 = a0 * a0;
        a1 = a1 * a1;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32((t < 4 ? a0 : a1)[t & 3], rg[b][t >> 2][t & 3],
                                                              acc[a][b], 0, 0, 0);
      };
      load_g(g0, 0);
#pragma unroll 1
      for (int c = 0; c < NCH; c += 2) {
        load_g(g1, c + 1);
        pool(c, g0);
        load_g(g0, c + 2);
        pool(c + 1, g1);
      }
      // Finish in the 16-byte layout: each 32x32 pool tile goes through a wave-private LDS patch so
      // that a lane owns 4 consecutive channels of a row, meets x there, and stores x, norm and y.
      long opix[4];
      bool rok[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int prow = m0 + wm0 + a * 32 + it * 8 + r8;
        rok[it] = prow < P;
        const int pr = rok[it] ? prow : 0;
        if (p.nphase > 1) {
          const int bb = fdiv(pr, p.dHW[phase]);
          const int rem = pr - bb * Hq * Wq;
          const int i = fdiv(rem, p.dW[phase]), jj = rem - i * Wq;
          opix[it] = ((long)bb * p.Ho + i * sph + py) * p.Wo + jj * sph + px;
        } else {
          opix[it] = pr;
        }
      }
      float* stg = smem + 64 * LDX + wave * 1024;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[a][b][r];
        __builtin_amdgcn_wave_barrier();
        const int col = wn0 + b * 32 + c4;
        const f32x4 be4 = *reinterpret_cast<const f32x4*>(p.aux2 + col);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (!rok[it]) continue;
          const int rr = it * 8 + r8;
          const f32x4 n4 = *reinterpret_cast<const f32x4*>(&stg[rr * 32 + c4]) + be4;
          const f32x4 x4 = *reinterpret_cast<const f32x4*>(&xt[(xr0 + rr) * LDX + col]);
          f32x4 y4;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            y4[e] = x4[e] * (inv ? __builtin_amdgcn_sqrtf(n4[e]) : __builtin_amdgcn_rsqf(n4[e]));
          if (p.out3) *reinterpret_cast<f32x4*>(p.out3 + opix[it] * p.out3_ld + col) = x4;
          if (p.out2) *reinterpret_cast<f32x4*>(p.out2 + opix[it] * p.out2_ld + col) = n4;
          *reinterpret_cast<f32x4*>(p.out + opix[it] * p.out_ld + col) = y4;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    return;
  }

  // ---- epilogue -------------------------------------------------------------------------------
  if (p.ksplit > 1) {  // raw partial sums -> slab [ks][pixel][Cout]; igemm_finish_kernel does the rest
    float* slab = p.slabs + (long)ks * ((long)p.B * p.Ho * p.Wo) * p.Cout;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int prow = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (prow >= P) continue;
        long opix = prow;
        if (p.nphase > 1) {
          const int b = fdiv(prow, p.dHW[phase]);
          const int rem = prow - b * Hq * Wq;
          const int i = fdiv(rem, p.dW[phase]), jj = rem - i * Wq;
          opix = ((long)b * p.Ho + i * sph + py) * p.Wo + jj * sph + px;
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int col = n0 + wn0 + b * 32 + li;
          if (col < p.Cout) slab[opix * p.Cout + col] = acc[a][b][r];
        }
      }
    return;
  }
  const int epi = p.epilogue;
  if (p.vec_out) {
    // Stage each 32x32 accumulator tile through LDS so that every lane owns 4 consecutive
    // channels of a row: 16-byte loads/stores (1 KiB per wave instruction instead of 256 B; the
    // GDN layers with their short K loops are bound by this epilogue's store issue).
    float* stg = smem + wave * 1024;
    const int c4 = (lane & 7) * 4, r8 = lane >> 3;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        // the patch is wave-private (the K loop ended with a workgroup barrier): wave-level ordering only
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[a][b][r];
        __builtin_amdgcn_wave_barrier();
        const int col = n0 + wn0 + b * 32 + c4;
        if (col >= p.Cout) continue;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int rr = it * 8 + r8;
          const int prow = m0 + wm0 + a * 32 + rr;
          if (prow >= P) continue;
          long opix;
          if (p.nphase > 1) {
            const int bb = fdiv(prow, p.dHW[phase]);
            const int rem = prow - bb * Hq * Wq;
            const int i = fdiv(rem, p.dW[phase]), jj = rem - i * Wq;
            opix = ((long)bb * p.Ho + i * sph + py) * p.Wo + jj * sph + px;
          } else {
            opix = prow;
          }
          f32x4 v = *reinterpret_cast<const f32x4*>(&stg[rr * 32 + c4]) + bias4;
          if (epi == LIC_EPI_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : v[e] * p.slope;
            if (p.res)
              *reinterpret_cast<f32x4*>(p.out2 + opix * p.out2_ld + col) =
                  v + *reinterpret_cast<const f32x4*>(p.res + opix * p.res_ld + col);
          } else {
            if (epi == LIC_EPI_MUL_LEAKY_MASK) {
              const f32x4 m = *reinterpret_cast<const f32x4*>(p.aux + opix * p.aux_ld + col);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.0f ? v[e] : v[e] * p.slope;
            } else if (epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN) {
              if (p.out2) *reinterpret_cast<f32x4*>(p.out2 + opix * p.out2_ld + col) = v;
              const f32x4 x = *reinterpret_cast<const f32x4*>(p.aux + opix * p.aux_ld + col);
#pragma unroll
              for (int e = 0; e < 4; ++e)  // v_rsq_f32 / v_sqrt_f32: 1 ulp, far inside the 1e-4 bar
                v[e] = x[e] * ((epi == LIC_EPI_GDN) ? __builtin_amdgcn_rsqf(v[e]) : __builtin_amdgcn_sqrtf(v[e]));
            } else if (epi == LIC_EPI_GDN_BWD || epi == LIC_EPI_IGDN_BWD) {
              const f32x4 n = *reinterpret_cast<const f32x4*>(p.aux3 + opix * p.aux3_ld + col);
              const f32x4 g = *reinterpret_cast<const f32x4*>(p.aux + opix * p.aux_ld + col);
              const f32x4 x = *reinterpret_cast<const f32x4*>(p.aux2 + opix * p.aux2_ld + col);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float f = (epi == LIC_EPI_GDN_BWD) ? __builtin_amdgcn_rsqf(n[e]) : __builtin_amdgcn_sqrtf(n[e]);
                v[e] = g[e] * f + 2.0f * x[e] * v[e];
              }
            }
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + opix * p.res_ld + col);
          }
          *reinterpret_cast<f32x4*>(p.out + opix * p.out_ld + col) = v;
        }
      }
  } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int prow = m0 + row;
        if (prow >= P) continue;
        long opix;
        if (p.nphase > 1) {
          const int b = fdiv(prow, p.dHW[phase]);
          const int rem = prow - b * Hq * Wq;
          const int i = fdiv(rem, p.dW[phase]), jj = rem - i * Wq;
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
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// ---- weight packing: dst[tap][chunk][n/32][q][lane][4], zero padded to Npad = lic_npad_f32(N), K to 16.
// Lane (col li = lane&31, K-half lh = lane>>5) of a wave owns k = lh*8 + q*4 + e of column
// n = 32*tile + li, so each of its two B loads per 32-column tile reads lane*16 B of one
// contiguous 1 KiB block (fully coalesced, whole cache lines).
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* src, float* dst, int taps, int K,
                                                          int N, int cpt, int Npad, long s_tap, long s_k,
                                                          long s_n) {
  const long total = (long)taps * cpt * Npad * IG_BK;
  const int ntile = Npad >> 5;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63), q = (int)((i >> 8) & 1);
    long t = i >> 9;
    const int tile = (int)(t % ntile);
    t /= ntile;
    const int cb = (int)(t % cpt);
    const int tap = (int)(t / cpt);
    const int n = tile * 32 + (lane & 31);
    const int k = cb * IG_BK + (lane >> 5) * 8 + q * 4 + e;
    dst[i] = (k < K && n < N) ? src[tap * s_tap + k * s_k + n * s_n] : 0.0f;
  }
}
// The same layout for the strides conv weights have (taps innermost and contiguous, then one of the two
// channel dims): one workgroup per (16-k chunk, 8 columns) reads its source block along the contiguous
// direction into LDS (four loads in flight per lane) and writes 128-byte pieces of the packed blocks with
// 16-byte stores.  RUN_K: for a fixed n the (k, tap) run is contiguous (s_k == taps); else for a fixed k the
// (n, tap) run is (s_n == taps).
#define PK_NS 8
// V4: whole blocks (K % 16 == 0, N % 8 == 0) whose runs are 16-byte aligned are read with 16-byte loads --
// right after the optimizer step the weights are cold, and 4-byte gathers keep too few bytes in flight.
template <bool RUN_K, bool V4>
__global__ __launch_bounds__(256) void pack_weight_tiled_kernel(const float* src, float* dst, int taps, int K,
                                                                int N, int cpt, int ntile, long s_k, long s_n) {
  extern __shared__ float pk_st[];  // [taps][16][PK_NS + 1]
  const int cb = blockIdx.x % cpt, nb = blockIdx.x / cpt;  // nb: 8-column block
  const int k0 = cb * IG_BK, n0 = nb * PK_NS;
  const int run = (RUN_K ? IG_BK : PK_NS) * taps, total = IG_BK * PK_NS * taps;
  if (V4 && n0 >= N) {  // a whole block of the zero padding between N and ceil32(N) (N % 8 == 0 here)
    for (int i = threadIdx.x; i < taps * IG_BK * (PK_NS + 1); i += 256) pk_st[i] = 0.0f;
  } else if (V4) {
    for (int base = threadIdx.x; base < total / 4; base += 4 * 256) {
      f32x4 v[4];
      int jj[4], oo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = (base + u * 256) * 4;
        oo[u] = -1;
        if (idx < total) {
          const int o = idx / run, j = idx - o * run;
          oo[u] = o;
          jj[u] = j;
          v[u] = *reinterpret_cast<const f32x4*>(RUN_K ? src + (long)(n0 + o) * s_n + (long)k0 * taps + j
                                                       : src + (long)(k0 + o) * s_k + (long)n0 * taps + j);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (oo[u] >= 0) {
          int i = jj[u] / taps, tap = jj[u] - i * taps;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kl = RUN_K ? i : oo[u], nl = RUN_K ? oo[u] : i;
            pk_st[(tap * IG_BK + kl) * (PK_NS + 1) + nl] = v[u][e];
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
          const int o = idx / run, j = idx - o * run;  // o: outer index (n if RUN_K else k)
          const int i = j / taps, tap = j - i * taps;  // i: inner index (k if RUN_K else n)
          const int kl = RUN_K ? i : o, nl = RUN_K ? o : i;
          slot[u] = (tap * IG_BK + kl) * (PK_NS + 1) + nl;
          if (k0 + kl < K && n0 + nl < N)
            v[u] = RUN_K ? src[(long)(n0 + nl) * s_n + (long)k0 * taps + j]
                         : src[(long)(k0 + kl) * s_k + (long)n0 * taps + j];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (slot[u] >= 0) pk_st[slot[u]] = v[u];
    }
  }
  __syncthreads();
  const int tile = n0 >> 5, nin = n0 & 31;
  for (int idx = threadIdx.x; idx < taps * 32; idx += 256) {
    const int tap = idx >> 5, r = idx & 31, q = r >> 4, lh = (r >> 3) & 1, nl = r & 7;
    const int kl = lh * 8 + q * 4, lane = lh * 32 + nin + nl;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pk_st[(tap * IG_BK + kl + e) * (PK_NS + 1) + nl];
    *reinterpret_cast<f32x4*>(dst + (((long)tap * cpt + cb) * ntile + tile) * 512 + q * 256 + lane * 4) = o;
  }
}
LIC_EXPORT int64_t lic_packed_weight_floats(int32_t taps, int32_t K, int32_t N) {
  if (taps <= 0 || K <= 0 || N <= 0) return 0;
  return (int64_t)taps * ((K + IG_BK - 1) / IG_BK) * lic_npad_f32(N) * IG_BK;
}
LIC_EXPORT int lic_pack_weight(const float* src, float* dst, int32_t taps, int32_t K, int32_t N,
                               int64_t s_tap, int64_t s_k, int64_t s_n, lic_stream_t stream) {
  if (!src || !dst || taps <= 0 || K <= 0 || N <= 0) return LIC_ERR_INVALID;
  const int cpt = (K + IG_BK - 1) / IG_BK, Npad = lic_npad_f32(N);
  const long total = (long)taps * cpt * Npad * IG_BK;
  const int ntile = Npad / 32;
  const size_t lds = (size_t)taps * IG_BK * (PK_NS + 1) * sizeof(float);
  const int nblk = Npad / PK_NS;  // Npad is a multiple of 32
  const bool tiled = s_tap == 1 && taps >= 4 && taps <= 64 && aligned16(dst) && (long)cpt * nblk < 0x7FFFFFFFL &&
                     getenv("LIC_PACK_NO_TILED") == nullptr;
  // 16-byte source loads: whole blocks, every (outer index, block) run starting on a 16-byte boundary
  const bool v4 = tiled && K % IG_BK == 0 && N % PK_NS == 0 && aligned16(src) && (s_k == taps ? s_n : s_k) % 4 == 0;
#define LIC_PACK_LAUNCH(rk, v)                                                                                  \
  hipLaunchKernelGGL((pack_weight_tiled_kernel<rk, v>), dim3(cpt * nblk), dim3(256), lds, (hipStream_t)stream, src, \
                     dst, taps, K, N, cpt, ntile, (long)s_k, (long)s_n)
  if (tiled && s_k == taps) {
    if (v4) LIC_PACK_LAUNCH(true, true);
    else LIC_PACK_LAUNCH(true, false);
  } else if (tiled && s_n == taps) {
    if (v4) LIC_PACK_LAUNCH(false, true);
    else LIC_PACK_LAUNCH(false, false);
  }
#undef LIC_PACK_LAUNCH
  else
    hipLaunchKernelGGL(pack_weight_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, taps, K, N, cpt, Npad, (long)s_tap, (long)s_k, (long)s_n);
  return lic_check_launch();
}

// split-K finish: out = epilogue(sum_s slab[s] + bias) for the NONE / LEAKY epilogues
__global__ __launch_bounds__(256) void igemm_finish_kernel(const float* slabs, int ksplit, long npix, int Cout,
                                                           const float* bias, float* out, long out_ld,
                                                           int leaky, float slope) {
  const long total = npix * Cout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / Cout;
    const int c = (int)(i - pix * Cout);
    float v = 0.0f;
    for (int s = 0; s < ksplit; ++s) v += slabs[(long)s * total + i];
    if (bias) v += bias[c];
    if (leaky) v = v > 0.0f ? v : v * slope;
    out[pix * out_ld + c] = v;
  }
}
// the same, four channels per lane (Cout % 4 == 0, 16-byte aligned rows); slabs are summed in the same
// order, so both variants give identical bits
__global__ __launch_bounds__(256) void igemm_finish4_kernel(const float* slabs, int ksplit, unsigned npix,
                                                            unsigned C4, const float* bias, float* out,
                                                            long out_ld, int leaky, float slope) {
  const unsigned total4 = npix * C4;
  const long total = (long)total4 * 4;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
    const unsigned pix = i / C4;
    const unsigned c = (i - pix * C4) * 4;
    const float* sp = slabs + (long)i * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= ksplit; s += 4) {  // four loads in flight; summed in slab order
      const f32x4 a = *reinterpret_cast<const f32x4*>(sp + (long)s * total);
      const f32x4 b = *reinterpret_cast<const f32x4*>(sp + (long)(s + 1) * total);
      const f32x4 c2 = *reinterpret_cast<const f32x4*>(sp + (long)(s + 2) * total);
      const f32x4 d = *reinterpret_cast<const f32x4*>(sp + (long)(s + 3) * total);
      v += a;
      v += b;
      v += c2;
      v += d;
    }
    for (; s < ksplit; ++s) v += *reinterpret_cast<const f32x4*>(sp + (long)s * total);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
    if (leaky) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : v[e] * slope;
    }
    *reinterpret_cast<f32x4*>(out + (long)pix * out_ld + c) = v;
  }
}

// conv + GDN fusion needs all channels in one full N tile and float4 gathers
LIC_EXPORT int lic_igemm_fused_gdn_supported(int32_t Cin, int32_t Cout) {
  return (Cout == 64 || Cout == 128 || Cout == 192) && Cin > 0 && Cin % 4 == 0;
}

// K-split factor of a launch (1 = no split).  A function of per-image geometry only -- never of the batch
// size or of the tile -- so an image's result does not depend on which batch it is computed in (bitwise
// batch-split invariance: the split changes the summation order, the tile does not).
static long igemm_geo_split(const lic_igemm_desc* d, int Npad, int max_chunks, int epi, bool fuse) {
  const bool simple_epi = (epi == LIC_EPI_NONE || epi == LIC_EPI_LEAKY) && !d->res && !d->out2 && d->prologue < 2;
  const long t_img = (((long)d->Ho * d->Wo + 63) / 64) * ((Npad + 63) / 64);
  const char* env_split = d->force_split > 0 ? nullptr : getenv("LIC_IGEMM_FORCE_SPLIT");
  if (!(simple_epi && !fuse && d->workspace && (t_img < 40 || env_split || d->force_split > 1) && d->force_split != 1 &&
        (max_chunks >= 16 || d->force_split > 1)))
    return 1;
  long S = (40 + t_img - 1) / t_img;
  if (S > max_chunks / 8) S = max_chunks / 8;  // at least 8 chunks per split
  if (S > 32) S = 32;
  if (env_split) S = atoi(env_split);  // tuning aid
  if (d->force_split > 1) S = d->force_split < max_chunks ? d->force_split : max_chunks;
  return S > 1 ? S : 1;
}

// fills the kernel parameter block; returns LIC_OK, or 1 when there is nothing to launch
static int igemm_prepare(const lic_igemm_desc* d, IgemmParams& p, int& BM, int& TN, long& nwg,
                         int64_t& live_macs) {
  if (!d || !d->in || !d->w || !d->out) return LIC_ERR_INVALID;
  if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 ||
      d->Cout <= 0 || d->kh <= 0 || d->kw <= 0)
    return LIC_ERR_INVALID;
  if (d->kh * d->kw > 28 || d->stride < 1 || d->stride > 2) return LIC_ERR_UNSUPPORTED;
  if (!aligned16(d->w)) return LIC_ERR_INVALID;
  const int epi = d->epilogue;
  if ((epi == LIC_EPI_MUL_LEAKY_MASK || epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN) && !d->aux)
    return LIC_ERR_INVALID;
  if ((epi == LIC_EPI_GDN_BWD || epi == LIC_EPI_IGDN_BWD) && (!d->aux || !d->aux2 || !d->aux3))
    return LIC_ERR_INVALID;
  if (epi == LIC_EPI_LEAKY && d->res && !d->out2) return LIC_ERR_INVALID;
  if (d->prologue == 2 || d->prologue == 3) {  // GDN backward: 1x1, float4 path, t -> out2
    if (!d->aux2 || !d->aux3 || !d->out2) return LIC_ERR_INVALID;
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->transposed) return LIC_ERR_UNSUPPORTED;
    if (d->Cin % 4 || d->in_ld % 4 || d->aux2_ld % 4 || d->aux3_ld % 4 || d->out2_ld % 4 || !aligned16(d->in) ||
        !aligned16(d->aux2) || !aligned16(d->aux3) || !aligned16(d->out2))
      return LIC_ERR_UNSUPPORTED;
  }
  const bool fuse = (epi == LIC_EPI_CONV_GDN || epi == LIC_EPI_CONV_IGDN);
  if (fuse) {
    if (!d->aux || !d->aux2 || d->res || d->prologue) return LIC_ERR_INVALID;  // (out2 / out3 may be NULL: inference)
    if (!lic_igemm_fused_gdn_supported(d->Cin, d->Cout)) return LIC_ERR_UNSUPPORTED;
    if (!aligned16(d->aux)) return LIC_ERR_INVALID;
  }

  p.in = d->in;
  p.w = d->w;
  p.bias = d->bias;
  p.out = d->out;
  p.out2 = d->out2;
  p.aux = d->aux;
  p.aux2 = d->aux2;
  p.aux3 = d->aux3;
  p.res = d->res;
  p.out3 = fuse ? d->out3 : nullptr;
  p.out3_ld = d->out3_ld;
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
  p.vec = (d->Cin % 4 == 0) && (d->in_ld % 4 == 0) && aligned16(d->in);
  p.cpt = (d->Cin + IG_BK - 1) / IG_BK;
  p.Npad = lic_npad_f32(d->Cout);
  {
    auto okp = [](const void* q, int64_t ld) { return q == nullptr || (aligned16(q) && ld % 4 == 0); };
    p.vec_out = (d->Cout % 4 == 0) && okp(d->out, d->out_ld) && okp(d->out2, d->out2_ld) &&
                okp(d->aux, d->aux_ld) && okp(d->aux2, d->aux2_ld) && okp(d->aux3, d->aux3_ld) &&
                okp(d->res, d->res_ld) && okp(d->bias, 4);
    if (fuse) {  // aux / aux2 are the packed gamma panel and beta here, not activations
      p.vec_out = okp(d->out, d->out_ld) && okp(d->out2, d->out2_ld) && okp(d->out3, d->out3_ld);
      if (!p.vec || !p.vec_out) return LIC_ERR_UNSUPPORTED;
    }
  }
  const uint32_t mask = d->tap_mask ? d->tap_mask : 0xFFFFFFFFu;
  p.nphase = (p.transposed && d->stride > 1) ? d->stride * d->stride : 1;
  long maxP = 0;
  live_macs = 0;
  for (int ph = 0; ph < 4; ++ph) {
    p.ntaps[ph] = 0;
    p.Hq[ph] = p.Wq[ph] = 0;
    p.dHW[ph] = p.dW[ph] = make_fastdiv(1);
  }
  for (int ph = 0; ph < p.nphase; ++ph) {
    const int py = (p.nphase > 1) ? ph / d->stride : 0, px = (p.nphase > 1) ? ph % d->stride : 0;
    const int st = (p.nphase > 1) ? d->stride : 1;
    p.Hq[ph] = (d->Ho - py + st - 1) / st;
    p.Wq[ph] = (d->Wo - px + st - 1) / st;
    if (p.Hq[ph] < 0) p.Hq[ph] = 0;
    if (p.Wq[ph] < 0) p.Wq[ph] = 0;
    const long Pp = (long)d->B * p.Hq[ph] * p.Wq[ph];
    p.dHW[ph] = make_fastdiv((unsigned)(p.Hq[ph] * p.Wq[ph]));
    p.dW[ph] = make_fastdiv((unsigned)p.Wq[ph]);
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

  // Tile selection.  Dead 32-column tiles are skipped by the waves, so every BN wastes the same
  // MFMA work; prefer the widest N tile (activations are gathered once per N tile) as long as
  // the grid keeps >= 512 workgroups, else fall back towards small tiles for parallelism.
  int max_taps = 0;
  for (int ph = 0; ph < p.nphase; ++ph) max_taps = p.ntaps[ph] > max_taps ? p.ntaps[ph] : max_taps;
  static const int cand[6][2] = {{128, 3}, {64, 3}, {128, 2}, {64, 2}, {128, 1}, {64, 1}};
  int best = 5;
  long best_wg = -1;
  bool found = false;
  // Small layers get their K loop split across workgroups (below).  The split factor depends on per-image
  // geometry only -- never on the tile -- so it is known here, and a full-N tile may count its splits as
  // workgroups: the 16x16 latent layers then take 64x192 tiles (activations gathered once per 192 columns)
  // instead of 64x64 ones.
  int max_chunks0 = 0;
  for (int ph = 0; ph < p.nphase; ++ph) max_chunks0 = p.ntaps[ph] * p.cpt > max_chunks0 ? p.ntaps[ph] * p.cpt : max_chunks0;
  const long S_geo = (getenv("LIC_IGEMM_SPLIT_AWARE") && getenv("LIC_IGEMM_SPLIT_AWARE")[0] == '0') ? 1 : igemm_geo_split(d, p.Npad, max_chunks0, epi, fuse);
  // first pass: shapes whose N tiling comes out full (branch-free MFMA block); second: any
  for (int pass = 0; pass < 2 && !found; ++pass)
    for (int c = 0; c < 6; ++c) {
      const int bm = cand[c][0], tn = cand[c][1];
      if (!p.vec && !(bm == 64 && tn == 1)) continue;  // scalar-A variant exists for one shape only
      // short-K layers (GDN contraction, 1x1 convs: <= 16 chunks) spend most of a workgroup's life
      // in prologue + epilogue: smaller M tiles put more workgroups in flight to overlap them
      // (the K = 80 GEMMs of the RGB layers -- 5 chunks, pure streaming -- would be 13 % faster on 128-row tiles,
      // 225 vs 260 us; not taken: 0.3 % of the step, and two HBM-bound launches per step would share the kernel
      // name whose MFMA roofline the benchmark reports)
      if (bm == 128 && max_taps * p.cpt <= 16) continue;
      if (p.Npad < 64 * tn && tn > 1 && p.Npad <= 64 * (tn - 1)) continue;  // wider than the problem
      if (pass == 0 && p.Npad % (64 * tn) != 0) continue;
      long wgs = ((maxP + bm - 1) / bm) * ((p.Npad + 64 * tn - 1) / (64 * tn)) * p.nphase;
      if (pass == 0) wgs *= S_geo;
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
  BM = cand[best][0];
  TN = cand[best][1];
  if (d->force_bm || d->force_tn) {  // descriptor override (parity tests, the entropy coder's pinned variant)
    const int fb = d->force_bm, ft = d->force_tn;
    if (!((fb == 64 || fb == 128) && ft >= 1 && ft <= 3 && p.vec && p.Npad % (64 * ft) == 0)) return LIC_ERR_UNSUPPORTED;
    BM = fb;
    TN = ft;
  } else if (const char* e = getenv("LIC_IGEMM_FORCE_TILE")) {  // tuning aid: "bm,tn"
    int fb = 0, ft = 0;
    if (sscanf(e, "%d,%d", &fb, &ft) == 2 && (fb == 64 || fb == 128) && ft >= 1 && ft <= 3 && p.vec &&
        p.Npad % (64 * ft) == 0) {
      BM = fb;
      TN = ft;
    }
  }
  // One N tile spanning every channel, 64 rows.  (A 128-row fused variant was built and measured:
  // all workgroups of a launch reach their epilogue together, so the pool is not hidden behind other
  // workgroups' K loops, and the big layers came out 15-20 % slower than conv + a separate GDN
  // launch; lic_igemm_fused_gdn_preferred tells the caller when fusing pays.)
  const int BM_unfused = BM;
  if (fuse) {
    TN = p.Npad / 64;
    BM = 64;
  }
  p.bm_unfused = BM_unfused;
  p.NT = (p.Npad + 64 * TN - 1) / (64 * TN);
  p.MT = (int)((maxP + BM - 1) / BM);
  p.pgroup = 0;
  p.porder = 0;
  if (p.nphase == 4 && p.MT >= 128 && getenv("LIC_IGEMM_NO_PSORT") == nullptr) {
    int ord[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j)
        if (p.ntaps[ord[j]] > p.ntaps[ord[i]]) {
          const int t = ord[i];
          ord[i] = ord[j];
          ord[j] = t;
        }
    p.porder = ord[0] | (ord[1] << 2) | (ord[2] << 4) | (ord[3] << 6);
    p.pgroup = 64;
    p.MT = ((p.MT + 63) / 64) * 64;  // whole groups; the padding tiles exit at once
  }
  nwg = (long)p.MT * p.NT * p.nphase;
  // Small layers (the 16x16 / 8x8 / 4x4 latent side) cannot fill 256 CUs with output tiles alone:
  // split their K loop across workgroups when the caller provided a workspace.
  p.ksplit = 1;
  p.slabs = nullptr;
  int max_chunks = 0;
  for (int ph = 0; ph < p.nphase; ++ph) max_chunks = p.ntaps[ph] * p.cpt > max_chunks ? p.ntaps[ph] * p.cpt : max_chunks;
  p.cps = max_chunks > 0 ? max_chunks : 1;
  {
    const long S = igemm_geo_split(d, p.Npad, max_chunks, epi, fuse);
    if (S > 1) {
      p.cps = (int)((max_chunks + S - 1) / S);
      p.ksplit = (max_chunks + p.cps - 1) / p.cps;
      const size_t need = (size_t)p.ksplit * d->B * d->Ho * d->Wo * d->Cout * sizeof(float);
      if (d->workspace_bytes < need) return LIC_ERR_WORKSPACE;
      p.slabs = (float*)d->workspace;
      nwg *= p.ksplit;
    }
  }
  if (nwg > 0x7FFFFFFFL) return LIC_ERR_UNSUPPORTED;
  return LIC_OK;
}

LIC_EXPORT size_t lic_igemm_workspace_bytes(const lic_igemm_desc* d) {
  if (!d) return 0;
  lic_igemm_desc q = *d;  // plan with stand-in pointers and an unlimited workspace
  static float dummy[4] __attribute__((aligned(16)));
  q.in = q.w = dummy;
  q.out = dummy;
  q.bias = q.aux = q.aux2 = q.aux3 = q.res = nullptr;
  q.out2 = q.out3 = nullptr;
  if (q.epilogue != LIC_EPI_NONE && q.epilogue != LIC_EPI_LEAKY) return 0;
  q.workspace = dummy;
  q.workspace_bytes = ~(size_t)0;
  IgemmParams p;
  int bm = 0, tn = 0;
  long nwg = 0;
  int64_t macs = 0;
  if (igemm_prepare(&q, p, bm, tn, nwg, macs) != LIC_OK || p.ksplit <= 1) return 0;
  return (size_t)p.ksplit * d->B * d->Ho * d->Wo * d->Cout * sizeof(float);
}

// 1 when the fused conv+GDN launch is expected to beat conv followed by a GDN contraction launch:
// layers whose plain launch would use 64-row tiles anyway (measured on MI355X: RGB stem -12 %,
// 32x32 / 64x64-output layers -0..7 %; the 128-row-tile layers lose 15-20 % when fused)
LIC_EXPORT int lic_igemm_fused_gdn_preferred(const lic_igemm_desc* d) {
  if (!d || !lic_igemm_fused_gdn_supported(d->Cin, d->Cout)) return 0;
  lic_igemm_desc q = *d;
  static float dummy[4] __attribute__((aligned(16)));
  q.in = q.w = q.aux = q.aux2 = dummy;
  q.out = q.out2 = dummy;
  q.out3 = nullptr;
  q.bias = q.aux3 = q.res = nullptr;
  q.workspace = nullptr;
  q.workspace_bytes = 0;
  q.prologue = 0;
  q.epilogue = LIC_EPI_CONV_GDN;
  IgemmParams p;
  int bm = 0, tn = 0;
  long nwg = 0;
  int64_t macs = 0;
  if (igemm_prepare(&q, p, bm, tn, nwg, macs) != LIC_OK) return 0;
  return p.bm_unfused == 64 ? 1 : 0;
}

LIC_EXPORT int lic_igemm_plan(const lic_igemm_desc* d, int32_t* BM, int32_t* BN, int64_t* live_macs) {
  IgemmParams p;
  int bm = 0, tn = 0;
  long nwg = 0;
  int64_t macs = 0;
  const int rc = igemm_prepare(d, p, bm, tn, nwg, macs);
  if (rc < 0) return rc;
  if (BM) *BM = bm;
  if (BN) *BN = 64 * tn;
  if (live_macs) *live_macs = macs;
  return LIC_OK;
}

// name of the kernel variant lic_igemm launches for `d`, as rocprofv3 prints it (profiling aid)
LIC_EXPORT int lic_igemm_kernel_name(const lic_igemm_desc* d, char* buf, size_t n) {
  IgemmParams p;
  int BM = 0, TN = 0;
  long nwg = 0;
  int64_t macs = 0;
  const int rc = igemm_prepare(d, p, BM, TN, nwg, macs);
  if (rc < 0) return rc;
  if (!buf || n == 0) return LIC_ERR_INVALID;
  const bool full = (p.Npad % (64 * TN)) == 0;
  const bool fuse = p.epilogue == LIC_EPI_CONV_GDN || p.epilogue == LIC_EPI_CONV_IGDN;
  const bool glds = full && p.vec && p.prologue == 0 && getenv("LIC_IGEMM_NO_GLDS") == nullptr;
  if (!p.vec)
    snprintf(buf, n, "igemm_kernel<64, 1, false, false, false, false>");
  else
    snprintf(buf, n, "igemm_kernel<%d, %d, true, %s, %s, %s>", BM, TN, full ? "true" : "false",
             fuse ? "true" : "false", (glds || fuse) ? "true" : "false");
  return LIC_OK;
}

LIC_EXPORT int lic_igemm(const lic_igemm_desc* d, lic_stream_t stream) {
  IgemmParams p;
  int BM = 0, TN = 0;
  long nwg = 0;
  int64_t macs = 0;
  const int rc = igemm_prepare(d, p, BM, TN, nwg, macs);
  if (rc < 0) return rc;
  if (rc == 1) return LIC_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)nwg), block(256);
  const bool full = (p.Npad % (64 * TN)) == 0;
  // (prologues run on the register-staged loop: a select per operand element in the DMA loop's fragment
  // reads cost every launch ~30 VALU per chunk)
  const bool glds = full && p.prologue == 0 && getenv("LIC_IGEMM_NO_GLDS") == nullptr;
#define LIC_IGEMM_LAUNCH(bm, tn)                                                        \
  do {                                                                                  \
    if (glds)                                                                           \
      hipLaunchKernelGGL((igemm_kernel<bm, tn, true, true, false, true>), grid, block, 0, s, p); \
    else if (full)                                                                      \
      hipLaunchKernelGGL((igemm_kernel<bm, tn, true, true>), grid, block, 0, s, p);     \
    else                                                                                \
      hipLaunchKernelGGL((igemm_kernel<bm, tn, true, false>), grid, block, 0, s, p);    \
  } while (0)
  if (p.epilogue == LIC_EPI_CONV_GDN || p.epilogue == LIC_EPI_CONV_IGDN) {
    if (TN == 3)
      hipLaunchKernelGGL((igemm_kernel<64, 3, true, true, true, true>), grid, block, 0, s, p);
    else if (TN == 2)
      hipLaunchKernelGGL((igemm_kernel<64, 2, true, true, true, true>), grid, block, 0, s, p);
    else
      hipLaunchKernelGGL((igemm_kernel<64, 1, true, true, true, true>), grid, block, 0, s, p);
  } else if (!p.vec)  // odd channel counts / unaligned views: scalar-load variant, one tile shape
    hipLaunchKernelGGL((igemm_kernel<64, 1, false, false>), grid, block, 0, s, p);
  else if (BM == 128 && TN == 3)
    LIC_IGEMM_LAUNCH(128, 3);
  else if (BM == 64 && TN == 3)
    LIC_IGEMM_LAUNCH(64, 3);
  else if (BM == 128 && TN == 2)
    LIC_IGEMM_LAUNCH(128, 2);
  else if (BM == 64 && TN == 2)
    LIC_IGEMM_LAUNCH(64, 2);
  else if (BM == 128 && TN == 1)
    LIC_IGEMM_LAUNCH(128, 1);
  else
    LIC_IGEMM_LAUNCH(64, 1);
#undef LIC_IGEMM_LAUNCH
  if (p.ksplit > 1) {
    int rc2 = lic_check_launch();
    if (rc2 != LIC_OK) return rc2;
    const long npix = (long)d->B * d->Ho * d->Wo;
    const bool v4 = d->Cout % 4 == 0 && d->out_ld % 4 == 0 && aligned16(d->out) && aligned16(p.slabs) &&
                    (!d->bias || aligned16(d->bias)) && npix * d->Cout < 0x7FFFFFFFL;
    if (v4)
      hipLaunchKernelGGL(igemm_finish4_kernel, dim3(ew_grid(npix * d->Cout / 4, 256)), dim3(256), 0, s,
                         (const float*)p.slabs, p.ksplit, (unsigned)npix, (unsigned)(d->Cout / 4), d->bias, d->out,
                         (long)d->out_ld, d->epilogue == LIC_EPI_LEAKY ? 1 : 0, d->slope);
    else
      hipLaunchKernelGGL(igemm_finish_kernel, dim3(ew_grid(npix * d->Cout, 256)), dim3(256), 0, s,
                         (const float*)p.slabs, p.ksplit, npix, d->Cout, d->bias, d->out, (long)d->out_ld,
                         d->epilogue == LIC_EPI_LEAKY ? 1 : 0, d->slope);
  }
  return lic_check_launch();
}

// ------------------------------------------------------------------------------------------------
// wgrad: R[tap][m][n] = sum over small-grid pixels of row_operand[pix][m] * col_operand[pix][n]
//
// Both operands are activations (pixel-major, channels contiguous), so a 16-pixel K chunk of
// each is staged through LDS as [16][channels] (16-byte global loads, conflict-free ds_read_b32
// fragments).  Workgroup = 2x2 waves, wave tile = (32*TM) x (32*TN) accumulators; the big
// variant (TM=2,TN=3: 128x192, 48 MFMAs per wave per chunk) keeps a whole 192-channel operand
// in one tile.  LDS is double-buffered (one barrier per chunk) and the loads of chunk c+2 are in
// flight under the MFMAs of chunk c.  K (pixels) is split across workgroups; partial slabs are
// summed in a fixed order by wgrad_reduce_kernel (bitwise reproducible, no float atomics).
// ------------------------------------------------------------------------------------------------
struct WgOperand {
  const float* ptr;
  long ld;
  int C;
  int gathered;  // sample the large grid at (hs*stride-pad+r, ws*stride-pad+s)
  int sq;
};
struct WgradParams {
  WgOperand row, col;
  float* slabs;  // [splitk][ntaps][Cm][Cn]
  int B, Hs, Ws, Hl, Wl;
  int kw, stride, pad, ntaps;
  int MTt, NTt;  // tiles
  int chunks_per_split, nchunks;
  long Ps;
  FastDiv dHW, dW;  // divide by Hs*Ws and by Ws
};

constexpr int WG_BK = 16;

template <int TM, int TN, bool VEC, bool FULL>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int BMt = 64 * TM, BNt = 64 * TN;
  constexpr int WM = BMt / 2, WN = BNt / 2;
  constexpr int APASS = (WG_BK * BMt / 4) / 256, BPASS = (WG_BK * BNt / 4) / 256;
  __shared__ __attribute__((aligned(16))) float sA[2][WG_BK * BMt];
  __shared__ __attribute__((aligned(16))) float sB[2][WG_BK * BNt];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware bijective remap: the hardware deals workgroup ids round-robin over the 8 XCDs; give
  // each XCD a contiguous range of K splits, so the (tile, tap) workgroups that stream the same
  // pixel range share one L2 instead of pulling it through the fabric into all eight.
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

  // live 32-wide sub-tiles of this wave (wave-uniform)
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

  // Load slots: thread t owns pixel row kr = t/16 of every chunk and, within it, the float4 at
  // channel (t%16)*4 + 64*j of each operand -- one pixel decode per thread per chunk serves all
  // its loads, and 16 lanes cover 256 contiguous bytes.
  static_assert(APASS == TM && BPASS == TN, "one float4 per 64 channels per thread");
  const int kr = tid >> 4, c16 = (tid & 15) * 4;
  f32x4 ra[APASS], rb[BPASS];
  bool ra_ok[APASS], rb_ok[BPASS];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // chunk indices past the end are clamped to the last one (harmless duplicate loads)
  auto load_chunk = [&](int c) {
    const long pk = (long)(c < c_end ? c : c_end - 1) * WG_BK + kr;
    const bool inb = pk < p.Ps;
    const long pix = inb ? pk : 0;
    // gathered position of this pixel on the large grid (used by whichever operand is gathered)
    const int b = fdiv((int)pix, p.dHW);
    const int rem = (int)pix - b * p.Hs * p.Ws;
    const int hs = fdiv(rem, p.dW), ws = rem - hs * p.Ws;
    const int hl = hs * p.stride - p.pad + r, wl = ws * p.stride - p.pad + s;
    const bool gok = inb && hl >= 0 && wl >= 0 && hl < p.Hl && wl < p.Wl;
    const long gpix = ((long)b * p.Hl + hl) * p.Wl + wl;
    auto load_op = [&](const WgOperand& op, int ch, f32x4& v, bool& okr) {
      bool ok = op.gathered ? gok : inb;
      const long px = op.gathered ? gpix : pix;
      if (VEC) {
        ok = ok && ch < op.C;
        v = *reinterpret_cast<const f32x4*>(op.ptr + (ok ? px * op.ld + ch : 0L));
        okr = ok;
      } else {
        v = zero4;
        if (ok) {
          const float* src = op.ptr + px * op.ld + ch;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ch + e < op.C) v[e] = src[e];
        }
        okr = true;
      }
    };
#pragma unroll
    for (int j = 0; j < APASS; ++j) load_op(p.row, m0 + c16 + 64 * j, ra[j], ra_ok[j]);
#pragma unroll
    for (int j = 0; j < BPASS; ++j) load_op(p.col, n0 + c16 + 64 * j, rb[j], rb_ok[j]);
  };
  const bool sqa = p.row.sq != 0, sqb = p.col.sq != 0;
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      f32x4 v = ra_ok[j] ? ra[j] : zero4;
      v = sqa ? v * v : v;
      *reinterpret_cast<f32x4*>(&sA[buf][kr * BMt + c16 + 64 * j]) = v;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      f32x4 v = rb_ok[j] ? rb[j] : zero4;
      v = sqb ? v * v : v;
      *reinterpret_cast<f32x4*>(&sB[buf][kr * BNt + c16 + 64 * j]) = v;
    }
  };
  auto compute = [&](int buf) {
    float af[TM][8], bf[TN][8];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int t = 0; t < 8; ++t) af[a][t] = sA[buf][(lh * 8 + t) * BMt + wm0 + a * 32 + li];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int t = 0; t < 8; ++t) bf[b][t] = sB[buf][(lh * 8 + t) * BNt + wn0 + b * 32 + li];
    if constexpr (FULL) {  // all sub-tiles live everywhere: branch-free MFMA block
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
    } else {
#pragma unroll
      for (int a = 0; a < TM; ++a)
        if (a < m_live) {
#pragma unroll
          for (int b = 0; b < TN; ++b)
            if (b < n_live) {
#pragma unroll
              for (int t = 0; t < 8; ++t)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
            }
        }
    }
  };

  if (nloc > 0) {
    load_chunk(c_begin);
    store_chunk(0);
    __syncthreads();
    load_chunk(c_begin + 1);
    int c = 0;
    for (; c + 1 < nloc; c += 2) {  // branch-free body, single exit (see igemm_kernel)
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

// LDS-DMA variant of wgrad_kernel for full tiles: both operands go
// global -> LDS with `global_load_lds_dwordx4` (no staging registers, no ds_write pass, no vmcnt
// wait in front of an LDS store).  The LDS image is [64-channel sub-tile][16 px][64 ch]: thread t
// owns pixel t/16, channels (t%16)*4.. of every sub-tile, i.e. byte t*16 of it -- exactly the
// wave-linear destination the DMA writes.  Two buffers; each iteration is
//   __syncthreads (vmcnt(0): my DMA of chunk c landed; barrier: everyone's did, and everyone is
//   done reading chunk c-1)  ->  issue DMA of chunk c+1  ->  MFMAs of chunk c.
// SQB: the column operand is squared at the fragment read (GDN d-gamma: t^T . x^2).  A template flag, not a
// runtime select: 48 multiplies + 48 selects per chunk in every launch cost the plain ones ~5 %.
// FULL: every tile is complete (channel counts multiples of the tile); otherwise channels past C are
// DMA'd from the zero page and the slab writes are guarded (288- and 640-channel layers).
template <int TM, int TN, bool SQB = false, bool FULL = true>
__global__ __launch_bounds__(256, (TM * TN >= 6 ? 2 : 1)) void wgrad_glds_kernel(const WgradParams p) {
  constexpr int BMt = 64 * TM, BNt = 64 * TN, NS = TM + TN;
  constexpr int WM = BMt / 2, WN = BNt / 2;
  __shared__ __attribute__((aligned(16))) float smem[2][NS][WG_BK * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int li = lane & 31, lh = lane >> 5;
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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;

  typedef lic_gptr_t gptr_t;
  typedef lic_lptr_t lptr_t;
  const int kr = tid >> 4, c16 = (tid & 15) * 4;
  // chunk indices past the end are clamped to the last one (harmless duplicate DMA into the idle buffer)
  auto issue = [&](int c, int buf) {
    const long pk = (long)(c < c_end ? c : c_end - 1) * WG_BK + kr;
    const bool inb = pk < p.Ps;
    const long pix = inb ? pk : 0;
    const int b = fdiv((int)pix, p.dHW);
    const int rem = (int)pix - b * p.Hs * p.Ws;
    const int hs = fdiv(rem, p.dW), ws = rem - hs * p.Ws;
    const int hl = hs * p.stride - p.pad + r, wl = ws * p.stride - p.pad + s;
    const bool gok = inb && hl >= 0 && wl >= 0 && hl < p.Hl && wl < p.Wl;
    const long gpix = ((long)b * p.Hl + hl) * p.Wl + wl;
    const float* rowp = (p.row.gathered ? gok : inb)
                            ? p.row.ptr + (p.row.gathered ? gpix : pix) * p.row.ld + m0 + c16
                            : nullptr;
    const float* colp = (p.col.gathered ? gok : inb)
                            ? p.col.ptr + (p.col.gathered ? gpix : pix) * p.col.ld + n0 + c16
                            : nullptr;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const bool ok = rowp && (FULL || m0 + c16 + 64 * j < p.row.C);
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? rowp + 64 * j : g_lic_zero16),
                                       (lptr_t)&smem[buf][j][wave * 256], 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool ok = colp && (FULL || n0 + c16 + 64 * j < p.col.C);
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? colp + 64 * j : g_lic_zero16),
                                       (lptr_t)&smem[buf][TM + j][wave * 256], 16, 0, 0);
    }
  };
  auto compute = [&](int buf) {
    float af[TM][8], bf[TN][8];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int ch = wm0 + a * 32;
#pragma unroll
      for (int t = 0; t < 8; ++t) af[a][t] = smem[buf][ch >> 6][(lh * 8 + t) * 64 + (ch & 63) + li];
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int ch = wn0 + b * 32;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = smem[buf][TM + (ch >> 6)][(lh * 8 + t) * 64 + (ch & 63) + li];
        bf[b][t] = SQB ? v * v : v;
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
  };

  if (nloc > 0) {
    issue(c_begin, 0);
    int c = 0;
    for (; c + 1 < nloc; c += 2) {  // branch-free body, single exit (see igemm_kernel)
      __syncthreads();
      issue(c_begin + c + 1, 1);
      compute(0);
      __builtin_amdgcn_sched_barrier(0);  // keep the vmcnt(0)+barrier BEHIND this chunk's MFMAs
      __syncthreads();
      issue(c_begin + c + 2, 0);
      compute(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // drains the last (possibly duplicate) DMA before the buffers die
    if (c < nloc) compute(0);
  }
  float* slab = p.slabs + ((long)split * p.ntaps + tap) * p.row.C * p.col.C;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + wm0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
      if (!FULL && m >= p.row.C) continue;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn0 + b * 32 + li;
        if (FULL || n < p.col.C) slab[(long)m * p.col.C + n] = acc[a][b][q];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slabs, float* dst, int splitk,
                                                           int ntaps, int Cm, int Cn, long sm, long sn,
                                                           long stap, float scale) {
  const long total = (long)ntaps * Cm * Cn;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    // eight independent partial sums (fixed association order => still bitwise reproducible): with up to
    // 512 slabs and only ntaps*Cm*Cn threads, one dependent chain per thread was latency-bound (119 us
    // for the 75 MB of a GDN d-gamma launch)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= splitk; z += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a8[k] += slabs[(long)(z + k) * total + i];
    }
    for (; z < splitk; ++z) a8[0] += slabs[(long)z * total + i];
    const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    const int n = (int)(i % Cn);
    const long t2 = i / Cn;
    const int m = (int)(t2 % Cm);
    const int tap = (int)(t2 / Cm);
    dst[m * sm + n * sn + tap * stap] = acc * scale;
  }
}

// Split count for a split-K launch of `base` workgroups per split on `slots` resident workgroups:
// whole rounds of the machine.  (75 workgroups x 16 splits on 1024 slots ran a second, 17 %-full
// round; 13 splits fill one round to 95 %.)  Fewest rounds whose last one is >= 90 % full.
long lic_pick_splits(long base, long slots, long max_sk) {
  long best = 1;
  double best_eff = 0.0;
  if (max_sk > 512) max_sk = 512;
  if (max_sk < 1) max_sk = 1;
  for (long r = 1; r <= 4; ++r) {
    long sk = (r * slots) / base;
    if (sk < 1) continue;
    if (sk > max_sk) sk = max_sk;
    const long wgs = base * sk;
    const double eff = (double)wgs / ((double)((wgs + slots - 1) / slots) * (double)slots);
    if (eff > best_eff + 0.03) {
      best = sk;
      best_eff = eff;
    }
    if (eff >= 0.9 || sk == max_sk) break;
  }
  return best;
}

struct WgPlan {
  int TM, TN, vec, MTt, NTt, ntaps, nchunks, splitk, cps;
  int Cm, Cn;
};
static int wg_plan(const lic_wgrad_desc* d, WgPlan* pl) {
  if (!d || d->B <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Cp <= 0 || d->Cg <= 0 || d->kh <= 0 ||
      d->kw <= 0 || d->Hl <= 0 || d->Wl <= 0)
    return LIC_ERR_INVALID;
  pl->Cm = d->g_is_row ? d->Cg : d->Cp;
  pl->Cn = d->g_is_row ? d->Cp : d->Cg;
  pl->ntaps = d->kh * d->kw;
  pl->vec = (d->Cp % 4 == 0) && (d->p_ld % 4 == 0) && aligned16(d->p) && (d->Cg % 4 == 0) &&
            (d->g_ld % 4 == 0) && aligned16(d->g);
  // tile: 128x192 when both operands are wide, else 64-granular
  // 128-row tiles only when they come out full (a half-dead tile parks two of the four waves)
  // 128-row tiles when they come out full, for wide operands, and for 65..128 channels (one tile, each
  // operand streamed once: the RGB stem's 76-column weight gradient is pure bandwidth)
  pl->TM = (pl->vec && pl->Cm > 64 && (pl->Cm % 128 == 0 || pl->Cm > 256 || pl->Cm <= 128)) ? 2 : 1;
  pl->TN = (pl->vec && pl->Cn > 128) ? 3 : (pl->vec && pl->Cn > 64 ? 2 : 1);
  if (pl->TM == 1 && pl->TN == 2) pl->TN = 1;  // instantiated shapes: (2,3) (2,2) (2,1) (1,3) (1,1)
  // 192 x 192 tiles (LDS-DMA kernel only) when both channel counts are multiples of 192: half the
  // L2 -> LDS bytes per MFMA of the 64 x 192 tile
  const long chunks16 = ((long)d->B * d->Hs * d->Ws + 255) / 256;  // splits of >= 16 chunks available
  if (pl->vec && pl->Cm % 192 == 0 && pl->Cn % 192 == 0 && getenv("LIC_WGRAD_NO_GLDS") == nullptr &&
      getenv("LIC_WGRAD_NO_T33") == nullptr && chunks16 * pl->ntaps >= 512 &&  // else too few workgroups
      !(d->g_is_row ? d->sq_g : d->sq_p)) {  // (the DMA kernel squares the column operand only)
    pl->TM = 3;
    pl->TN = 3;
  }
  if (d->force_tm || d->force_tn) {  // descriptor override: one of the instantiated shapes
    const int tm = d->force_tm, tn = d->force_tn;
    const bool sq_row = d->g_is_row ? d->sq_g : d->sq_p;
    const bool t33 = tm == 3 && tn == 3 && pl->Cm % 192 == 0 && pl->Cn % 192 == 0 && !sq_row &&
                     getenv("LIC_WGRAD_NO_GLDS") == nullptr;
    const bool listed = (tm == 1 && (tn == 1 || tn == 3)) || (tm == 2 && tn >= 1 && tn <= 3);
    if (!pl->vec || !(t33 || listed)) return LIC_ERR_UNSUPPORTED;
    pl->TM = tm;
    pl->TN = tn;
  }
  pl->MTt = (pl->Cm + 64 * pl->TM - 1) / (64 * pl->TM);
  pl->NTt = (pl->Cn + 64 * pl->TN - 1) / (64 * pl->TN);
  const long Ps = (long)d->B * d->Hs * d->Ws;
  pl->nchunks = (int)((Ps + WG_BK - 1) / WG_BK);
  const long base = (long)pl->MTt * pl->NTt * pl->ntaps;
  // ~2.5 rounds of the 1024 resident workgroups, whole splits per XCD.  Measured on the big layers
  // (75 workgroups per split): 16 splits 2.47 ms, 40 splits 2.31 ms, 64 splits 2.28 ms -- shorter
  // workgroups even out the tail -- while the slab reduction grows by ~1.7 us per split; whole-round
  // counts that are not multiples of 8 (13 splits) lose the XCD-local L2 reuse and were no better.
  const long resident = 256L * (pl->TM == 1 ? 4 : 2);  // workgroups the chip holds at once
  long sk = (resident * 5 / 2 + base - 1) / base;
  const long max_sk = (pl->nchunks + 15) / 16;  // at least 16 chunks (256 pixels) per split
  if (sk > max_sk) sk = max_sk;
  if (sk < 1) sk = 1;
  if (sk > 256) sk = 256;
  if (sk > 8) sk = (sk + 7) & ~7L;  // whole splits per XCD (see the kernel's remap)
  // 128/192-row tiles (512 resident workgroups): whole rounds of the machine matter more than anything
  // else.  Measured on the 192x192 tile, 25 workgroups per split: 20 splits (0.98 rounds) 2.09 ms,
  // 32 (1.56) 2.55 ms, 40 (1.95) 2.06 ms, 56 (2.73) 2.21 ms, 61 (2.98) 2.08 ms -- so take the fewest
  // rounds that fill >= 90 %, which also keeps the slab reduction small.
  if (pl->TM >= 2) sk = lic_pick_splits(base, resident, max_sk);
  if (const char* e = getenv("LIC_WGRAD_SPLITS")) sk = atol(e) > 0 ? atol(e) : sk;  // tuning aid
  if (sk > max_sk) sk = max_sk;
  if (d->force_split > 0) sk = d->force_split < pl->nchunks ? d->force_split : pl->nchunks;  // tests: any split
  pl->cps = (int)((pl->nchunks + sk - 1) / sk);
  pl->splitk = (pl->nchunks + pl->cps - 1) / pl->cps;
  return LIC_OK;
}

LIC_EXPORT size_t lic_wgrad_workspace_bytes(const lic_wgrad_desc* d) {
  WgPlan pl;
  if (wg_plan(d, &pl) != LIC_OK) return 0;
  return (size_t)pl.splitk * pl.ntaps * pl.Cm * pl.Cn * sizeof(float);
}

static int wgrad_run(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, int stage,
                     lic_stream_t stream);
LIC_EXPORT int lic_wgrad(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes,
                         lic_stream_t stream) {
  return wgrad_run(d, workspace, workspace_bytes, 0, stream);
}
// stage 1: the partial-sum kernel only; stage 2: the slab reduction only (0 = both, as lic_wgrad).
// Lets a profiler time the MFMA kernel apart from the reduction.
LIC_EXPORT int lic_wgrad_stage(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes,
                               int32_t stage, lic_stream_t stream) {
  if (stage < 0 || stage > 2) return LIC_ERR_INVALID;
  return wgrad_run(d, workspace, workspace_bytes, stage, stream);
}
// the MFMA launch only; `job` receives the slab reduction for a later lic_reduce_batch (the workspace must live until then)
LIC_EXPORT int lic_wgrad_partial(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, lic_reduce_job* job,
                                 lic_stream_t stream) {
  if (!job) return LIC_ERR_INVALID;
  WgPlan pl;
  int rc = wg_plan(d, &pl);
  if (rc != LIC_OK) return rc;
  rc = wgrad_run(d, workspace, workspace_bytes, 1, stream);
  if (rc != LIC_OK) return rc;
  *job = lic_reduce_job{};
  job->src = (const float*)workspace;
  job->dst = d->dst;
  job->kind = LIC_REDUCE_SLABS;
  job->splitk = pl.splitk;
  job->ntaps = pl.ntaps;
  job->Cm = pl.Cm;
  job->Cn = pl.Cn;
  job->Mvalid = pl.Cm;
  job->Nvalid = pl.Cn;
  job->sm = d->dst_sm;
  job->sn = d->dst_sn;
  job->stap = d->dst_stap;
  job->scale = d->scale;
  return LIC_OK;
}
LIC_EXPORT int lic_wgrad_kernel_name(const lic_wgrad_desc* d, char* buf, size_t n) {
  WgPlan pl;
  const int rc = wg_plan(d, &pl);
  if (rc != LIC_OK) return rc;
  if (!buf || n == 0) return LIC_ERR_INVALID;
  const bool full = (pl.Cm % (64 * pl.TM) == 0) && (pl.Cn % (64 * pl.TN) == 0);
  if (!pl.vec)
    snprintf(buf, n, "wgrad_kernel<1, 1, false, false>");
  else if (!(d->g_is_row ? d->sq_g : d->sq_p) && getenv("LIC_WGRAD_NO_GLDS") == nullptr)
    snprintf(buf, n, "wgrad_glds_kernel<%d, %d, %s, %s>", pl.TM, pl.TN,
             (d->g_is_row ? d->sq_p : d->sq_g) ? "true" : "false", full ? "true" : "false");
  else
    snprintf(buf, n, "wgrad_kernel<%d, %d, true, %s>", pl.TM, pl.TN, full ? "true" : "false");
  return LIC_OK;
}
static int wgrad_run(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, int stage,
                     lic_stream_t stream) {
  WgPlan pl;
  int rc = wg_plan(d, &pl);
  if (rc != LIC_OK) return rc;
  if (!d->p || !d->g || !d->dst || !workspace) return LIC_ERR_INVALID;
  if (d->stride < 1) return LIC_ERR_UNSUPPORTED;
  const size_t need = (size_t)pl.splitk * pl.ntaps * pl.Cm * pl.Cn * sizeof(float);
  if (workspace_bytes < need) return LIC_ERR_WORKSPACE;
  if ((long)d->B * d->Hs * d->Ws > 0x7FFFFFFFL) return LIC_ERR_UNSUPPORTED;
  WgOperand P, G;
  P.ptr = d->p;
  P.ld = d->p_ld;
  P.C = d->Cp;
  P.gathered = 0;
  P.sq = d->sq_p;
  G.ptr = d->g;
  G.ld = d->g_ld;
  G.C = d->Cg;
  G.gathered = !(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->Hl == d->Hs &&
                 d->Wl == d->Ws);
  G.sq = d->sq_g;
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
  p.dHW = make_fastdiv((unsigned)(d->Hs * d->Ws));
  p.dW = make_fastdiv((unsigned)d->Ws);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(pl.MTt * pl.NTt * pl.ntaps * pl.splitk), block(256);
  const bool full = (pl.Cm % (64 * pl.TM) == 0) && (pl.Cn % (64 * pl.TN) == 0);
  const bool glds = !p.row.sq && getenv("LIC_WGRAD_NO_GLDS") == nullptr;  // (vec is checked below)
  if (stage != 2) {
#define LIC_WGRAD_LAUNCH(tm, tn)                                                      \
  do {                                                                                \
    if (glds && p.col.sq && full)                                                     \
      hipLaunchKernelGGL((wgrad_glds_kernel<tm, tn, true>), grid, block, 0, s, p);    \
    else if (glds && p.col.sq)                                                        \
      hipLaunchKernelGGL((wgrad_glds_kernel<tm, tn, true, false>), grid, block, 0, s, p); \
    else if (glds && full)                                                            \
      hipLaunchKernelGGL((wgrad_glds_kernel<tm, tn>), grid, block, 0, s, p);          \
    else if (glds)                                                                    \
      hipLaunchKernelGGL((wgrad_glds_kernel<tm, tn, false, false>), grid, block, 0, s, p); \
    else if (full)                                                                    \
      hipLaunchKernelGGL((wgrad_kernel<tm, tn, true, true>), grid, block, 0, s, p);   \
    else                                                                              \
      hipLaunchKernelGGL((wgrad_kernel<tm, tn, true, false>), grid, block, 0, s, p);  \
  } while (0)
  if (!pl.vec)
    hipLaunchKernelGGL((wgrad_kernel<1, 1, false, false>), grid, block, 0, s, p);
  else if (pl.TM == 3 && pl.TN == 3 && p.col.sq)
    hipLaunchKernelGGL((wgrad_glds_kernel<3, 3, true>), grid, block, 0, s, p);
  else if (pl.TM == 3 && pl.TN == 3)
    hipLaunchKernelGGL((wgrad_glds_kernel<3, 3>), grid, block, 0, s, p);
  else if (pl.TM == 2 && pl.TN == 3)
    LIC_WGRAD_LAUNCH(2, 3);
  else if (pl.TM == 2 && pl.TN == 2)
    LIC_WGRAD_LAUNCH(2, 2);
  else if (pl.TM == 2 && pl.TN == 1)
    LIC_WGRAD_LAUNCH(2, 1);
  else if (pl.TM == 1 && pl.TN == 3)
    LIC_WGRAD_LAUNCH(1, 3);
  else
    LIC_WGRAD_LAUNCH(1, 1);
#undef LIC_WGRAD_LAUNCH
  rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  }
  if (stage == 1) return LIC_OK;
  const long total = (long)pl.ntaps * pl.Cm * pl.Cn;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s,
                     (const float*)workspace, d->dst, pl.splitk, pl.ntaps, pl.Cm, pl.Cn, (long)d->dst_sm,
                     (long)d->dst_sn, (long)d->dst_stap, d->scale);
  return lic_check_launch();
}

// kernel-variant name lic_wgrad will launch for `d` (profiling aid)
LIC_EXPORT int lic_wgrad_plan(const lic_wgrad_desc* d, int32_t* TM, int32_t* TN, int32_t* splitk) {
  WgPlan pl;
  const int rc = wg_plan(d, &pl);
  if (rc != LIC_OK) return rc;
  if (TM) *TM = pl.vec ? pl.TM : 1;
  if (TN) *TN = pl.vec ? pl.TN : 1;
  if (splitk) *splitk = pl.splitk;
  return LIC_OK;
}

LIC_EXPORT int lic_version(void) { return LIC_ABI_VERSION; }
LIC_EXPORT int lic_last_hip_error(void) { return g_lic_last_hip_error; }
LIC_EXPORT const char* lic_arch(void) { return "gfx950"; }
