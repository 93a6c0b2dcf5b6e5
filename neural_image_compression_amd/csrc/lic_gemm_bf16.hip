// bf16-storage / fp32-accumulate variants of the implicit-GEMM kernels (BASELINE config 3) on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  igemm: the LDS-DMA structure of
// lic_gemm.hip's default loop (both operands global -> LDS by `global_load_lds_dwordx4`, 2x2 waves,
// LDS-staged 16-byte epilogue) with 32-deep chunks.  A chunk is only TM*TN*2 MFMAs of 32 cycles per
// wave -- shorter than a memory round trip -- so, as in the bf16 wgrad below, the DMA runs TWO
// chunks ahead through a ring of three LDS buffers (counted `s_waitcnt vmcnt`, raw `s_barrier`).
#include "lic_common.h"
#include <algorithm>
#include <type_traits>

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

constexpr int HB_BK = 32;  // bf16 elements per K chunk (64-byte rows, 2 MFMA K steps)

__device__ __attribute__((aligned(16))) float g_lic_zero16h[4];  // DMA source of padding / tail lanes
typedef const __attribute__((address_space(1))) void* lich_gptr_t;
typedef __attribute__((address_space(3))) void* lich_lptr_t;

struct IgemmHParams {
  const bf16_t* in;
  const bf16_t* w;  // packed [tap][cpt][Npad/32][2][64 lanes][8] bf16
  const float* bias;
  void* out;        // bf16 or fp32 (out_f32)
  bf16_t* out2;     // GDN norm (bf16)
  const bf16_t* aux;
  const bf16_t* aux2;
  const bf16_t* aux3;
  bf16_t* out3;        // FUSE: the convolution output before the normalisation (bf16), or null
  const float* beta;   // FUSE: beta_eff [Cout] (p.aux = gamma_eff^T packed like a 1x1 weight)
  long in_ld, out_ld, out2_ld, aux_ld, aux2_ld, aux3_ld, out3_ld;
  int B, Hi, Wi, Cin, Ho, Wo, Cout;
  int kw, stride, pad, transposed, prologue, epilogue, out_f32;
  float slope;  // LIC_EPI_LEAKY
  int cpt, Npad, nphase, MT, NT;
  int ksplit, cps;  // K split across workgroups (1 = none): chunks per split; fp32 partial tiles go to `slabs`
  float* slabs;     // [ksplit][B*Ho*Wo][Cout]
  int ring;         // host side only: the RING template argument of the launch
  int pgroup, porder;  // 4-phase launches: phase-sorted groups of `pgroup` M tiles, order 2 bits per rank
  int htx, hty;        // halo-resident variant (lic_halo_bf16.h): output tiles per image along x / y
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

// SQ: prologue 1 (GDN pool: the operand is x^2), squared at the fragment read.  A template flag: as a
// runtime condition hipcc computes the squares in every launch and selects (100 VALU per 12-MFMA chunk).
// FUSE: LIC_EPI_CONV_GDN / CONV_IGDN -- the tile holds every output channel of its pixels (NT == 1), so the
// workgroup that convolved them also pools them.  The four waves then sit side by side along M (32 pixels x all
// channels each) and run the MFMAs with the operands swapped, so the accumulators hold the TRANSPOSED tile: lane
// (li, lh) owns pixel li and, of every 32-channel tile, channels 4*lh + 8*g + j (g, j = 0..3).  Eight of those
// (g = 2s, 2s+1) are exactly one lane's share of a 16-deep MFMA B operand if K is counted in that order, which is
// how lic_pack_weight_bf16_kperm lays gamma_eff^T out: x -> bf16, x^2 -> bf16 and the pool's operand never leave
// the registers (no LDS transpose, no barrier), y = x * rsqrt(norm) is element-wise in the same layout, and one
// v_permlane32_swap per dword pair turns a lane's 4+4 channels into 8 consecutive ones for 16-byte stores.
// The element-wise part is written with 2-wide vectors (v_pk_add/mul_f32, v_cvt_pk_bf16_f32): at 2^26 outputs
// per launch of the first layer every VALU instruction per element is 1.7 us.
template <int... Is, class F>
__device__ __forceinline__ void lich_for_seq(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// RING: LDS buffers of the DMA ring (chunks RING-1 ahead).  Three for most variants; the 128 x 128 tile runs on
// four (64 KB: still two workgroups per CU): with a chunk taking 0.3-0.6 us of a workgroup's time, two chunks
// ahead is shorter than a loaded memory round trip.
//
// NWV = 8 (BM = 256, RING = 4, one workgroup per CU): the PING-PONG variant of the big layers.  Waves 0-3 own the
// upper 128 rows of the tile, waves 4-7 the lower 128 (each SIMD hosts one wave of either group), both share the
// weight panel, and the groups run HALF A CHUNK APART: a chunk is a load phase (wait for the DMA, read all
// fragments, issue the DMAs three chunks ahead, wait for the fragments) and an MFMA phase, a workgroup barrier
// after each -- while one group's waves occupy the matrix pipe the other group's do their LDS / address / DMA
// work on the same SIMDs, instead of two independent workgroups colliding at random (PMC of the 4-wave kernel on
// the 107-GFLOP layer: 37 % of wave cycles parked, 35 % issue-stalled).  Group g DMAs only its own 128 rows of A
// and half the weight panel.  Every load phase ends with the wave's wait for its DMAs of the NEXT chunk and the
// barrier, so a chunk is read one phase after the wait that retires it, whichever wave loaded the piece; a buffer is
// refilled only after both groups waited lgkmcnt(0) on their reads of it and passed a barrier.  Same chunk and k order per output as the 4-wave
// kernel: the results are bitwise the same.
template <int BM, int TN, bool SQ = false, bool FUSE = false, int RING = 3, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void igemm_bf16_kernel(const IgemmHParams p) {
  constexpr int BN = 64 * TN;
  constexpr int NTH = 64 * NWV;                       // threads
  constexpr int WGN = FUSE ? 1 : 2, WGM = NWV / WGN;  // wave grid
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TW = WN / 32;           // 32x32 MFMA tiles per wave
  static_assert(!FUSE || (BM == 32 * NWV && !SQ), "the fused pool gives every wave 32 rows of all channels");
  static_assert(NWV == 4 || (NWV == 8 && BM == 256 && RING == 4 && !SQ), "ping-pong variant: 256 rows, 8 waves");
  constexpr int APASS = BM / (NTH / 4);       // 16-byte DMA pieces per thread per A tile
  constexpr int BPASS = (256 * TN + NTH - 1) / NTH;  // ... per weight panel (NWV = 8: wrapped duplicates fill the last pass)
  constexpr int NL = APASS + BPASS;           // DMA instructions per thread per chunk
  constexpr int BUF = (BM + BN) * HB_BK;      // bf16 elements of one (A tile, B panel) buffer
  // three DMA buffers, later reused as the fp32 staging area of the epilogue (4 KiB per wave);
  // a plain 2-D array indexed with compile-time buffer numbers, so hipcc can tell the buffers apart
  constexpr int BUFP = (3 * BUF * 2 >= 4 * 4096) ? BUF : (4 * 4096 / 2 + 2) / 3;
  __shared__ __attribute__((aligned(16))) bf16_t smem_all[RING * BUFP + 64];  // + the decoded tap list
  auto bufp = [&](int b) { return smem_all + b * BUFP; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int li = lane & 31, lh = lane >> 5;

  const int nwg = gridDim.x;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int ksp = 0;
  if (p.ksplit > 1) {  // the splits of a tile are neighbours in launch order (they share its activations in L2)
    ksp = wg % p.ksplit;
    wg /= p.ksplit;
  }
  const int nt = wg % p.NT;
  const int kq = wg / p.NT;
  int phase = 0, mt = kq;
  if (p.nphase == 4) {
    if (p.pgroup > 0) {  // phases sorted by tap count inside groups of M tiles (see lic_gemm.hip)
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
  const int sph = (p.nphase > 1) ? p.stride : 1;
  const int py = (p.nphase > 1) ? phase / p.stride : 0;
  const int px = (p.nphase > 1) ? phase % p.stride : 0;

  // A tile image [BM][32] bf16, packed 64-byte rows: thread t of pass j owns row t/4 + 64j, 16-byte
  // slot t%4 = byte 16t + 4096j, the wave-linear order the DMA writes.  Rows r, r+4, r+8, r+12 of a
  // ds_read_b128 lane group would share banks, so slot s of row r holds K-octet s ^ ((r>>2)&3):
  // applied to the per-lane SOURCE address here and to the fragment reads below.
  const int gq = ((tid & 3) ^ ((tid >> 4) & 3)) * 8;  // this thread's logical channel offset in a chunk
  // NWV = 8: group g = wave / 4 loads (and reads) only rows 128 g .. 128 g + 127
  const int grp = (NWV == 8) ? (wave >> 2) : 0;
  const int arow0 = (NWV == 8) ? 128 * grp + ((tid & 255) >> 2) : (tid >> 2);
  int a_base[APASS], a_hy[APASS], a_wx[APASS];
  bool a_ok[APASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    const int prow = m0 + arow0 + 64 * j;
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

  f32x16 acc[TM][TW];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int ntaps = p.ntaps[phase];
  // tap list decoded once into LDS (tap | r << 8 | s << 16): a per-chunk kernarg lookup would be a
  // global load whose wait drains the whole in-order DMA queue
  // (kept INSIDE the one staging array: a second __shared__ object next to LDS-DMA buffers makes
  // hipcc wait vmcnt(0) before every LDS read)
  int* s_taps = reinterpret_cast<int*>(smem_all + RING * BUFP);
  if (tid < 28) {
    const int tt = p.taps[phase][tid < ntaps ? tid : 0];
    const int tr = tt / p.kw;
    s_taps[tid] = tt | (tr << 8) | ((tt - tr * p.kw) << 16);
  }
  __syncthreads();
  const int nch_all = ntaps * p.cpt;
  const int c_lo = ksp * p.cps < nch_all ? ksp * p.cps : nch_all;                       // this split's chunks
  const int nchunks = (p.ksplit > 1 ? (c_lo + p.cps < nch_all ? p.cps : nch_all - c_lo) : nch_all);
  const int sgn = p.transposed ? -1 : 1;
  const int sh = (p.transposed && p.stride == 2) ? 1 : 0;
  const int last_tap = ntaps - 1;
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_lic_zero16h);
  // Per-tap gather state: the pixel a row reads changes only when the tap does (every cpt chunks); inside a tap a
  // chunk just moves 32 channels on.  The DMA sources are therefore RUNNING POINTERS: a tap change (once per cpt
  // chunks) computes them in full (~70 VALU), every other chunk is one 64-bit add per piece -- the 46 SALU + 45
  // VALU instructions a chunk used to spend on addresses were as long as its 8 MFMAs, and in the 8-wave variant
  // they ARE the load phase the other group's MFMAs wait for.
  const bf16_t* a_ptr[APASS];   // this lane's 16-byte piece of row j (or the zero block)
  int a_inc[APASS];             // elements to move per chunk (0 for rows that read padding)
  const bf16_t* w_ptr = p.w;    // the chunk's weight panel (wave-uniform)
  const long w_inc = (long)p.Npad * HB_BK;
  int ci_cur = 0;               // first channel of this lane's piece (only consulted when Cin % 32 != 0)
  const bool cin_tail = (p.Cin & (HB_BK - 1)) != 0;
  int t_tap = -1;
#pragma unroll
  for (int j = 0; j < APASS; ++j) {
    a_ptr[j] = zsrc;
    a_inc[j] = 0;
  }
  auto issue = [&](int tapi, int cb, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    // (cursor past the end: the pointers stay on the last chunk -- a harmless duplicate DMA into the idle buffer)
    if (tapi <= last_tap) {
      if (cb == 0 || t_tap < 0) {  // wave-uniform: a new tap begins (or a split starts inside one)
        const int code = __builtin_amdgcn_readfirstlane(s_taps[tapi]);
        const int r = (code >> 8) & 0xFF, s = code >> 16;
        t_tap = code & 0xFF;
        ci_cur = cb * HB_BK + gq;
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
          const int nh = a_hy[j] + sgn * r, nw = a_wx[j] + sgn * s;
          const int ih = nh >> sh, iw = nw >> sh;
          const bool ok = a_ok[j] && nh >= 0 && nw >= 0 && ih < p.Hi && iw < p.Wi;
          a_ptr[j] = ok ? p.in + (long)(a_base[j] + ih * p.Wi + iw) * p.in_ld + ci_cur : zsrc;
          a_inc[j] = ok ? HB_BK : 0;
        }
        w_ptr = p.w + ((long)t_tap * p.cpt + cb) * w_inc + (long)n0 * HB_BK;
      } else {  // the next 32 channels of the same tap
        ci_cur += HB_BK;
#pragma unroll
        for (int j = 0; j < APASS; ++j) a_ptr[j] += a_inc[j];
        w_ptr += w_inc;
      }
    }
    bf16_t* dstA = bufp(buf);
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const bf16_t* src = a_ptr[j];
      if (cin_tail && ci_cur >= p.Cin) src = zsrc;  // (wave-uniform flag: Cin % 32 == 0 layers skip the compare)
      // (A image: 64-byte rows in row order; a wave's 64 lanes cover 16 consecutive rows = 1 KiB)
      bf16_t* da = (NWV == 8) ? dstA + 4096 * grp + 2048 * j + 512 * (wave & 3) : dstA + j * 2048 + wave * 512;
      __builtin_amdgcn_global_load_lds((lich_gptr_t)src, (lich_lptr_t)da, 16, 0, 0);
    }
    bf16_t* dstB = dstA + BM * HB_BK;
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      // 16-byte piece (j * NTH + tid) of the 256 TN pieces of the panel; past the end it wraps (wave-uniformly:
      // the wrap is a multiple of 64 pieces) onto pieces that are loaded twice with the same bytes
      int wv = j * NWV + wave;
      if (wv * 64 >= 256 * TN) wv -= 4 * TN;
      __builtin_amdgcn_global_load_lds((lich_gptr_t)(w_ptr + wv * 512 + lane * 8), (lich_lptr_t)(dstB + wv * 512), 16, 0, 0);
    }
  };
  // A step reads ALL its fragments first (TM*2 + TW*2 ds_read_b128), then issues the next chunk's DMAs -- ~40
  // scalar / vector address instructions -- and only then runs the MFMAs: the LDS latency hides under the address
  // arithmetic instead of standing in front of the first MFMA (left to itself hipcc placed each pair of reads
  // directly before its use, three exposed LDS round trips per chunk).
  bf16x8 af[TM][2], bf[TW][2];
  auto load_frags = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const bf16_t* bA = bufp(buf);
    const bf16_t* bB = bA + BM * HB_BK + (wn0 >> 5) * 1024 + lane * 8;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int row = wm0 + a * 32 + li;
      const int sw = (row >> 2) & 3;
#pragma unroll
      for (int q = 0; q < 2; ++q) af[a][q] = *reinterpret_cast<const bf16x8*>(bA + row * HB_BK + (((q * 2 + lh) ^ sw) * 8));
    }
#pragma unroll
    for (int b = 0; b < TW; ++b)
#pragma unroll
      for (int q = 0; q < 2; ++q) bf[b][q] = *reinterpret_cast<const bf16x8*>(bB + (b * 2 + q) * 512);
  };
  auto mfmas = [&]() {
    if constexpr (SQ) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) af[a][q] = sq8(af[a][q]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int b = 0; b < TW; ++b)
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          if constexpr (FUSE)  // transposed tile: channels down the rows, pixels across the lanes
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[b][q], af[a][q], acc[a][b], 0, 0, 0);
          else
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][q], bf[b][q], acc[a][b], 0, 0, 0);
        }
  };

  int l_tap = c_lo / p.cpt, l_cb = c_lo - (c_lo / p.cpt) * p.cpt;
  auto advance = [&]() {
    if (++l_cb == p.cpt) {
      l_cb = 0;
      ++l_tap;
    }
  };
  if (nchunks > 0) {
    // (A ring of six buffers -- DMAs five chunks ahead -- for the launches of one workgroup per CU changed
    // nothing: the last analysis layer 33 -> 36 us, the 3x3 hyper layer 15 -> 17 us.  Their 0.33 us per chunk
    // is not memory latency but one wave per SIMD walking ds_read -> wait -> MFMA -> barrier with nothing else
    // to issue; more workgroups per CU (the K split below) is what helps.)
    // Ring of RING buffers, unrolled by RING so that buffer indices are compile-time constants.
    // At the top of a step the DMAs of chunks c .. c+RING-2 are in flight: vmcnt((RING-2)*NL) retires mine of
    // chunk c, the barrier says everyone's landed and everyone finished reading chunk c-1, whose
    // buffer takes chunk c+RING-1 (past-the-end chunks are clamped duplicates, so the count is exact).
    static_assert(RING >= 3 && (RING - 2) * NL <= 63, "vmcnt is a 6-bit counter");
    if constexpr (NWV == 8) {
      // A chunk is read one phase AFTER the wait that retires it: every wave ends its load phase of chunk c by
      // waiting for its own DMAs of chunk c+1 (vmcnt leaves the two youngest chunks, c+2 and c+3, in flight) and
      // then passes the barrier -- so when either group starts reading chunk c+1, all eight waves' pieces of it
      // (a wave reads rows and panel pieces that OTHER waves loaded) have landed.  Waiting at the START of the
      // load phase instead, after the barrier, orders only a wave's own pieces: that version passed every
      // small test and produced NaNs in the two-stream training step.
      auto load_phase = [&](auto cur) {
        constexpr int CUR = decltype(cur)::value;
        load_frags(cur);
        __builtin_amdgcn_sched_barrier(0);
        issue(l_tap, l_cb, std::integral_constant<int, (CUR + RING - 1) % RING>{});
        advance();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"((RING - 2) * NL) : "memory");
      };
      auto chunk = [&](auto cur) {
        load_phase(cur);
        mfmas();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
      };
      lich_for_seq(std::make_integer_sequence<int, RING - 1>{}, [&](auto i) {
        issue(l_tap, l_cb, i);
        advance();
      });
      // chunk 0 has landed for everyone before the first load phase
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"((RING - 2) * NL) : "memory");
      // group 1 starts one phase late; group 0 idles one phase at the end: both execute 2 * nchunks + 2 barriers
      if (grp == 1) asm volatile("s_barrier" ::: "memory");
      int c = 0;
      for (; c + RING - 1 < nchunks; c += RING) lich_for_seq(std::make_integer_sequence<int, RING>{}, chunk);
      const int left = nchunks - c;
      lich_for_seq(std::make_integer_sequence<int, RING>{}, [&](auto i) {
        if (decltype(i)::value < left) chunk(i);
      });
      if (grp == 0) asm volatile("s_barrier" ::: "memory");
    } else {
    auto step = [&](auto cur) {
      constexpr int CUR = decltype(cur)::value;
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"((RING - 2) * NL) : "memory");
      load_frags(cur);
      __builtin_amdgcn_sched_barrier(0);
      issue(l_tap, l_cb, std::integral_constant<int, (CUR + RING - 1) % RING>{});
      advance();
      __builtin_amdgcn_sched_barrier(0);
      mfmas();
      __builtin_amdgcn_sched_barrier(0);
    };
    lich_for_seq(std::make_integer_sequence<int, RING - 1>{}, [&](auto i) {
      issue(l_tap, l_cb, i);
      advance();
    });
    int c = 0;
    for (; c + RING - 1 < nchunks; c += RING) lich_for_seq(std::make_integer_sequence<int, RING>{}, step);
    const int left = nchunks - c;
    lich_for_seq(std::make_integer_sequence<int, RING>{}, [&](auto i) {
      if (decltype(i)::value < left) step(i);
    });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the epilogue reuses the buffers
  }

  // ---- epilogue: stage each 32x32 fp32 tile through LDS; a lane then owns 8 consecutive
  // channels of a row (16-byte bf16 accesses; fp32 output writes two 16-byte halves) ------------
  const bool split = p.ksplit > 1;  // partial sums: raw fp32 to this split's slab, lic_igemm_bf16 finishes them
  const int epi = split ? (int)LIC_EPI_NONE : p.epilogue;
  const float* biasp = split ? nullptr : p.bias;
  void* const outp = split ? (void*)(p.slabs + (long)ksp * ((long)p.B * p.Ho * p.Wo) * p.Cout) : p.out;
  const long out_ld = split ? (long)p.Cout : p.out_ld;
  const bool of32 = split || p.out_f32;
  float* stg = reinterpret_cast<float*>(smem_all) + wave * 1024;
  const int c8 = (lane & 3) * 8, r16 = lane >> 2;
  if constexpr (FUSE) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto pack2 = [](f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); };
    // this lane's pixel (TM == 1) and where it lives in the output tensors
    const int prow = m0 + wm0 + li;
    const bool rok = prow < P;
    long opix = rok ? prow : 0;
    if (p.nphase > 1) {
      const int bb = fdivb((int)opix, p.dHW[phase]);
      const int rem = (int)opix - bb * Hq * Wq;
      const int i = fdivb(rem, p.dW[phase]), jj = rem - i * Wq;
      opix = ((long)bb * p.Ho + i * sph + py) * p.Wo + jj * sph + px;
    }
    // a lane's 16 channels of tile b as 8 bf16 pairs -> two 16-byte stores of 8 consecutive channels each
    auto store_tile = [&](bf16_t* base, long ld, const unsigned (&pk)[8], int b) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x2 r0 = __builtin_amdgcn_permlane32_swap(pk[4 * s], pk[4 * s + 2], false, false);
        const u32x2 r1 = __builtin_amdgcn_permlane32_swap(pk[4 * s + 1], pk[4 * s + 3], false, false);
        if (rok) {
          const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
          *reinterpret_cast<u32x4*>(base + opix * ld + b * 32 + 16 * s + 8 * lh) = v;
        }
      }
    };
    // 1. x = conv + bias, rounded to bf16 (what the backward pass reads); x^2 rounded again: the pool's operand
    unsigned sqpk[TW][8];
#pragma unroll
    for (int b = 0; b < TW; ++b) {
      unsigned xpk[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 bs = {0.0f, 0.0f, 0.0f, 0.0f};
        if (p.bias) bs = *reinterpret_cast<const f32x4*>(p.bias + b * 32 + 4 * lh + 8 * g);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 v = {acc[0][b][4 * g + 2 * h] + bs[2 * h], acc[0][b][4 * g + 2 * h + 1] + bs[2 * h + 1]};
          const unsigned pk = pack2(v);
          xpk[2 * g + h] = pk;
          const f32x2 xb = {__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
          acc[0][b][4 * g + 2 * h] = xb[0];
          acc[0][b][4 * g + 2 * h + 1] = xb[1];
          sqpk[b][2 * g + h] = pack2(xb * xb);
        }
      }
      if (p.out3) store_tile(p.out3, p.out3_ld, xpk, b);
    }
    // 2. norm^T = gamma_eff . (x^2)^T: A fragments of gamma_eff^T straight from L2 (16 bytes per lane, fragment order)
    f32x16 nacc[TW];
#pragma unroll
    for (int b = 0; b < TW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) nacc[b][r] = 0.0f;
    const bf16_t* gA = p.aux + lane * 8;
    const int ntile = p.Npad >> 5;
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4 bq = {sqpk[t][4 * s], sqpk[t][4 * s + 1], sqpk[t][4 * s + 2], sqpk[t][4 * s + 3]};
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, bq);
#pragma unroll
        for (int bo = 0; bo < TW; ++bo) {
          const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(gA + ((long)t * ntile + bo) * 1024 + s * 512);
          nacc[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, nacc[bo], 0, 0, 0);
        }
      }
    // 3. y = x * norm^-1/2 (GDN) or x * norm^1/2 (IGDN), element-wise in the accumulator layout
    auto finish = [&](auto inv) {
      constexpr bool INV = decltype(inv)::value;
#pragma unroll
      for (int b = 0; b < TW; ++b) {
        unsigned npk[8], ypk[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + b * 32 + 4 * lh + 8 * g);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 nv = {nacc[b][4 * g + 2 * h] + be[2 * h], nacc[b][4 * g + 2 * h + 1] + be[2 * h + 1]};
            npk[2 * g + h] = pack2(nv);
            const f32x2 f = {INV ? __builtin_amdgcn_sqrtf(nv[0]) : __builtin_amdgcn_rsqf(nv[0]),
                             INV ? __builtin_amdgcn_sqrtf(nv[1]) : __builtin_amdgcn_rsqf(nv[1])};
            const f32x2 xv = {acc[0][b][4 * g + 2 * h], acc[0][b][4 * g + 2 * h + 1]};
            ypk[2 * g + h] = pack2(xv * f);
          }
        }
        if (p.out2) store_tile(p.out2, p.out2_ld, npk, b);
        store_tile(reinterpret_cast<bf16_t*>(p.out), p.out_ld, ypk, b);
      }
    };
    if (epi == LIC_EPI_CONV_IGDN) finish(std::true_type{});
    else finish(std::false_type{});
    return;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TW; ++b) {
      __builtin_amdgcn_wave_barrier();  // wave-private patch
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[a][b][r];
      __builtin_amdgcn_wave_barrier();
      const int col = n0 + wn0 + b * 32 + c8;
      if (col >= p.Cout) continue;
      float bias8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias8[e] = biasp ? biasp[col + e] : 0.0f;
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
        if (epi == LIC_EPI_LEAKY) {  // hyper / entropy-parameter layers (Components.py:70-72,100-102; ParametersModels.py:23-25)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.0f ? v[e] : v[e] * p.slope;
        } else if (epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN) {
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
            // (explicit fma: left to the compiler, the 4- and 8-wave instantiations contracted this differently)
            v[e] = __builtin_fmaf(2.0f * (float)x[e], v[e], (float)g[e] * f);
          }
        }
        if (of32) {
          float* o = reinterpret_cast<float*>(outp) + opix * out_ld + col;
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
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(outp) + opix * out_ld + col) = ob;
        }
      }
    }
}

// out = epilogue(sum of the K-split slabs in slab order + bias), 8 channels per thread (C % 8 == 0)
__global__ __launch_bounds__(256) void igemm_bf16_finish_kernel(const float* slabs, int ksplit, long npix, int C8,
                                                               const float* bias, void* out, long out_ld, int out_f32,
                                                               int leaky, float slope) {
  const long total8 = npix * C8, total = total8 * 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
    const long pix = i / C8;
    const int c = (int)(i - pix * C8) * 8;
    const float* sp = slabs + i * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ksplit; ++s) {
      v0 += *reinterpret_cast<const f32x4*>(sp + (long)s * total);
      v1 += *reinterpret_cast<const f32x4*>(sp + (long)s * total + 4);
    }
    if (bias) {
      v0 += *reinterpret_cast<const f32x4*>(bias + c);
      v1 += *reinterpret_cast<const f32x4*>(bias + c + 4);
    }
    if (leaky) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = v0[e] > 0.0f ? v0[e] : v0[e] * slope;
        v1[e] = v1[e] > 0.0f ? v1[e] : v1[e] * slope;
      }
    }
    if (out_f32) {
      float* o = reinterpret_cast<float*>(out) + pix * out_ld + c;
      *reinterpret_cast<f32x4*>(o) = v0;
      *reinterpret_cast<f32x4*>(o + 4) = v1;
    } else {
      bf16x8 ob;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ob[e] = (bf16_t)v0[e];
        ob[4 + e] = (bf16_t)v1[e];
      }
      *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(out) + pix * out_ld + c) = ob;
    }
  }
}

#include "lic_halo_bf16.h"
#include "lic_halot_bf16.h"

#ifdef LIC_HALO_ABLATE
LIC_EXPORT int lic_halo_debug_read(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_halo_dbg), bytes);
}
#endif
static bool al16h(const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

LIC_EXPORT int lic_igemm_bf16_fused_gdn_supported(int32_t Cin, int32_t Cout) {
  return (Cout == 64 || Cout == 128 || Cout == 192) && Cin > 0 && Cin % 8 == 0;
}

// ---- weight packing to bf16: dst[tap][chunk][n/32][kstep][lane][8], zero padded (K to 32, N to 64).
// Lane (col = lane&31, h = lane>>5) of a wave owns k = 16*kstep + 8h + e of column 32*tile + col:
// the B fragment of one 32x32x16 MFMA is lane*16 B of one contiguous KiB.
__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float* src, bf16_t* dst, int taps, int K,
                                                               int N, int cpt, int Npad, long s_tap, long s_k,
                                                               long s_n, int kperm) {
  const long total = (long)taps * cpt * Npad * HB_BK;
  const int ntile = Npad >> 5;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), q = (int)((i >> 9) & 1);
    long t = i >> 10;
    const int tile = (int)(t % ntile);
    t /= ntile;
    const int cb = (int)(t % cpt);
    const int tap = (int)(t / cpt);
    const int n = tile * 32 + (lane & 31);
    const int k = cb * HB_BK + q * 16 + (kperm ? ((e >> 2) * 8 + (lane >> 5) * 4 + (e & 3)) : ((lane >> 5) * 8 + e));
    dst[i] = (bf16_t)((k < K && n < N) ? src[tap * s_tap + k * s_k + n * s_n] : 0.0f);
  }
}
LIC_EXPORT int64_t lic_packed_weight_bf16_elems(int32_t taps, int32_t K, int32_t N) {
  if (taps <= 0 || K <= 0 || N <= 0) return 0;
  return (int64_t)taps * ((K + HB_BK - 1) / HB_BK) * (((N + 63) / 64) * 64) * HB_BK;
}
static int pack_bf16_launch(const float* src, void* dst, int32_t taps, int32_t K, int32_t N, int64_t s_tap, int64_t s_k,
                            int64_t s_n, int kperm, lic_stream_t stream) {
  if (!src || !dst || taps <= 0 || K <= 0 || N <= 0) return LIC_ERR_INVALID;
  const int cpt = (K + HB_BK - 1) / HB_BK, Npad = ((N + 63) / 64) * 64;
  const long total = (long)taps * cpt * Npad * HB_BK;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, taps, K, N, cpt, Npad, (long)s_tap, (long)s_k, (long)s_n, kperm);
  return lic_check_launch();
}
LIC_EXPORT int lic_pack_weight_bf16(const float* src, void* dst, int32_t taps, int32_t K, int32_t N,
                                    int64_t s_tap, int64_t s_k, int64_t s_n, lic_stream_t stream) {
  return pack_bf16_launch(src, dst, taps, K, N, s_tap, s_k, s_n, 0, stream);
}
LIC_EXPORT int lic_pack_weight_bf16_kperm(const float* src, void* dst, int32_t taps, int32_t K, int32_t N,
                                          int64_t s_tap, int64_t s_k, int64_t s_n, lic_stream_t stream) {
  return pack_bf16_launch(src, dst, taps, K, N, s_tap, s_k, s_n, 1, stream);
}

// d uses the lic_igemm_desc layout; activation / aux / out2 pointers are bf16, `w` is the bf16
// packed weight, bias is fp32; out is bf16 unless out_f32.
// fills the kernel parameter block and picks the tile; returns LIC_OK, or 1 when there is nothing to launch
static int igemmh_prepare(const lic_igemm_desc* d, int32_t out_f32, IgemmHParams& p, int& BM, int& TN, long& nwg) {
  if (!d || !d->in || !d->w || !d->out) return LIC_ERR_INVALID;
  if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 ||
      d->kh <= 0 || d->kw <= 0)
    return LIC_ERR_INVALID;
  if (d->kh * d->kw > 28 || d->stride < 1 || d->stride > 2) return LIC_ERR_UNSUPPORTED;
  const int epi = d->epilogue;
  const bool fuse = (epi == LIC_EPI_CONV_GDN || epi == LIC_EPI_CONV_IGDN);
  if (!(epi == LIC_EPI_NONE || epi == LIC_EPI_LEAKY || epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN ||
        epi == LIC_EPI_GDN_BWD || epi == LIC_EPI_IGDN_BWD || fuse) || d->res)
    return LIC_ERR_UNSUPPORTED;
  if ((epi == LIC_EPI_GDN || epi == LIC_EPI_IGDN) && !d->aux) return LIC_ERR_INVALID;
  if (fuse) {  // aux = gamma_eff^T packed (taps 1, K = N = Cout), aux2 = beta_eff (fp32); the tile spans every channel
    if (!d->aux || !d->aux2 || d->prologue) return LIC_ERR_INVALID;
    if (!lic_igemm_bf16_fused_gdn_supported(d->Cin, d->Cout)) return LIC_ERR_UNSUPPORTED;
    if ((d->out3 && d->out3_ld % 8) || !al16h(d->out3)) return LIC_ERR_INVALID;
    if (out_f32) return LIC_ERR_UNSUPPORTED;
  }
  if ((epi == LIC_EPI_GDN_BWD || epi == LIC_EPI_IGDN_BWD) && (!d->aux || !d->aux2 || !d->aux3)) return LIC_ERR_INVALID;
  // bf16 path: 16-byte pieces everywhere -> channel counts and pitches multiples of 8
  if (d->Cin % 8 || d->Cout % 8 || d->in_ld % 8 || d->out_ld % 8 || (d->out2 && d->out2_ld % 8) ||
      (!fuse && ((d->aux && d->aux_ld % 8) || (d->aux2 && d->aux2_ld % 8))) || (d->aux3 && d->aux3_ld % 8))
    return LIC_ERR_UNSUPPORTED;
  if (!al16h(d->in) || !al16h(d->w) || !al16h(d->out) || !al16h(d->out2) || !al16h(d->aux) || !al16h(d->aux2) ||
      !al16h(d->aux3))
    return LIC_ERR_INVALID;
  p.in = (const bf16_t*)d->in;
  p.w = (const bf16_t*)d->w;
  p.bias = d->bias;
  p.out = d->out;
  p.out2 = (bf16_t*)d->out2;
  p.aux = (const bf16_t*)d->aux;
  p.aux2 = (const bf16_t*)d->aux2;
  p.aux3 = (const bf16_t*)d->aux3;
  p.out3 = fuse ? (bf16_t*)d->out3 : nullptr;
  p.beta = fuse ? d->aux2 : nullptr;
  p.out3_ld = d->out3_ld;
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
  p.slope = d->slope;
  p.cpt = (d->Cin + HB_BK - 1) / HB_BK;
  p.Npad = ((d->Cout + 63) / 64) * 64;  // whole 64-column wave pairs: every tile is full
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
  if (maxP <= 0) return 1;
  if (maxP > 0x7FFFFFFFL / 2) return LIC_ERR_UNSUPPORTED;
  // N tile: the widest of 192 / 128 / 64 columns that divides Npad; M tile: 128 rows while the grid
  // keeps >= 512 workgroups (or the descriptor's force_bm: parity tests at small sizes)
  TN = (p.Npad % 192 == 0) ? 3 : ((p.Npad % 128 == 0) ? 2 : 1);
  p.NT = p.Npad / (64 * TN);
  BM = (((maxP + 127) / 128) * p.NT * p.nphase >= 512) ? 128 : 64;
  if (d->force_bm) {
    if (d->force_bm != 64 && d->force_bm != 128 && d->force_bm != 256 && d->force_bm != 512) return LIC_ERR_UNSUPPORTED;
    if (d->force_bm != 512) BM = d->force_bm;
  }
  if (fuse) BM = 128;  // (the fused pool's wave layout)
  // K-split candidates (the tiny latent-side layers, see below) are decided from per-image geometry BEFORE the
  // batch-dependent tile choice, so that the choice can never switch a split off: an image's bits do not depend
  // on its batch (the 8-wave tile used to disable the split once the batch made it eligible)
  int max_chunks = 0;
  for (int ph = 0; ph < p.nphase; ++ph) max_chunks = p.ntaps[ph] * p.cpt > max_chunks ? p.ntaps[ph] * p.cpt : max_chunks;
  const bool simple = (epi == LIC_EPI_NONE || epi == LIC_EPI_LEAKY) && !d->out2 && d->prologue == 0;
  const long t_img = (((long)d->Ho * d->Wo + 63) / 64) * (p.Npad / 64);
  const bool split_candidate = simple && d->workspace && d->force_split != 1 &&
                               ((t_img < 4 && max_chunks >= 48) || d->force_split > 1) && max_chunks >= 2;
  // 256-row, 8-wave ping-pong variant (see the kernel): where it leaves at least one workgroup per CU
  {
    const char* e = getenv("LIC_BF16_PP");  // tuning aid: 0 = never
    const bool off = e && e[0] == '0';
    const long wgs256 = ((maxP + 255) / 256) * p.NT * p.nphase;
    // Where it pays (measured with the final, race-free hand-over; same box, two runs each): 192-channel layers --
    // config 2h 4720-4760 -> 4980-5000 img/s, the fused-pool launch 344 -> 270 us (the 4-wave fused variant is
    // stuck at one wave per SIMD there) -- but NOT 128-channel ones: config 3 7400 -> 7180 img/s, and the
    // fused 8x1 layout 135 -> 153 us (80 KB of fragment reads per chunk).
    const bool pays = TN == 3;
    if (p.prologue != 1 && ((d->force_bm == 256) || (!d->force_bm && !off && pays && wgs256 >= 256 && !split_candidate)))
      BM = 256;
    else if (d->force_bm == 256) BM = 128;  // (the squaring prologue has no 8-wave variant)
  }
  // Halo-resident variant (lic_halo_bf16.h; BM = 512 names it): the 5x5 stride-2 layers of the analysis / synthesis
  // stacks (Components.py:12,14,41,43 -- forward convolutions and the data gradients of the transposed ones) with
  // 128 output channels and at least four 8 x 32 output tiles per IMAGE -- a rule of per-image geometry only: this
  // variant sums K chunk-major, the others tap-major, and an image's bits must not depend on the batch it is
  // computed in (tests/test_gpu_fullsize.py).  force_bm = 512 forces it on every
  // launch it covers (parity tests on small shapes; other launches keep their automatic tile), any other force_bm
  // and LIC_BF16_HALO=0 keep the implicit-GEMM tiles.
  p.htx = p.hty = 0;
  {
    const bool shape_ok = !p.transposed && d->kh == 5 && d->kw == 5 && d->stride == 2 && d->pad == 2 &&
                          (d->tap_mask == 0 || (d->tap_mask & 0x1FFFFFFu) == 0x1FFFFFFu) && d->prologue == 0 &&
                          (((epi == LIC_EPI_NONE || epi == LIC_EPI_LEAKY) && !d->out2) || (fuse && !out_f32)) && d->Cin % 64 == 0 &&
                          d->Cout == 128 && d->Ho == (d->Hi - 1) / 2 + 1 &&
                          d->Wo == (d->Wi - 1) / 2 + 1 && (long)d->B * d->Hi * d->Wi * d->in_ld < 0x7FFFFFFFL;
    const int htx = (d->Wo + halo::TWD - 1) / halo::TWD, hty = (d->Ho + halo::TH - 1) / halo::TH;
    const char* e = getenv("LIC_BF16_HALO");  // tuning aid: 0 = never
    const bool off = e && e[0] == '0';
    // ... and the transposed layers (ConvTranspose2d forward, data gradient of the strided convolution): BM = 513,
    // tiles of 8 x 32 phase pixels (lic_halot_bf16.h)
    const bool shape_t = p.transposed && d->kh == 5 && d->kw == 5 && d->stride == 2 && d->pad == 2 &&
                         (d->tap_mask == 0 || (d->tap_mask & 0x1FFFFFFu) == 0x1FFFFFFu) && d->prologue == 0 &&
                         (((epi == LIC_EPI_NONE || epi == LIC_EPI_LEAKY) && !d->out2) || (fuse && !out_f32)) && d->Cin % 64 == 0 &&
                         d->Cout == 128 && d->Ho == 2 * d->Hi && d->Wo == 2 * d->Wi &&
                         (long)d->B * d->Hi * d->Wi * d->in_ld < 0x7FFFFFFFL;
    const int qtx = (d->Wi + halot::TWD - 1) / halot::TWD, qty = (d->Hi + halot::TH - 1) / halot::TH;
    // (at least 16 tiles per image here: a tile is 4 phases = 1024 output pixels, and with 4 tiles per image a
    // batch-32 launch has only 128 of them -- the 32 -> 64 layers measured 88 us against 44 us on the implicit GEMM)
    if (shape_t && (d->force_bm == 512 || (!d->force_bm && !off && qtx * qty >= 16))) {
      BM = 513;
      p.htx = qtx;
      p.hty = qty;
      p.MT = d->B * qtx * qty;
      p.NT = 1;
      p.pgroup = p.porder = 0;
      p.ksplit = 1;
      p.cps = 0;
      p.slabs = nullptr;
      p.ring = 0;
      nwg = p.MT;
      return LIC_OK;
    }
    if (shape_ok && (d->force_bm == 512 || (!d->force_bm && !off && htx * hty >= 4))) {
      BM = 512;
      p.htx = htx;
      p.hty = hty;
      p.MT = d->B * htx * hty;
      p.NT = 1;
      p.pgroup = p.porder = 0;
      p.ksplit = 1;
      p.cps = 0;
      p.slabs = nullptr;
      p.ring = 0;
      nwg = p.MT;
      return LIC_OK;
    }
  }
  if (d->force_tn && d->force_tn != TN) return LIC_ERR_UNSUPPORTED;
  p.MT = (int)((maxP + BM - 1) / BM);
  p.pgroup = 0;
  p.porder = 0;
  if (p.nphase == 4 && p.MT >= 128) {
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
    p.MT = ((p.MT + 63) / 64) * 64;
  }
  nwg = (long)p.MT * p.NT * p.nphase;
  // K split for layers far too small to fill the chip: the two 5x5 stride-2 layers of the hyper encoder
  // (Components.py:71-73) are 32 and 8 workgroups of 100 chunks each -- one latency-bound K loop per CU, 32 us
  // apiece; split 12 ways they take 12.5 + 5 us (partial fp32 tiles + igemm_bf16_finish_kernel).  Only where the
  // gain clearly exceeds the extra launch (measured: at 128 workgroups -- the last analysis layer, the 3x3 hyper
  // layer -- 33 -> 25 us and 15 -> 17 us).  A function of per-image geometry only, like lic_igemm's: the batch an
  // image is computed in does not change its bits.
  p.ksplit = 1;
  p.cps = 0;
  p.slabs = nullptr;
  {
    long S = 1;
    if (split_candidate && BM != 256) {
      S = (24 + t_img - 1) / t_img;
      if (S > max_chunks / 8) S = max_chunks / 8;  // at least 8 chunks (256 K) per split
      if (S > 32) S = 32;
      if (d->force_split > 1) S = d->force_split < max_chunks ? d->force_split : max_chunks;
    }
    if (S > 1) {
      p.cps = (int)((max_chunks + S - 1) / S);
      p.ksplit = (max_chunks + p.cps - 1) / p.cps;
      if (p.ksplit > 1) {
        const size_t need = (size_t)p.ksplit * d->B * d->Ho * d->Wo * d->Cout * sizeof(float);
        if (d->workspace_bytes < need || !al16h(d->workspace)) return LIC_ERR_WORKSPACE;
        p.slabs = (float*)d->workspace;
        nwg *= p.ksplit;
      } else {
        p.ksplit = 1;
      }
    }
  }
  if (nwg > 0x7FFFFFFFL) return LIC_ERR_UNSUPPORTED;
  {
    const char* e = getenv("LIC_BF16_RING");  // tuning aid: 3 = the three-buffer ring everywhere
    p.ring = (BM == 256) ? 4 : ((BM == 128 && TN <= 2 && p.prologue != 1 && !(e && e[0] == '3')) ? 4 : 3);
  }
  return LIC_OK;
}

LIC_EXPORT size_t lic_igemm_bf16_workspace_bytes(const lic_igemm_desc* d) {
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
  IgemmHParams p;
  int bm = 0, tn = 0;
  long nwg = 0;
  if (igemmh_prepare(&q, 0, p, bm, tn, nwg) != LIC_OK || p.ksplit <= 1) return 0;
  return (size_t)p.ksplit * d->B * d->Ho * d->Wo * d->Cout * sizeof(float);
}

LIC_EXPORT int lic_igemm_bf16_kernel_name(const lic_igemm_desc* d, char* buf, size_t n) {
  IgemmHParams p;
  int BM = 0, TN = 0;
  long nwg = 0;
  const int rc = igemmh_prepare(d, 0, p, BM, TN, nwg);
  if (rc < 0) return rc;
  if (!buf || n == 0) return LIC_ERR_INVALID;
  if (BM == 513) {
    snprintf(buf, n, "halo_convt_bf16_kernel<%d, %s>", p.Npad / 64,
             (p.epilogue == LIC_EPI_CONV_GDN || p.epilogue == LIC_EPI_CONV_IGDN) ? "true" : "false");
    return LIC_OK;
  }
  if (BM == 512) {
    snprintf(buf, n, "halo_conv_bf16_kernel<%d, %s, 0>", p.Npad / 64,
             (p.epilogue == LIC_EPI_CONV_GDN || p.epilogue == LIC_EPI_CONV_IGDN) ? "true" : "false");
    return LIC_OK;
  }
  const bool fuse = p.epilogue == LIC_EPI_CONV_GDN || p.epilogue == LIC_EPI_CONV_IGDN;
  // (all five template arguments, as rocprofv3 prints them)
  snprintf(buf, n, BM == 256 ? "igemm_bf16_kernel<%d, %d, %s, %s, %d, 8>" : "igemm_bf16_kernel<%d, %d, %s, %s, %d, 4>", BM, TN,
           p.prologue == 1 ? "true" : "false", fuse ? "true" : "false", p.ring);
  return LIC_OK;
}

LIC_EXPORT int lic_igemm_bf16(const lic_igemm_desc* d, int32_t out_f32, lic_stream_t stream) {
  IgemmHParams p;
  int BM = 0, TN = 0;
  long nwg = 0;
  const int rc = igemmh_prepare(d, out_f32, p, BM, TN, nwg);
  if (rc < 0) return rc;
  if (rc == 1) return LIC_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)nwg), block(256);
  const bool fuse = p.epilogue == LIC_EPI_CONV_GDN || p.epilogue == LIC_EPI_CONV_IGDN;
  if (BM == 512 || BM == 513) {  // halo-resident 5x5 stride-2 variants: persistent workgroups, one per CU
    static int ncu = 0;
    if (!ncu) {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;
      ncu = n;
    }
    grid = dim3((unsigned)(nwg < ncu ? nwg : ncu));
#ifdef LIC_HALO_ABLATE
    if (const char* e = getenv("LIC_HALO_ABL")) {
      const int a = atoi(e);
      if (a == 1) hipLaunchKernelGGL((halo_conv_bf16_kernel<2, false, 1>), grid, block, 0, s, p);
      if (a == 2) hipLaunchKernelGGL((halo_conv_bf16_kernel<2, false, 2>), grid, block, 0, s, p);
      if (a == 4) hipLaunchKernelGGL((halo_conv_bf16_kernel<2, false, 4>), grid, block, 0, s, p);
      if (a == 3) hipLaunchKernelGGL((halo_conv_bf16_kernel<2, false, 3>), grid, block, 0, s, p);
      if (a == 7) hipLaunchKernelGGL((halo_conv_bf16_kernel<2, false, 7>), grid, block, 0, s, p);
      if (a) return lic_check_launch();
    }
#endif
    // (a 192-channel instance, 128 x 96 per wave, compiles but needs more than the 256 + 256 registers: hipcc moves
    // fragments that are still in flight; not dispatched)
    if (BM == 513) {
      if (fuse) hipLaunchKernelGGL((halo_convt_bf16_kernel<2, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((halo_convt_bf16_kernel<2>), grid, block, 0, s, p);
      return lic_check_launch();
    }
    if (fuse) hipLaunchKernelGGL((halo_conv_bf16_kernel<2, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((halo_conv_bf16_kernel<2>), grid, block, 0, s, p);
    return lic_check_launch();
  }
  if (BM == 256) {  // 8-wave ping-pong variant
    dim3 block8(512);
    if (fuse) {
      if (TN == 3) hipLaunchKernelGGL((igemm_bf16_kernel<256, 3, false, true, 4, 8>), grid, block8, 0, s, p);
      else if (TN == 2) hipLaunchKernelGGL((igemm_bf16_kernel<256, 2, false, true, 4, 8>), grid, block8, 0, s, p);
      else hipLaunchKernelGGL((igemm_bf16_kernel<256, 1, false, true, 4, 8>), grid, block8, 0, s, p);
    } else {
      if (TN == 3) hipLaunchKernelGGL((igemm_bf16_kernel<256, 3, false, false, 4, 8>), grid, block8, 0, s, p);
      else if (TN == 2) hipLaunchKernelGGL((igemm_bf16_kernel<256, 2, false, false, 4, 8>), grid, block8, 0, s, p);
      else hipLaunchKernelGGL((igemm_bf16_kernel<256, 1, false, false, 4, 8>), grid, block8, 0, s, p);
    }
    return lic_check_launch();
  }
#define LIC_IGEMMH_LAUNCH(bm, tn)                                                           \
  do {                                                                                      \
    constexpr int R4 = (tn <= 2) ? 4 : 3; /* the ring p.ring asks for on 128-row tiles */   \
    if (fuse && p.ring == 4)                                                                \
      hipLaunchKernelGGL((igemm_bf16_kernel<128, tn, false, true, R4>), grid, block, 0, s, p); \
    else if (fuse)                                                                          \
      hipLaunchKernelGGL((igemm_bf16_kernel<128, tn, false, true>), grid, block, 0, s, p);  \
    else if (p.prologue == 1)                                                               \
      hipLaunchKernelGGL((igemm_bf16_kernel<bm, tn, true>), grid, block, 0, s, p);          \
    else if (p.ring == 4)                                                                   \
      hipLaunchKernelGGL((igemm_bf16_kernel<128, tn, false, false, R4>), grid, block, 0, s, p); \
    else                                                                                    \
      hipLaunchKernelGGL((igemm_bf16_kernel<bm, tn>), grid, block, 0, s, p);                \
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
  if (p.ksplit > 1) {
    const long npix = (long)d->B * d->Ho * d->Wo;
    hipLaunchKernelGGL(igemm_bf16_finish_kernel, dim3(ew_grid(npix * d->Cout / 8, 256)), dim3(256), 0, s,
                       (const float*)p.slabs, p.ksplit, npix, d->Cout / 8, d->bias, d->out, (long)d->out_ld,
                       out_f32 ? 1 : 0, d->epilogue == LIC_EPI_LEAKY ? 1 : 0, d->slope);
  }
  return lic_check_launch();
}

// ------------------------------------------------------------------------------------------------
// wgrad (bf16 operands, fp32 slabs): R[tap][m][n] = sum_pix A[pix][m] * B[pix][n].
// Both operands are activations stored pixel-major, but the MFMA wants 8 consecutive K (= pixels)
// per lane.  Each chunk of BK pixels is DMA'd global -> LDS as it lies in memory
// ([64-channel sub-tile][BK px][64 ch], 128-byte rows: thread t owns pixel t/8 (+32 per pass),
// 16-byte slot t%8 = byte 16t, the wave-linear DMA order) and the transpose happens in the read:
// `ds_read_b64_tr_b16` hands lane (channel i of a 16-channel group) 4 consecutive pixels of its
// channel, two of them make one MFMA operand.  The 4 pixel rows of a transposed read are 128 B
// apart (rows q and q+2 on the same banks), so the two 64-byte halves of a row are swapped on rows
// with bit 1 set -- on the DMA's per-lane SOURCE address and in the read address.
// Workgroup = 2x2 waves, wave tile = (32*TM) x (32*TN), BK = 32 pixels.  A chunk is only
// TM*TN*2 MFMAs of 32 cycles per wave -- far shorter than an L2/HBM round trip -- so the DMA runs
// TWO chunks ahead through a ring of three LDS buffers: `s_waitcnt vmcnt(N)` leaves the youngest
// chunk's N DMAs in flight, a raw `s_barrier` (no fence: `__syncthreads()` would drain vmcnt to 0)
// publishes the oldest one, and the buffer freed by the previous chunk is refilled at once.
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
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int WH_BK = 32;

// SQB: the column operand is squared at the read (GDN d-gamma: t^T . x^2)
template <int TM, int TN, bool SQB = false>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradHParams p) {
  constexpr int BK = WH_BK;
  constexpr int NPASS = BK / 32;               // DMA pieces per thread per sub-tile
  constexpr int SUB = BK * 64;                 // bf16 elements of one [BK px][64 ch] sub-tile
  constexpr int NS = TM + TN;
  constexpr int NL = NS * NPASS;               // DMA instructions per thread per chunk
  constexpr int BMt = 64 * TM, BNt = 64 * TN;
  constexpr int WM = BMt / 2, WN = BNt / 2;
  __shared__ __attribute__((aligned(16))) bf16_t smem[3][NS][SUB];
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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.0f;

  // DMA slots: thread t owns pixel row t/8 (+32 per pass) and physical 16-byte slot t%8 of every
  // sub-tile; it fetches logical slot (t%8) ^ 4*bit1(row)
  const int kr = tid >> 3;
  const int c8 = (((tid & 7) ^ (((tid >> 4) & 1) << 2))) * 8;
  const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_lic_zero16h);
  auto issue = [&](int c, int buf) {
    const int cc = c < c_end ? c : c_end - 1;  // past the end: harmless duplicate DMA into the idle buffer
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const long pk = (long)cc * BK + kr + 32 * j;
      const bool inb = pk < p.Ps;
      const int pix = inb ? (int)pk : 0;
      const int b = fdivb(pix, p.dHW);
      const int rem = pix - b * p.Hs * p.Ws;
      const int hs = fdivb(rem, p.dW), ws = rem - hs * p.Ws;
      const int hl = hs * p.stride - p.pad + r, wl = ws * p.stride - p.pad + s;
      const bool gok = inb && hl >= 0 && wl >= 0 && hl < p.Hl && wl < p.Wl;
      const long gpix = ((long)b * p.Hl + hl) * p.Wl + wl;
      const bool rok = p.row.gathered ? gok : inb, cok = p.col.gathered ? gok : inb;
      const long rpx = (p.row.gathered ? gpix : (long)pix), cpx = (p.col.gathered ? gpix : (long)pix);
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int ch = m0 + 64 * t + c8;
        const bool ok = rok && ch < p.row.C;
        const bf16_t* src = ok ? p.row.ptr + rpx * p.row.ld + ch : zsrc;
        __builtin_amdgcn_global_load_lds((lich_gptr_t)src, (lich_lptr_t)&smem[buf][t][j * 2048 + wave * 512], 16, 0,
                                         0);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int ch = n0 + 64 * t + c8;
        const bool ok = cok && ch < p.col.C;
        const bf16_t* src = ok ? p.col.ptr + cpx * p.col.ld + ch : zsrc;
        __builtin_amdgcn_global_load_lds((lich_gptr_t)src, (lich_lptr_t)&smem[buf][TM + t][j * 2048 + wave * 512],
                                         16, 0, 0);
      }
    }
  };
  // transposed-read lane address inside a sub-tile, for channel half (0 / 32): lane 4q+pp of the
  // 16-lane group g supplies row (8*lh + q), channels 32*half + 16*(g&1) + 4*pp .. +3.
  // The reads are inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of the tr16 builtin whenever
  // an LDS-DMA is in flight (it cannot see that the ring buffers are disjoint), which would undo
  // the two-chunk prefetch; with asm the lgkmcnt wait is ours to place (one per K step).
  const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
  unsigned tr_addr[2];  // LDS byte address of this lane's element in sub-tile 0 of buffer 0
  {
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&smem[0][0][0];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int cs = 32 * half + 16 * tg + 4 * tp;
      const int slot = (cs >> 3) ^ ((tq >> 1) << 2);
      tr_addr[half] = base + 2u * (unsigned)((8 * lh + tq) * 64 + slot * 8 + (cs & 7));
    }
  }
  auto compute = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      // pixels 16*ks + 8*lh + {0..3} (lo) and {4..7} (hi) of this lane's channel, per 32-channel tile
      unsigned long long lo[NS], hi[NS];
#pragma unroll
      for (int t = 0; t < TM + TN; ++t) {
        const int ch = t < TM ? wm0 + t * 32 : wn0 + (t - TM) * 32;  // wave-uniform
        const int sub = t < TM ? (ch >> 6) : TM + (ch >> 6);
        const unsigned addr = tr_addr[(ch >> 5) & 1] + 2u * (unsigned)((buf * NS + sub) * SUB + 16 * ks * 64);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[t]) : "v"(addr));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi[t]) : "v"(addr));
      }
      // one wait for the K step; the operands pass through it so the MFMAs cannot be scheduled above
      if constexpr (NS == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]));
      else if constexpr (NS == 3)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]));
      else if constexpr (NS == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]),
                       "+v"(hi[3]));
      else if constexpr (NS == 5)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]),
                       "+v"(hi[3]), "+v"(lo[4]), "+v"(hi[4]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]),
                       "+v"(hi[3]), "+v"(lo[4]), "+v"(hi[4]), "+v"(lo[5]), "+v"(hi[5]));
      bf16x8 fr[NS];
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        struct {
          unsigned long long a, b;
        } pr{lo[t], hi[t]};
        fr[t] = __builtin_bit_cast(bf16x8, pr);
        if constexpr (SQB)
          if (t >= TM) fr[t] = sq8(fr[t]);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[a], fr[TM + b], acc[a][b], 0, 0, 0);
    }
  };

  if (nloc > 0) {
    // Ring of three buffers.  At the top of iteration c the DMAs of chunks c and c+1 are in flight
    // (c+1 younger).  vmcnt(NL) retires mine of chunk c; the barrier tells me everyone's landed
    // and that everyone finished reading chunk c-1, whose buffer takes chunk c+2.
    // (Buffer indices are compile-time constants -- the ring is unrolled by three -- because hipcc
    // waits vmcnt(0) in front of any LDS read it cannot prove disjoint from an LDS-DMA in flight.)
    issue(c_begin, 0);
    issue(c_begin + 1, 1);
    auto step = [&](int c, auto cur, auto fill) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(NL) : "memory");
      issue(c_begin + c + 2, decltype(fill)::value);
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    int c = 0;
    for (; c + 2 < nloc; c += 3) {
      step(c, I0{}, I2{});
      step(c + 1, I1{}, I0{});
      step(c + 2, I2{}, I1{});
    }
    if (c < nloc) step(c, I0{}, I2{});
    if (c + 1 < nloc) step(c + 1, I1{}, I0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // duplicate tail DMAs must land before the wave ends
  }
  float* slab = p.slabs + ((long)split * p.ntaps + tap) * p.row.C * p.col.C;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + wn0 + b * 32 + li;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = m0 + wm0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        if (m < p.row.C && n < p.col.C) slab[(long)m * p.col.C + n] = acc[a][b][q];
      }
      __builtin_amdgcn_sched_barrier(0);  // one accumulator tile at a time: bounds the live VGPRs
    }
}

__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* slabs, float* dst, int splitk,
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

long lic_pick_splits(long base, long slots, long max_sk);  // lic_gemm.hip

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
  // widest tile (in 64-channel units, at most 3) that divides the channel count's 64-padding
  auto pick = [](int C) {
    const int u = (C + 63) / 64;
    return u % 3 == 0 ? 3 : (u % 2 == 0 ? 2 : (u == 1 ? 1 : (u > 4 ? 3 : 2)));
  };
  pl->TM = pick(pl->Cm);
  pl->TN = pick(pl->Cn);
  pl->MTt = (pl->Cm + 64 * pl->TM - 1) / (64 * pl->TM);
  pl->NTt = (pl->Cn + 64 * pl->TN - 1) / (64 * pl->TN);
  const long Ps = (long)d->B * d->Hs * d->Ws;
  pl->nchunks = (int)((Ps + WH_BK - 1) / WH_BK);
  const long base = (long)pl->MTt * pl->NTt * pl->ntaps;
  const long max_sk = (pl->nchunks + 15) / 16;  // at least 16 chunks (512 pixels) per split
  // resident workgroups per CU: 2 for the 5-6 sub-tile variants (registers / 72 KiB LDS), else 3
  long per_cu = pl->TM + pl->TN >= 5 ? 2 : 3;
  if (const char* e = getenv("LIC_WGRAD_BF16_ROUND")) {   // tuning aid: workgroups per CU one round of splits should fill
    const long v = atol(e);
    if (v >= 1 && v <= 3) per_cu = v < per_cu ? v : per_cu;
  }
  const long sk = lic_pick_splits(base, 256L * per_cu, max_sk);
  pl->cps = (int)((pl->nchunks + sk - 1) / sk);
  pl->splitk = (pl->nchunks + pl->cps - 1) / pl->cps;
  return LIC_OK;
}
LIC_EXPORT size_t lic_wgrad_bf16_workspace_bytes(const lic_wgrad_desc* d) {
  WgHPlan pl;
  if (wgh_plan(d, &pl) != LIC_OK) return 0;
  return (size_t)pl.splitk * pl.ntaps * pl.Cm * pl.Cn * sizeof(float);
}
LIC_EXPORT int lic_wgrad_bf16_kernel_name(const lic_wgrad_desc* d, char* buf, size_t n) {
  WgHPlan pl;
  const int rc = wgh_plan(d, &pl);
  if (rc != LIC_OK) return rc;
  if (!buf || n == 0) return LIC_ERR_INVALID;
  snprintf(buf, n, "wgrad_bf16_kernel<%d, %d, %s>", pl.TM, pl.TN, (d->g_is_row ? d->sq_p : d->sq_g) ? "true" : "false");
  return LIC_OK;
}
// p / g are bf16 activations; dst and the workspace are fp32
static int wgrad_bf16_impl(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, lic_reduce_job* job,
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
  if (p.row.sq) return LIC_ERR_UNSUPPORTED;  // only the column operand can be squared (GDN d-gamma)
  if (p.col.sq) {
    if (pl.TM != pl.TN) return LIC_ERR_UNSUPPORTED;  // C x C products only
    if (pl.TM == 1) hipLaunchKernelGGL((wgrad_bf16_kernel<1, 1, true>), grid, block, 0, s, p);
    if (pl.TM == 2) hipLaunchKernelGGL((wgrad_bf16_kernel<2, 2, true>), grid, block, 0, s, p);
    if (pl.TM == 3) hipLaunchKernelGGL((wgrad_bf16_kernel<3, 3, true>), grid, block, 0, s, p);
  } else {
#define LIC_WGH(tm, tn) \
  if (pl.TM == tm && pl.TN == tn) hipLaunchKernelGGL((wgrad_bf16_kernel<tm, tn>), grid, block, 0, s, p)
  LIC_WGH(1, 1);
  LIC_WGH(1, 2);
  LIC_WGH(1, 3);
  LIC_WGH(2, 1);
  LIC_WGH(2, 2);
  LIC_WGH(2, 3);
  LIC_WGH(3, 1);
  LIC_WGH(3, 2);
  LIC_WGH(3, 3);
#undef LIC_WGH
  }
  rc = lic_check_launch();
  if (rc != LIC_OK) return rc;
  const long total = (long)pl.ntaps * pl.Cm * pl.Cn;
  if (job) {  // the slab reduction is left to a later lic_reduce_batch
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
  hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s, (const float*)workspace,
                     d->dst, pl.splitk, pl.ntaps, pl.Cm, pl.Cn, (long)d->dst_sm, (long)d->dst_sn, (long)d->dst_stap,
                     d->scale);
  return lic_check_launch();
}
LIC_EXPORT int lic_wgrad_bf16(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes,
                              lic_stream_t stream) {
  return wgrad_bf16_impl(d, workspace, workspace_bytes, nullptr, stream);
}
// the MFMA launch only; `job` receives the slab reduction for a later lic_reduce_batch (the workspace must live until then)
LIC_EXPORT int lic_wgrad_bf16_partial(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes,
                                      lic_reduce_job* job, lic_stream_t stream) {
  if (!job) return LIC_ERR_INVALID;
  return wgrad_bf16_impl(d, workspace, workspace_bytes, job, stream);
}

// ---- lic_reduce_batch: every pending reduction of a backward pass in ONE launch -------------------------------------
// A step of the bf16 configurations ends 24 weight-gradient launches with a slab reduction and 17 column-sum launches
// with a second stage: 41 launches of 5-15 us each (15 % of the step's kernel time) whose arithmetic is tiny.  The
// producers can leave them pending (lic_wgrad_bf16_partial, lic_colsum*_bf16_partial fill a lic_reduce_job); this
// kernel runs a table of them -- same per-element arithmetic and association order as the stand-alone kernels, so the
// results are bitwise theirs -- with the two things that sit behind a reduction folded in: the GDN
// re-parametrisation's backward (epilogue 1: lic_gdn_reparam_bwd on the reduced value) and the (tap, channel) <-> column
// index maps of the RGB layers' weight gradients (mdiv / ndiv: the split index of lic_prep_job).
struct ReduceTable {
  lic_reduce_job j[LIC_REDUCE_MAX_JOBS];
  int n;
};
__global__ __launch_bounds__(256) void reduce_batch_kernel(const ReduceTable tb) {
  int ji = 0;
  while (ji + 1 < tb.n && (int)blockIdx.x >= tb.j[ji + 1].block0) ++ji;
  const lic_reduce_job& q = tb.j[ji];
  const int blk = (int)blockIdx.x - q.block0;
  if (q.kind == LIC_REDUCE_SLABS) {
    const long total = (long)q.ntaps * q.Cm * q.Cn;
    for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)q.nblocks * 256) {
      // the stand-alone kernels' association order -- slab z into partial sum z % 8 in slab order, the slabs behind the last
      // full group of eight into sum 0, then the tree -- with the LOADS arranged: 32 slabs requested before the first add (a
      // 512-slab job's threads went through 64 dependent round trips of 8 loads: 42 us for 21 MB, and the launch waits for
      // its slowest block), the tail's up to 7 slabs as one round trip
      float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* sp = q.src + i;
      int z = 0;
      for (; z + 32 <= q.splitk; z += 32) {
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = sp[(long)(z + k) * total];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int k = 0; k < 8; ++k) a8[k] += v[8 * r + k];
      }
      for (; z + 8 <= q.splitk; z += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sp[(long)(z + k) * total];
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] += v[k];
      }
      if (z < q.splitk) {
        float v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = sp[(long)(z + k < q.splitk ? z + k : q.splitk - 1) * total];
#pragma unroll
        for (int k = 0; k < 7; ++k)
          if (z + k < q.splitk) a8[0] += v[k];
      }
      const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
      const int n = (int)(i % q.Cn);
      const long t2 = i / q.Cn;
      const int m = (int)(t2 % q.Cm);
      const int tap = (int)(t2 / q.Cm);
      if (m >= q.Mvalid || n >= q.Nvalid) continue;
      const long mo = q.mdiv ? (long)(m / q.mdiv) * q.sm + (long)(m % q.mdiv) * q.smr : (long)m * q.sm;
      const long no = q.ndiv ? (long)(n / q.ndiv) * q.sn + (long)(n % q.ndiv) * q.snr : (long)n * q.sn;
      const long o = mo + no + tap * q.stap;
      float v = acc * q.scale;
      if (q.epilogue == LIC_REDUCE_EPI_REPARAM) {
        const float pv = q.param[o], w = pv > q.bound ? pv : q.bound;
        const float g = v * 2.0f * w;
        v = (pv >= q.bound || g < 0.0f) ? g : 0.0f;
      }
      q.dst[o] = v;
    }
  } else {  // column partials [splitk = nchunk][Cn = C] -> dst[c]: colsum_bf16_stage2's arithmetic, 16 columns per block
    __shared__ double red[16][17];
    const int cx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int c = blk * 16 + cx;
    double acc = 0.0;
    if (c < q.Cn) {
      // (same sums in the same order; eight rows' loads requested before the first add: the per-workgroup rows of the GDN
      // backward make 2048-row jobs, 128 dependent round trips per thread as a plain loop -- 58 us for 1 MB)
      int y = ly;
      for (; y + 16 * 7 < q.splitk; y += 16 * 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = q.src[(long)(y + 16 * k) * q.Cn + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (double)v[k];
      }
      for (; y < q.splitk; y += 16) acc += (double)q.src[(long)y * q.Cn + c];
    }
    red[ly][cx] = acc;
    __syncthreads();
    if (ly == 0 && c < q.Cn) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][cx];
      float v = (float)(t * (double)q.scale);
      if (q.epilogue == LIC_REDUCE_EPI_REPARAM) {
        const float pv = q.param[c], w = pv > q.bound ? pv : q.bound;
        const float g = v * 2.0f * w;
        v = (pv >= q.bound || g < 0.0f) ? g : 0.0f;
      }
      q.dst[c] = v;
    }
  }
}
LIC_EXPORT int lic_reduce_batch(const lic_reduce_job* jobs, int32_t njobs, lic_stream_t stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return LIC_ERR_INVALID;
  for (int base = 0; base < njobs; base += LIC_REDUCE_MAX_JOBS) {
    ReduceTable tb;
    tb.n = njobs - base < LIC_REDUCE_MAX_JOBS ? njobs - base : LIC_REDUCE_MAX_JOBS;
    int blocks = 0;
    // the launch ends with its slowest block: jobs whose threads walk the longest dependent chains (many slabs or rows, few
    // blocks) get the lowest block numbers, so they start first and run beside the bandwidth-sized jobs instead of after them
    int order[LIC_REDUCE_MAX_JOBS];
    for (int i = 0; i < tb.n; ++i) order[i] = i;
    auto depth = [&](int i) {
      const lic_reduce_job& q = jobs[base + i];
      return q.kind == LIC_REDUCE_COLUMNS ? (long)q.splitk * 2 : (long)q.splitk;   // (a column thread adds doubles, one row in 16)
    };
    std::stable_sort(order, order + tb.n, [&](int a, int b) { return depth(a) > depth(b); });
    for (int oi = 0; oi < tb.n; ++oi) {
      const int i = oi;
      lic_reduce_job q = jobs[base + order[oi]];
      if (!q.src || !q.dst || q.splitk <= 0 || q.Cn <= 0 || (q.epilogue == LIC_REDUCE_EPI_REPARAM && !q.param)) return LIC_ERR_INVALID;
      if (q.kind == LIC_REDUCE_SLABS) {
        if (q.ntaps <= 0 || q.Cm <= 0 || q.mdiv < 0 || q.ndiv < 0) return LIC_ERR_INVALID;
        if (q.Mvalid <= 0 || q.Mvalid > q.Cm) q.Mvalid = q.Cm;   // (0 = every row / column)
        if (q.Nvalid <= 0 || q.Nvalid > q.Cn) q.Nvalid = q.Cn;
        q.nblocks = ew_grid((long)q.ntaps * q.Cm * q.Cn, 256);
      } else if (q.kind == LIC_REDUCE_COLUMNS) {
        q.nblocks = (q.Cn + 15) / 16;
      } else {
        return LIC_ERR_INVALID;
      }
      q.block0 = blocks;
      blocks += q.nblocks;
      tb.j[i] = q;
    }
    hipLaunchKernelGGL(reduce_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tb);
    const int rc = lic_check_launch();
    if (rc != LIC_OK) return rc;
  }
  return LIC_OK;
}
