// Halo-resident 5x5 stride-2 TRANSPOSED convolution in bf16 storage (included by lic_gemm_bf16.hip after
// lic_halo_bf16.h): ConvTranspose2d(k=5, s=2, p=2, output_padding=1) forward and the data gradient of the strided
// convolution (Components.py:12-14, 41-43).  Same machinery as halo_conv_bf16_kernel -- input patch resident in LDS,
// weight fragments straight from L2 into a register ring, no barrier inside the tap loop, persistent workgroups --
// with the geometry of the transposed layer:
//
//   * the four output phases (oy & 1, ox & 1) of a stride-2 transposed layer are four stride-1 convolutions of the
//     SAME input with 9 / 6 / 6 / 4 of the 25 taps.  A workgroup owns an 8 x 32 tile of phase pixels q (= 16 x 64
//     output pixels) and runs the four phases one after the other on one input halo of (8+2) x (32+2) pixels -- 21 KB
//     per 32-channel chunk, a single plane (stride 1: no even / odd split), two chunk buffers.  Per phase the loop is
//     chunk outside, the phase's taps inside, exactly as in the strided kernel; the last chunk of a phase prefetches
//     the first chunk of the next one (the same pixels again: 4x the halo DMA of a tile, still 2.5x fewer L2 -> LDS
//     bytes than a gather per (tap, chunk), and the weight panels dominate the traffic anyway).
//   * the taps of a phase are not a multiple of anything convenient (9, 6, 6, 4), so the weight ring is sized per
//     phase: D = 6, 6, 6, 4 register sets with a tap's set always requested THREE steps ahead; the loop body is two
//     chunks (18 / 12 / 12 / 8 steps, a multiple of D), every body starts at slot 0, and the three sets in flight at
//     a phase (or tile) boundary are the next body's slots 0..2 whatever the two ring sizes are.
//   * halo pieces are issued in the first taps of a chunk only (never in its last three), so every `vmcnt` count is
//     a function of the tap index alone, across chunk, phase and tile boundaries; the barrier tap of a chunk waits
//     for the pieces as well as for its own weight set.
//   * the A fragments of the next phase's first tap are requested after the phase's loop (the in-loop prefetch of the
//     last step assumed the same phase again): their latency hides under the epilogue.
//
// K order per output: chunk-major like the strided halo kernel (fp32 summation order differs from the tap-major
// implicit GEMM, results agree to rounding).
#pragma once

namespace halot {
constexpr int TH = 8, TWD = 32;                 // tile of phase pixels
constexpr int HR = TH + 2, WP = TWD + 2;        // input halo
constexpr int PL = HR * WP;                     // 340 pixels
constexpr int ROWB = WP * 64;                   // bytes of a halo row (32-channel chunk)
constexpr int BUFB = 24 * 1024;                 // one chunk buffer: 24 DMA pieces of 1 KiB (340 px x 64 B + pad)
constexpr int NPIECE = 6;                       // pieces per thread per chunk
constexpr int DIST = 3;                         // a tap's weight set is requested this many steps ahead
constexpr int XCH = 64 * 1024;                  // x^2 exchange area of the fused pool (behind the two chunk buffers)
constexpr int nt(int ph) { return ph == 0 ? 9 : (ph == 3 ? 4 : 6); }
constexpr int dd(int ph) { return ph == 3 ? 4 : 6; }
constexpr int py(int ph) { return ph >> 1; }
constexpr int px(int ph) { return ph & 1; }
constexpr int ns(int ph) { return px(ph) ? 2 : 3; }                       // taps per filter row of the phase
constexpr int tap_r(int ph, int k) { return 2 * (k / ns(ph)) + py(ph); }
constexpr int tap_s(int ph, int k) { return 2 * (k % ns(ph)) + px(ph); }
constexpr int tap_id(int ph, int k) { return tap_r(ph, k) * 5 + tap_s(ph, k); }
// input pixel of output (2 qy + py, 2 qx + px) under tap (r, s): (qy + (py + 2 - r) / 2, qx + (px + 2 - s) / 2);
// relative to the halo origin (qy0 - 1, qx0 - 1):
constexpr int dyoff(int ph, int k) { return (py(ph) + 2 - tap_r(ph, k)) / 2 + 1; }
constexpr int shoff(int ph, int k) { return (px(ph) + 2 - tap_s(ph, k)) / 2 + 1; }
// halo pieces issued at tap t of a chunk (none in the last three taps of any phase)
constexpr int np(int ph, int t) { return ph == 0 ? (t < 3 ? 2 : 0) : (ph == 3 ? (t == 0 ? 6 : 0) : (t < 2 ? 3 : 0)); }
constexpr int piece0(int ph, int t) { return ph == 0 ? 2 * t : (ph == 3 ? 0 : 3 * t); }
constexpr int last_piece_tap(int ph) { return ph == 0 ? 2 : (ph == 3 ? 0 : 1); }
// VMEM operations issued after the weight set of tap t (requested at tap t - 3, before that tap's pieces); taps
// before the chunk's first belong to the previous chunk's tail: weight loads only
constexpr int younger(int ph, int t, int NB) {
  int n = t - DIST >= 0 ? np(ph, t - DIST) : 0;
  for (int k = t - DIST + 1; k <= t - 1; ++k) n += NB + (k >= 0 ? np(ph, k) : 0);
  return n;
}
// the barrier tap (last of a chunk) must also have this chunk's halo pieces landed
constexpr int barrier_wait(int ph, int NB) {
  const int a = younger(ph, nt(ph) - 1, NB), b = (nt(ph) - 2 - last_piece_tap(ph)) * NB;
  return a < b ? a : b;
}
}  // namespace halot

template <int TW, bool FUSE = false>
__global__ __launch_bounds__(256, 1) void halo_convt_bf16_kernel(const IgemmHParams p) {
  using namespace halot;
  constexpr int NB = 2 * TW;
  constexpr int PANEL = 64 * TW * HB_BK * 2;
  constexpr int NM = 8 * TW;
  static_assert(NM >= 16, "13 slots of other work + 3 piece slots per tap");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUFB + (FUSE ? XCH : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int G = gridDim.x, ntiles = p.MT;
  int tile = blockIdx.x;
  {
    const int q = G >> 3, r = G & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }

  // ---- halo DMA (buffer_load ... lds on a per-tile descriptor, out-of-image lanes zero-filled by the range check)
  unsigned rel[NPIECE];
  unsigned meta[NPIECE / 2];
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) {
    const int n = (j * 4 + wave) * 64 + lane;
    const int P = n >> 2, slot = n & 3;
    const int hr = P / WP, hc = P - hr * WP;
    const int o = slot ^ ((hc >> 2) & 3);
    rel[j] = 2u * (unsigned)((hr * p.Wi + hc) * (int)p.in_ld + o * 8);
    const unsigned m = (unsigned)(hr & 31) | ((unsigned)hc << 5) | (hr < HR ? 0u : 0x8000u);
    if (j & 1) meta[j >> 1] |= m << 16;
    else meta[j >> 1] = m;
  }
  struct Geo {
    int b, qy0, qx0;
    long base;
    unsigned vmask;
  };
  auto geom = [&](int t) {
    Geo g;
    const int tpi = p.htx * p.hty;
    g.b = t / tpi;
    const int trem = t - g.b * tpi;
    const int ty = trem / p.htx, tx = trem - ty * p.htx;
    g.qy0 = ty * TH;
    g.qx0 = tx * TWD;
    const int iy0 = g.qy0 - 1, ix0 = g.qx0 - 1;
    g.base = ((long)(g.b * p.Hi + iy0) * p.Wi + ix0) * p.in_ld;
    unsigned vm = 0;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      unsigned mm = meta[j >> 1];
      asm volatile("" : "+v"(mm));
      const unsigned m = (mm >> (16 * (j & 1))) & 0xFFFFu;
      const unsigned iy = (unsigned)(iy0 + (int)(m & 31u)), ix = (unsigned)(ix0 + (int)((m >> 5) & 127u));
      const unsigned ok = (unsigned)(iy < (unsigned)p.Hi) & (unsigned)(ix < (unsigned)p.Wi) & ((m >> 15) ^ 1u);
      vm |= ok << j;
    }
    g.vmask = vm;
    return g;
  };
  const int nch = p.cpt;
  constexpr unsigned OOB = 0xFFFFFFF0u, NREC = 0x7FFFFFF0u;
  auto halo_rsrc = [&](long base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.in + base), 0, NREC, 0x00020000);
  };
  auto dma_piece = [&](int j, __amdgpu_buffer_rsrc_t rs, int soff, unsigned vm, int buf) {
    const unsigned off = ((vm >> j) & 1u) ? rel[j] : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lich_lptr_t)(smem + buf * BUFB + (j * 4 + wave) * 1024), 16, (int)off, soff,
                                             0, 0);
  };

  // ---- A fragment addresses: lane (li, lh) of row tile a reads halo pixel (4 wm + a + dyoff, li + shoff), K octet
  // (2 ks + lh) ^ ((column >> 2) & 3).  abase[buf][shoff][ks]; (a + dyoff) * ROWB is the immediate offset.
  unsigned abase[2][3][2];
  {
    const unsigned s0 = (unsigned)(size_t)(lich_lptr_t)smem;
#pragma unroll
    for (int sh = 0; sh < 3; ++sh) {
      const int hc = li + sh;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned a0 = s0 + 4 * wm * ROWB + hc * 64 + 16 * ((2 * ks + lh) ^ ((hc >> 2) & 3));
        abase[0][sh][ks] = a0;
        abase[1][sh][ks] = a0 + BUFB;
      }
    }
  }

  // ---- weight fragments; bptr walks the panels in consumption order: per phase, per chunk, the phase's taps
  const char* bptr = reinterpret_cast<const char*>(p.w) + (long)wn * TW * 2048 + lane * 16 + 2048;
  const long tap_inc = (long)nch * PANEL;

  f32x16 acc[4][TW];
  hu32x4 af[2][4][2];
  hu32x4 bq[6][TW][2];
  auto& af_ = af;
  auto& bq_ = bq;

  auto a_read1 = [&](auto slotc, auto bufc, auto phc, auto tapc, auto ic) {
    constexpr int SLOT = decltype(slotc)::value, BUF = decltype(bufc)::value, PH = decltype(phc)::value;
    constexpr int K = decltype(tapc)::value, I = decltype(ic)::value, a = I >> 1, ks = I & 1;
    auto& dst = af[SLOT][a][ks];
    const unsigned addr = abase[BUF][shoff(PH, K)][ks];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "i"((a + dyoff(PH, K)) * ROWB));
  };
  auto b_load1 = [&](auto slotc, auto ic) {
    constexpr int SLOT = decltype(slotc)::value, I = decltype(ic)::value, t = I >> 1, ks = I & 1;
    auto& dst = bq[SLOT][t][ks];
    const char* src = bptr;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(src), "i"(I * 1024 - 2048));
  };
  auto wait_frags = [&](auto slota, auto slotb, auto nc) {
    constexpr int SA = decltype(slota)::value, SB = decltype(slotb)::value, N = decltype(nc)::value;
    auto& af = af_;
    auto& bq = bq_;
    static_assert(TW == 2, "128 output channels");
    asm volatile("s_waitcnt vmcnt(%12) lgkmcnt(0)"
                 : "+a"(af[SA][0][0]), "+a"(af[SA][0][1]), "+a"(af[SA][1][0]), "+a"(af[SA][1][1]), "+a"(af[SA][2][0]),
                   "+a"(af[SA][2][1]), "+a"(af[SA][3][0]), "+a"(af[SA][3][1]), "+v"(bq[SB][0][0]), "+v"(bq[SB][0][1]),
                   "+v"(bq[SB][1][0]), "+v"(bq[SB][1][1])
                 : "i"(N));
  };
  auto keep_set = [&](auto slotb) {
    constexpr int SB = decltype(slotb)::value;
    auto& bq = bq_;
    asm volatile("" : "+v"(bq[SB][0][0]), "+v"(bq[SB][0][1]), "+v"(bq[SB][1][0]), "+v"(bq[SB][1][1]));
  };

  // One step of phase PH: U = step inside the two-chunk body (0 .. 2 NT - 1), c = the chunk it belongs to.
  auto tap_step = [&](auto phc, auto uc, int c, __amdgpu_buffer_rsrc_t drs, int dsoff, unsigned dvm) {
    constexpr int PH = decltype(phc)::value, U = decltype(uc)::value;
    constexpr int NT = nt(PH), DP = dd(PH), T = U % NT, BUF = U / NT;
    constexpr int SA = U & 1, SB = U % DP;
    constexpr int UN = (U + 1) % (2 * NT), TN1 = UN % NT, BUFN = UN / NT;
    constexpr int UL = U + DIST, TL = UL % NT;        // the step whose weight set is requested now
    static_assert((2 * NT) % DP == 0 && DP > DIST, "ring size");
    wait_frags(std::integral_constant<int, SA>{}, std::integral_constant<int, SB>{},
               std::integral_constant<int, (T == NT - 1) ? barrier_wait(PH, NB) : younger(PH, T, NB)>{});
    if constexpr (T == NT - 1) asm volatile("s_barrier" ::: "memory");
    lich_for_seq(std::make_integer_sequence<int, NM>{}, [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int ks = I / (4 * TW), rem = I % (4 * TW), t = rem / 4, a = rem % 4;
      acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[SB][t][ks]),
                                                           __builtin_bit_cast(bf16x8, af[SA][a][ks]), acc[a][t], 0, 0, 0);
      if constexpr (I < 8) {
        a_read1(std::integral_constant<int, 1 - SA>{}, std::integral_constant<int, BUFN>{}, phc,
                std::integral_constant<int, TN1>{}, ic);
      } else if constexpr (I < 8 + NB) {
        b_load1(std::integral_constant<int, UL % DP>{}, std::integral_constant<int, I - 8>{});
      } else if constexpr (I == 8 + NB) {
        // step to the panel consumed after the one just requested
        if constexpr (TL == NT - 1) {
          const int cl = c - BUF + UL / NT;   // the chunk the requested set belongs to
          constexpr int PN = (PH + 1) & 3;
          const long next_chunk = (long)(tap_id(PH, 0) - tap_id(PH, NT - 1)) * tap_inc + PANEL;
          const long next_phase = (long)(tap_id(PN, 0) - tap_id(PH, NT - 1)) * tap_inc - (long)(nch - 1) * PANEL;
          bptr += (cl + 1 < nch) ? next_chunk : next_phase;
        } else if constexpr (UL >= 2 * NT) {
          // a set of the NEXT body: this phase's next chunk pair, or -- from the last body -- the next phase's first
          // steps, whose taps are spaced differently
          constexpr int PN = (PH + 1) & 3, KN = UL - 2 * NT;
          const bool last_body = c - BUF + 2 >= nch;
          bptr += (long)(last_body ? tap_id(PN, KN + 1) - tap_id(PN, KN) : tap_id(PH, TL + 1) - tap_id(PH, TL)) * tap_inc;
        } else {
          bptr += (long)(tap_id(PH, TL + 1) - tap_id(PH, TL)) * tap_inc;
        }
      } else if constexpr (I >= 13 && np(PH, T) > 0) {
        // up to two pieces per slot in the three slots behind the pointer step
        constexpr int per = (np(PH, T) + 2) / 3;
#pragma unroll
        for (int k = 0; k < per; ++k)
          if ((I - 13) * per + k < np(PH, T)) dma_piece(piece0(PH, T) + (I - 13) * per + k, drs, dsoff, dvm, 1 - BUF);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto read_first = [&](auto phc) {   // tap 0 of phase `phc` from buffer 0 into A slot 0
    lich_for_seq(std::make_integer_sequence<int, 8>{}, [&](auto ic) {
      a_read1(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, phc, std::integral_constant<int, 0>{}, ic);
    });
  };

  // ---- prologue: chunk 0 into buffer 0, the weight sets of phase 0's first three steps, its first A fragments
  Geo cur = geom(tile);
  {
    const __amdgpu_buffer_rsrc_t rs0 = halo_rsrc(cur.base);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) dma_piece(j, rs0, 0, cur.vmask, 0);
  }
  lich_for_seq(std::make_integer_sequence<int, DIST>{}, [&](auto sc) {
    constexpr int S = decltype(sc)::value;
    lich_for_seq(std::make_integer_sequence<int, NB>{}, [&](auto ic) { b_load1(sc, ic); });
    bptr += (long)(tap_id(0, S + 1) - tap_id(0, S)) * tap_inc;
  });
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(DIST * NB) : "memory");
  read_first(std::integral_constant<int, 0>{});

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  for (;;) {
    const int tnext = tile + G;
    const bool more = tnext < ntiles;
    Geo nxt = cur;
    if (more) nxt = geom(tnext);
    const unsigned nvm = more ? nxt.vmask : 0u;
    lich_for_seq(std::make_integer_sequence<int, 4>{}, [&](auto phc) {
      constexpr int PH = decltype(phc)::value, NT = nt(PH);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.0f;
      for (int c0 = 0; c0 < nch; c0 += 2) {
        // chunk c0 fetches chunk c0 + 1; chunk c0 + 1 fetches chunk c0 + 2, or chunk 0 again for the next phase, or the
        // next tile's chunk 0 after the last phase
        const bool last = c0 + 2 >= nch;
        const bool to_next_tile = last && PH == 3;
        const __amdgpu_buffer_rsrc_t rsA = halo_rsrc(cur.base), rsB = halo_rsrc(to_next_tile ? nxt.base : cur.base);
        const int sA = (c0 + 1) * HB_BK * 2, sB = last ? 0 : (c0 + 2) * HB_BK * 2;
        const unsigned vB = to_next_tile ? nvm : cur.vmask;
        lich_for_seq(std::make_integer_sequence<int, 2 * NT>{}, [&](auto uc) {
          constexpr int U = decltype(uc)::value;
          if constexpr (U < NT) tap_step(phc, uc, c0, rsA, sA, cur.vmask);
          else tap_step(phc, uc, c0 + 1, rsB, sB, vB);
        });
      }
      // the last step requested tap 0 of THIS phase's geometry: wait for it (its registers must not be re-used while
      // the read is in flight), then request the next phase's; that latency hides under the epilogue
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+a"(af[0][0][0]), "+a"(af[0][0][1]), "+a"(af[0][1][0]), "+a"(af[0][1][1]), "+a"(af[0][2][0]),
                     "+a"(af[0][2][1]), "+a"(af[0][3][0]), "+a"(af[0][3][1]));
      read_first(std::integral_constant<int, (PH + 1) & 3>{});

      // ---- epilogue of the phase, straight from the registers (see halo_conv_bf16_kernel)
      int lho = lh, lio = li, wno = wn, wmo = wm;
      asm volatile("" : "+v"(lho), "+v"(lio), "+s"(wno), "+s"(wmo));
      f32x4 bs[TW][4];
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) bs[t][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (p.bias) {
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            bs[t][g] = *reinterpret_cast<const f32x4*>(p.bias + (wno * TW + t) * 32 + 4 * lho + 8 * g);
      }
      auto out_pixel = [&](int a, bool& rok) {
        const int oy = 2 * (cur.qy0 + 4 * wmo + a) + py(PH), ox = 2 * (cur.qx0 + lio) + px(PH);
        rok = oy < p.Ho && ox < p.Wo;
        return rok ? ((long)cur.b * p.Ho + oy) * p.Wo + ox : 0L;
      };
      auto pack2 = [](f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); };
      auto store_tile = [&](bf16_t* base, long ld, long opix, bool rok, int cb, const unsigned (&pk)[8]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const u32x2 r0 = __builtin_amdgcn_permlane32_swap(pk[4 * s2], pk[4 * s2 + 2], false, false);
          const u32x2 r1 = __builtin_amdgcn_permlane32_swap(pk[4 * s2 + 1], pk[4 * s2 + 3], false, false);
          if (rok) {
            const hu32x4 o = {r0[0], r1[0], r0[1], r1[1]};
            *reinterpret_cast<hu32x4*>(base + opix * ld + cb + 16 * s2 + 8 * lho) = o;
          }
        }
      };
      if constexpr (FUSE) {
        // conv -> IGDN / GDN in the same launch: see the fused epilogue of halo_conv_bf16_kernel (x^2 fragments of the
        // two channel halves exchanged through LDS, here a dedicated area behind the chunk buffers)
        const bool inv = p.epilogue == LIC_EPI_CONV_IGDN;
        unsigned char* xch = smem + 2 * BUFB;
        // (gamma_eff^T fragments and beta_eff requested up front, all at once: see halo_conv_bf16_kernel)
        const bf16_t* gA = p.aux + lane * 8;
        const int ntile = p.Npad >> 5;
        bf16x8 gfr[2 * TW][2][TW];
#pragma unroll
        for (int tt = 0; tt < 2 * TW; ++tt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int t = 0; t < TW; ++t)
              gfr[tt][s2][t] = *reinterpret_cast<const bf16x8*>(gA + ((long)tt * ntile + (wno * TW + t)) * 1024 + s2 * 512);
        f32x4 be[TW][4];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) be[t][g] = *reinterpret_cast<const f32x4*>(p.beta + (wno * TW + t) * 32 + 4 * lho + 8 * g);
        auto frag_at = [&](int wmi, int a, int tt, int s2) { return xch + ((((wmi * 4 + a) * (2 * TW) + tt) * 2 + s2) * 64 + lane) * 16; };
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          bool rok;
          const long opix = out_pixel(a, rok);
#pragma unroll
          for (int t = 0; t < TW; ++t) {
            unsigned xpk[8], sqpk[8];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const f32x2 v = {acc[a][t][4 * g + 2 * h] + bs[t][g][2 * h], acc[a][t][4 * g + 2 * h + 1] + bs[t][g][2 * h + 1]};
                const unsigned pk = pack2(v);
                xpk[2 * g + h] = pk;
                const f32x2 xb = {__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
                acc[a][t][4 * g + 2 * h] = xb[0];
                acc[a][t][4 * g + 2 * h + 1] = xb[1];
                sqpk[2 * g + h] = pack2(xb * xb);
              }
            if (p.out3) store_tile(p.out3, p.out3_ld, opix, rok, (wno * TW + t) * 32, xpk);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
              *reinterpret_cast<hu32x4*>(frag_at(wmo, a, wno * TW + t, s2)) =
                  hu32x4{sqpk[4 * s2], sqpk[4 * s2 + 1], sqpk[4 * s2 + 2], sqpk[4 * s2 + 3]};
          }
        }
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x16 nacc[2][TW];
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) nacc[a2][t][r] = 0.0f;
#pragma unroll
          for (int tt = 0; tt < 2 * TW; ++tt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              bf16x8 b2[2];
#pragma unroll
              for (int a2 = 0; a2 < 2; ++a2) b2[a2] = *reinterpret_cast<const bf16x8*>(frag_at(wmo, 2 * hh + a2, tt, s2));
#pragma unroll
              for (int t = 0; t < TW; ++t) {
                const bf16x8 a2f = gfr[tt][s2][t];
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
                  nacc[a2][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2f, b2[a2], nacc[a2][t], 0, 0, 0);
              }
            }
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2) {
            const int a = 2 * hh + a2;
            bool rok;
            const long opix = out_pixel(a, rok);
#pragma unroll
            for (int t = 0; t < TW; ++t) {
              const int cb = (wno * TW + t) * 32;
              unsigned npk[8], ypk[8];
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                #pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const f32x2 nv = {nacc[a2][t][4 * g + 2 * h] + be[t][g][2 * h], nacc[a2][t][4 * g + 2 * h + 1] + be[t][g][2 * h + 1]};
                  npk[2 * g + h] = pack2(nv);
                  const f32x2 f = {inv ? __builtin_amdgcn_sqrtf(nv[0]) : __builtin_amdgcn_rsqf(nv[0]),
                                   inv ? __builtin_amdgcn_sqrtf(nv[1]) : __builtin_amdgcn_rsqf(nv[1])};
                  const f32x2 xv = {acc[a][t][4 * g + 2 * h], acc[a][t][4 * g + 2 * h + 1]};
                  ypk[2 * g + h] = pack2(xv * f);
                }
              }
              if (p.out2) store_tile(p.out2, p.out2_ld, opix, rok, cb, npk);
              store_tile(reinterpret_cast<bf16_t*>(p.out), p.out_ld, opix, rok, cb, ypk);
            }
          }
        }
        __syncthreads();   // (the next phase's pool writes the exchange area again)
      } else {
        const float sl = p.epilogue == LIC_EPI_LEAKY ? p.slope : 1.0f;
        const bool of32 = p.out_f32 != 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          bool rok;
          const long opix = out_pixel(a, rok);
#pragma unroll
          for (int t = 0; t < TW; ++t) {
            const int cb = (wno * TW + t) * 32;
            f32x4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              v[g] = f32x4{acc[a][t][4 * g], acc[a][t][4 * g + 1], acc[a][t][4 * g + 2], acc[a][t][4 * g + 3]} + bs[t][g];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[g][e] = __builtin_fmaxf(v[g][e], v[g][e] * sl);
            }
            if (of32) {
              if (rok) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                  *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + opix * p.out_ld + cb + 8 * g + 4 * lho) = v[g];
              }
            } else {
              unsigned pk[8];
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const f32x2 v0 = {v[g][0], v[g][1]}, v1 = {v[g][2], v[g][3]};
                pk[2 * g] = pack2(v0);
                pk[2 * g + 1] = pack2(v1);
              }
              store_tile(reinterpret_cast<bf16_t*>(p.out), p.out_ld, opix, rok, cb, pk);
            }
          }
        }
      }
    });
    if (!more) break;
    cur = nxt;
    tile = tnext;
  }
  // in flight: the (unused) weight sets of the three steps past the end, A slot 0, zero pieces into buffer 0
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+a"(af[0][0][0]), "+a"(af[0][0][1]), "+a"(af[0][1][0]), "+a"(af[0][1][1]), "+a"(af[0][2][0]),
                 "+a"(af[0][2][1]), "+a"(af[0][3][0]), "+a"(af[0][3][1]));
  lich_for_seq(std::make_integer_sequence<int, DIST>{}, keep_set);
}
