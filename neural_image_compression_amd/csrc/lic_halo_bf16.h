// Halo-resident 5x5 stride-2 convolution in bf16 storage (included by lic_gemm_bf16.hip).
//
// The implicit-GEMM kernel above gathers a [rows][32 channel] slice per (tap, channel chunk): every input pixel of a
// stride-2 5x5 layer comes through L2 -> LDS 25/4 times, and the launches move 1.2-1.6 GB that way for 107 GFLOP
// (DESIGN: the DMA traffic costs a quarter of the launch, the 64x64-per-wave LDS-read skeleton caps it at half the
// MFMA peak).  Here the LOOP ORDER is turned round -- channel chunk outside, the 25 taps inside -- so that one
// workgroup keeps the input patch of its output tile resident in LDS:
//
//   * output tile 8 rows x 32 columns (256 pixels) x all output channels; its input halo is 19 x 67 pixels.
//     One 32-channel chunk of it (81.5 KB) is DMA'd ONCE (`global_load_lds_dwordx4`, 64 bytes per pixel) and all 25
//     taps read their A fragments from it: 4.8x fewer L2 -> LDS bytes for the activations.  Two buffers of 80 KiB
//     (all of the CU's LDS): the next chunk lands while this one is multiplied.
//   * the halo is split into an EVEN-column and an ODD-column plane ([plane][19 rows][34 | 33 columns][64 B]): the
//     32 output pixels of an MFMA row tile then read 32 CONSECUTIVE plane pixels for every tap (column 2 ox + s ->
//     plane s & 1, column ox + (s >> 1)).  The four 16-byte K octets of a pixel are XOR-swizzled with (column >> 2) & 3
//     (on the DMA's per-lane source address and in the read address), which makes every ds_read_b128 lane group hit
//     16 distinct bank slots for every tap (checked exhaustively, tools/halo_bank_check.py); because the swizzle
//     depends on the column only, a tap moves a lane's address by a compile-time constant -- the immediate offset
//     of the read -- from one of five per-lane bases (plane x column shift): no address arithmetic in the loop.
//   * four waves, one per SIMD, 2 (pixel halves) x 2 (channel halves): each wave owns 128 pixels x 64 (96) channels =
//     4 x 2 (3) accumulator tiles, so a tap is 16 (24) MFMAs on 8 + 4 (6) fragments -- half the fragment bytes per
//     MFMA of the 64x64 wave tile.  The kernel runs at one wave per SIMD with the whole 512-entry register file.
//   * the weight fragments never touch LDS: each lane loads its 16 bytes of the packed panel (already in fragment
//     order) straight from L2 into registers, four taps ahead through a ring of five register sets (inline-asm
//     loads: hipcc would wait vmcnt(0) for them beside the LDS-DMA).  Nothing in the tap loop is shared between
//     waves any more, so there is NO barrier in it: one `s_barrier` per chunk (25 taps, 400 MFMAs per wave), where
//     the halo buffers change hands.
//   * `vmcnt` retires in order over weight loads and halo DMA pieces alike, so every count in the unrolled body is
//     a compile-time constant: a tap waits for its own weight set with everything younger still in flight.
//   * PERSISTENT workgroups (one per CU, all of its LDS): a workgroup walks tiles w, w + G, ...; the last chunk of a
//     tile prefetches the first chunk of the next one and the weight pointer wraps to the panels that tile starts with,
//     so only a workgroup's first tile pays a cold prologue (measured with in-kernel stamps on the one-tile-per-
//     workgroup version: prologue 8.7 us + LDS-staged epilogue 6.5 us around a 41 us main loop).
//   * the accumulators hold the TRANSPOSED tile (MFMA operands swapped: channels down the rows, pixels across the
//     lanes), as in the fused conv+GDN kernel above: a lane owns 4 x 4 consecutive channels of one pixel per tile, so
//     bias / LeakyReLU / bf16 packing are element-wise and one v_permlane32_swap per dword pair gives 16-byte stores --
//     the epilogue uses no LDS (the buffers already hold the next tile's first chunk).
//
// Same packed weights, same epilogue conventions as igemm_bf16_kernel.  The K order per output differs (chunk-major
// instead of tap-major): results agree with the other variants to fp32 summation order, not bitwise.
#pragma once

namespace halo {
constexpr int TH = 8, TWD = 32;                 // output tile (rows x columns)
constexpr int HR = 2 * TH + 3;                  // halo rows
constexpr int WP0 = TWD + 2, WP1 = TWD + 1;     // columns of the even / odd plane
constexpr int PL0 = HR * WP0, PL1 = HR * WP1;   // pixels per plane
constexpr int BUFB = 80 * 1024;                 // bytes of one halo buffer: 80 DMA pieces of 1 KiB (1273 px x 64 B + pad)
constexpr int NPIECE = 20;                      // pieces per thread per chunk (4 waves x 20)
constexpr int D = 5;                            // weight-fragment register sets (a tap's set is loaded D-1 taps ahead)
constexpr int NTAP = 25;
// halo DMA pieces issued at tap t of a chunk (two per tap over the first ten taps: they must have landed by the
// barrier at tap 24, and the weight set that tap waits for was issued at tap 20)
constexpr int np(int t) { return (((t % NTAP) + NTAP) % NTAP) < NPIECE / 2 ? 2 : 0; }
// VMEM operations issued after the weight set of tap t: the pieces of the tap it was issued in (t - (D-1)), then
// D-2 whole taps of NB weight loads + pieces
constexpr int younger(int t, int NB) {
  int n = np(t - (D - 1));
  for (int k = t - (D - 2); k <= t - 1; ++k) n += NB + np(k);
  return n;
}
}  // namespace halo

typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
#ifdef LIC_HALO_ABLATE
__device__ unsigned long long g_halo_dbg[1024 * 10];
#define HALO_STAMP(i)                                                    \
  do {                                                                   \
    st[i] = __builtin_amdgcn_s_memtime();                                \
    st[5 + i] = __builtin_amdgcn_s_memrealtime();                        \
  } while (0)
#else
#define HALO_STAMP(i)
#endif

template <int TW, bool FUSE = false, int ABL = 0>
__global__ __launch_bounds__(256, 1) void halo_conv_bf16_kernel(const IgemmHParams p) {
  using namespace halo;
  constexpr int NB = 2 * TW;                      // weight fragment loads per tap per lane
  constexpr int PANEL = 64 * TW * HB_BK * 2;      // bytes of one (tap, chunk) weight panel: Npad x 32 x bf16
  constexpr int NM = 8 * TW;                      // MFMAs per tap per wave
  static_assert(younger(0, 2 * TW) <= 63 && younger(12, 2 * TW) <= 63, "vmcnt is a 6-bit counter");
  static_assert(8 + NB + 1 + 2 <= NM, "a tap's other work is interleaved one piece per MFMA");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUFB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
#ifdef LIC_HALO_ABLATE
  unsigned long long st[10];
  for (int i = 0; i < 10; ++i) st[i] = 0;
#endif
  HALO_STAMP(0);

  const int G = gridDim.x, ntiles = p.MT;
  int tile = blockIdx.x;
  {  // XCD-aware bijective remap: an XCD's workgroups walk neighbouring tiles (shared halo rows / columns in its L2)
    const int q = G >> 3, r = G & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }

  // ---- halo DMA: piece (4 j + wave) of a buffer = LDS bytes [1024 (4 j + wave), +1024); lane -> 16-byte slot
  // n = 64 (4 j + wave) + lane = (plane-linear pixel n >> 2, physical octet n & 3).  Tile-independent per lane:
  // rel[j] = BYTE offset of the slot's source from the halo origin; meta[j >> 1] packs, 16 bits per piece, the
  // slot's halo row (5 bits) | column << 5 (7 bits) | pad << 15.  The pieces are `buffer_load_dwordx4 ... lds` on a
  // descriptor whose base is the tile's halo origin: a lane whose slot lies outside the image gets an offset past
  // num_records and the hardware writes zeros without touching memory -- per piece that is 3 VALU + the load,
  // where a flat address with a zero-page select took a dozen scalar / vector instructions (the pieces cost 17 %
  // of the loop's cycles that way: in-kernel stamps with and without them).
  unsigned rel[NPIECE];
  unsigned meta[NPIECE / 2];
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) {
    const int n = (j * 4 + wave) * 64 + lane;
    const int P = n >> 2, slot = n & 3;
    const int pl = P >= PL0 ? 1 : 0;
    const int Pp = pl ? P - PL0 : P;
    const int hr = pl ? Pp / WP1 : Pp / WP0;
    const int hc = pl ? Pp - hr * WP1 : Pp - hr * WP0;
    const int o = slot ^ ((hc >> 2) & 3);
    const int col = 2 * hc + pl;
    rel[j] = 2u * (unsigned)((hr * p.Wi + col) * (int)p.in_ld + o * 8);
    const unsigned m = (unsigned)(hr & 31) | ((unsigned)col << 5) | (hr < HR ? 0u : 0x8000u);
    if (j & 1) meta[j >> 1] |= m << 16;
    else meta[j >> 1] = m;
  }
  struct Geo {
    int b, oy0, ox0;
    long base;      // element offset of the halo origin (may lie before the image: masked lanes never fetch)
    unsigned vmask; // bit j: piece j of this lane reads the image (else the zero page)
  };
  auto geom = [&](int t) {
    Geo g;
    const int tpi = p.htx * p.hty;
    g.b = t / tpi;
    const int trem = t - g.b * tpi;
    const int ty = trem / p.htx, tx = trem - ty * p.htx;
    g.oy0 = ty * TH;
    g.ox0 = tx * TWD;
    const int iy0 = 2 * g.oy0 - 2, ix0 = 2 * g.ox0 - 2;
    g.base = ((long)(g.b * p.Hi + iy0) * p.Wi + ix0) * p.in_ld;
    unsigned vm = 0;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      unsigned mm = meta[j >> 1];
      asm volatile("" : "+v"(mm));  // (opaque: hipcc otherwise keeps all 40 unpacked fields live across the main loop)
      const unsigned m = (mm >> (16 * (j & 1))) & 0xFFFFu;
      // (unsigned compares: 0 <= iy < Hi and 0 <= ix < Wi in one test each; branch-free)
      const unsigned iy = (unsigned)(iy0 + (int)(m & 31u)), ix = (unsigned)(ix0 + (int)((m >> 5) & 127u));
      const unsigned ok = (unsigned)(iy < (unsigned)p.Hi) & (unsigned)(ix < (unsigned)p.Wi) & ((m >> 15) ^ 1u);
      vm |= ok << j;
    }
    g.vmask = vm;
    return g;
  };
  const int nch = p.cpt;  // 32-channel chunks (even)
  constexpr unsigned OOB = 0xFFFFFFF0u, NREC = 0x7FFFFFF0u;
  auto halo_rsrc = [&](long base) {  // (wave-uniform by construction: kernel arguments and the tile index)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.in + base), 0, NREC, 0x00020000);
  };
  // soff: byte offset of the channel chunk; vm = 0 (past the last tile): every lane out of range, zeros into the
  // idle buffer, which keeps the vmcnt arithmetic exact
  auto dma_piece = [&](int j, __amdgpu_buffer_rsrc_t rs, int soff, unsigned vm, int buf) {
    const unsigned off = ((vm >> j) & 1u) ? rel[j] : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lich_lptr_t)(smem + buf * BUFB + (j * 4 + wave) * 1024), 16, (int)off, soff,
                                             0, 0);
  };

  // ---- A fragment addresses: lane (li, lh) of row tile a reads pixel (row 2 (4 wm + a) + r, column li + (s >> 1)) of
  // plane s & 1, K octet (2 ks + lh) ^ swizzle.  abase[buf][plane x shift][ks]; the tap and the row tile are the
  // immediate offset (2 a + r) * row bytes.
  unsigned abase[2][5][2];
  {
    const unsigned s0 = (unsigned)(size_t)(lich_lptr_t)smem;
#pragma unroll
    for (int ps = 0; ps < 5; ++ps) {
      const int pl = ps >= 3 ? 1 : 0, sh = pl ? ps - 3 : ps;
      const int hc = li + sh;
      const int rowb = (pl ? WP1 : WP0) * 64;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned a0 = s0 + (pl ? PL0 * 64 : 0) + 8 * wm * rowb + hc * 64 + 16 * ((2 * ks + lh) ^ ((hc >> 2) & 3));
        abase[0][ps][ks] = a0;
        abase[1][ps][ks] = a0 + BUFB;
      }
    }
  }

  // ---- weight fragments: lane's 16 bytes of column tile (wn TW + t), k step ks of the (tap, chunk) panel.
  // bptr walks the panels in the order the taps consume them: tap 0..24 of chunk 0, tap 0..24 of chunk 1, ...,
  // and from the last panel back to the first (the next tile's)
  const char* bptr = reinterpret_cast<const char*>(p.w) + (long)wn * TW * 2048 + lane * 16;
  const long tap_inc = (long)nch * PANEL;                     // next tap, same chunk
  const long wrap_next = PANEL - (long)(NTAP - 1) * tap_inc;  // tap 24 of chunk c -> tap 0 of chunk c + 1

  f32x16 acc[4][TW];
  hu32x4 af[2][4][2];   // [slot][row tile][k step]
  hu32x4 bq[D][TW][2];  // [slot][column tile][k step]
  auto& af_ = af;
  auto& bq_ = bq;

  auto a_read1 = [&](auto slotc, auto bufc, auto tapc, auto ic) {  // one of the 8 fragment reads of tap `tapc`
    constexpr int SLOT = decltype(slotc)::value, BUF = decltype(bufc)::value, TAP = decltype(tapc)::value;
    constexpr int I = decltype(ic)::value, a = I >> 1, ks = I & 1;
    constexpr int r = TAP / 5, s = TAP % 5, pl = s & 1, sh = s >> 1, ps = pl ? 3 + sh : sh;
    constexpr int rowb = (pl ? WP1 : WP0) * 64;
    auto& dst = af[SLOT][a][ks];  // (named outside the asm: clang does not capture a variable a generic lambda only uses there)
    const unsigned addr = abase[BUF][ps][ks];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "i"((2 * a + r) * rowb));  // (A fragments live in AGPRs)
  };
  auto b_load1 = [&](auto slotc, auto ic) {  // one of the NB weight loads of a tap
    constexpr int SLOT = decltype(slotc)::value, I = decltype(ic)::value, t = I >> 1, ks = I & 1;
    static_assert(I * 1024 - 2048 < 4096, "13-bit signed offset");
    auto& dst = bq[SLOT][t][ks];
    const char* src = bptr;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(src), "i"(I * 1024 - 2048));
  };
  // wait for this tap's fragments: all LDS reads, and the weight loads with `N` younger VMEM operations in flight
  auto wait_frags = [&](auto slota, auto slotb, auto nc) {
    constexpr int SA = decltype(slota)::value, SB = decltype(slotb)::value, N = decltype(nc)::value;
    auto& af = af_;
    auto& bq = bq_;
    if constexpr (TW == 2)
      asm volatile("s_waitcnt vmcnt(%12) lgkmcnt(0)"
                   : "+a"(af[SA][0][0]), "+a"(af[SA][0][1]), "+a"(af[SA][1][0]), "+a"(af[SA][1][1]), "+a"(af[SA][2][0]),
                     "+a"(af[SA][2][1]), "+a"(af[SA][3][0]), "+a"(af[SA][3][1]), "+v"(bq[SB][0][0]), "+v"(bq[SB][0][1]),
                     "+v"(bq[SB][1][0]), "+v"(bq[SB][1][1])
                   : "i"(N));
    else
      asm volatile("s_waitcnt vmcnt(%14) lgkmcnt(0)"
                   : "+a"(af[SA][0][0]), "+a"(af[SA][0][1]), "+a"(af[SA][1][0]), "+a"(af[SA][1][1]), "+a"(af[SA][2][0]),
                     "+a"(af[SA][2][1]), "+a"(af[SA][3][0]), "+a"(af[SA][3][1]), "+v"(bq[SB][0][0]), "+v"(bq[SB][0][1]),
                     "+v"(bq[SB][1][0]), "+v"(bq[SB][1][1]), "+v"(bq[SB][2][0]), "+v"(bq[SB][2][1])
                   : "i"(N));
  };
  // registers with a load still in flight must stay allocated until it has landed: name them in a statement
  auto keep_set = [&](auto slotb) {
    constexpr int SB = decltype(slotb)::value;
    auto& bq = bq_;
    if constexpr (TW == 2)
      asm volatile("" : "+v"(bq[SB][0][0]), "+v"(bq[SB][0][1]), "+v"(bq[SB][1][0]), "+v"(bq[SB][1][1]));
    else
      asm volatile("" : "+v"(bq[SB][0][0]), "+v"(bq[SB][0][1]), "+v"(bq[SB][1][0]), "+v"(bq[SB][1][1]),
                   "+v"(bq[SB][2][0]), "+v"(bq[SB][2][1]));
  };

  // One tap: U = tap index inside the two-chunk loop body (0..49); U / 25 = halo buffer = chunk parity.
  //   wait -> (tap 24: chunk barrier) -> [MFMA i | one piece of the tap's other work] x NM
  // other work, in issue order: the 8 A reads of the NEXT tap (into the A slot tap U-1 used), the NB weight loads of
  // tap U + D-1 (into the set tap U-1 used) and the pointer step, then this tap's halo DMA pieces (drs / dsoff / dvm:
  // the chunk after this one -- of this tile, of the next tile, or nothing).
  auto tap_step = [&](auto uc, int c, __amdgpu_buffer_rsrc_t drs, int dsoff, unsigned dvm) {
    constexpr int U = decltype(uc)::value, T = U % NTAP, BUF = U / NTAP;
    constexpr int SA = U & 1, SB = U % D;
    constexpr int UN = (U + 1) % (2 * NTAP), TN1 = UN % NTAP, BUFN = UN / NTAP;  // the next tap
    wait_frags(std::integral_constant<int, SA>{}, std::integral_constant<int, SB>{},
               std::integral_constant<int, younger(T, NB)>{});
    if constexpr (T == NTAP - 1) {
      // every halo piece of the next chunk is older than the weight set just waited for: mine have landed; after the
      // barrier everyone's have, and everyone has read the last fragments of this chunk (lgkmcnt(0) above), so the
      // next tap may read the other buffer and the next chunk may refill this one
      asm volatile("s_barrier" ::: "memory");
    }
    lich_for_seq(std::make_integer_sequence<int, NM>{}, [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int ks = I / (4 * TW), rem = I % (4 * TW), t = rem / 4, a = rem % 4;
      // operands swapped: rows = the 32 channels of column tile t, columns (lanes) = the 32 pixels of row tile a
      acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[SB][t][ks]),
                                                           __builtin_bit_cast(bf16x8, af[SA][a][ks]), acc[a][t], 0, 0, 0);
      if constexpr (I < 8) {
        if constexpr (!(ABL & 2))
          a_read1(std::integral_constant<int, 1 - SA>{}, std::integral_constant<int, BUFN>{},
                  std::integral_constant<int, TN1>{}, ic);
      } else if constexpr (I < 8 + NB) {
        if constexpr (!(ABL & 1))
          b_load1(std::integral_constant<int, (U + D - 1) % D>{}, std::integral_constant<int, I - 8>{});
      } else if constexpr (I == 8 + NB) {
        // the set just requested was tap (T + D-1) % 25's; step to the next panel in consumption order
        if constexpr ((T + D - 1) % NTAP == NTAP - 1) {
          const int cl = c + (T + D - 1) / NTAP;  // the chunk that panel belonged to
          bptr += (cl + 1 < nch) ? wrap_next : wrap_next - (long)nch * PANEL;  // after the last chunk: the first again
        } else {
          bptr += tap_inc;
        }
      } else if constexpr (I < 8 + NB + 1 + np(T)) {
        if constexpr (!(ABL & 4)) dma_piece(2 * T + (I - 8 - NB - 1), drs, dsoff, dvm, 1 - BUF);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- prologue (first tile only): the weight sets of taps 0..D-2, chunk 0 into buffer 0, then tap 0's A fragments
  Geo cur = geom(tile);
  {
    const __amdgpu_buffer_rsrc_t rs0 = halo_rsrc(cur.base);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) dma_piece(j, rs0, 0, cur.vmask, 0);
  }
  bptr += 2048;  // (the loads address [-2048, +3072] around the pointer: 13-bit signed immediates)
  lich_for_seq(std::make_integer_sequence<int, D - 1>{}, [&](auto sc) {
    lich_for_seq(std::make_integer_sequence<int, NB>{}, [&](auto ic) { b_load1(sc, ic); });
    bptr += tap_inc;
  });
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"((D - 1) * NB) : "memory");
  lich_for_seq(std::make_integer_sequence<int, 8>{}, [&](auto ic) {
    a_read1(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ic);
  });
  HALO_STAMP(1);

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  for (int it = 0;; ++it) {
    const int tnext = tile + G;
    const bool more = tnext < ntiles;
    Geo nxt = cur;
    if (more) nxt = geom(tnext);
    const unsigned nvm = more ? nxt.vmask : 0u;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.0f;
    for (int c0 = 0; c0 < nch; c0 += 2) {
      // chunk c0 fetches chunk c0 + 1 of this tile; chunk c0 + 1 fetches chunk c0 + 2, or the next tile's first
      const bool last = c0 + 2 >= nch;
      const __amdgpu_buffer_rsrc_t rsA = halo_rsrc(cur.base), rsB = halo_rsrc(last ? nxt.base : cur.base);
      const int sA = (c0 + 1) * HB_BK * 2, sB = last ? 0 : (c0 + 2) * HB_BK * 2;
      const unsigned vB = last ? nvm : cur.vmask;
      lich_for_seq(std::make_integer_sequence<int, 2 * NTAP>{}, [&](auto uc) {
        constexpr int U = decltype(uc)::value;
        if constexpr (U < NTAP) tap_step(uc, c0, rsA, sA, cur.vmask);
        else tap_step(uc, c0 + 1, rsB, sB, vB);
      });
    }
    if (it == 0) HALO_STAMP(2);

    if constexpr (FUSE) {
      // ---- conv -> GDN / IGDN in the same launch (LIC_EPI_CONV_GDN / CONV_IGDN; Components.py:12-15), as the fused
      // variant of igemm_bf16_kernel does it: x = conv + bias rounded to bf16, x^2 rounded again is the B operand of
      // norm^T = gamma_eff . (x^2)^T in the K order a lane owns its channels in (gamma_eff^T packed by
      // lic_pack_weight_bf16_kperm), y = x * norm^-1/2 (or ^1/2) element-wise in the accumulator layout.  Here a wave
      // holds only HALF the channels of its pixels (2 x 2 waves), so the x^2 fragments are exchanged through LDS:
      // every wave writes its 16 fragments (4 row tiles x 2 channel tiles x 2 k steps, lane-linear 1 KiB each) into
      // the halo buffer that is idle now (buffer 1: buffer 0 already holds the next tile's first chunk), one barrier,
      // and reads back all 32 fragments of its pixel half -- its partner's lanes own the same pixels, so the exchange
      // is a lane-wise copy.  A second barrier at the end keeps the next tile's DMA out of the buffer until every
      // wave has read.
      int lho = lh, lio = li, wno = wn, wmo = wm;
      asm volatile("" : "+v"(lho), "+v"(lio), "+s"(wno), "+s"(wmo));
      const bool inv = p.epilogue == LIC_EPI_CONV_IGDN;
      unsigned char* xch = smem + BUFB;                       // [wm][a][channel tile 0..2TW-1][k step][64 lanes][16 B]
      auto frag_at = [&](int wmi, int a, int tt, int s2) { return xch + ((((wmi * 4 + a) * (2 * TW) + tt) * 2 + s2) * 64 + lane) * 16; };
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
      // (gamma_eff^T fragments and beta_eff are requested NOW, all at once: a load in front of each pool MFMA /
      // each finish tile was one exposed L2 round trip after the other at one wave per SIMD -- the fused epilogue
      // took 14 us per tile that way)
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
      // 1. x -> bf16 (kept in the accumulators as the rounded value), x^2 -> bf16 -> LDS; the conv output if asked for
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int oy = cur.oy0 + 4 * wmo + a, ox = cur.ox0 + lio;
        const bool rok = oy < p.Ho && ox < p.Wo;
        const long opix = rok ? ((long)cur.b * p.Ho + oy) * p.Wo + ox : 0;
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
      // 2. + 3. two row tiles at a time (register budget): pool over all 2 TW input channel tiles, then finish
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
          const int oy = cur.oy0 + 4 * wmo + a, ox = cur.ox0 + lio;
          const bool rok = oy < p.Ho && ox < p.Wo;
          const long opix = rok ? ((long)cur.b * p.Ho + oy) * p.Wo + ox : 0;
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
      __syncthreads();   // (the next tile's second chunk is DMA'd into the exchange buffer)
    } else
    // ---- epilogue, straight from the registers: lane (li, lh) holds, of tile (a, t), pixel li of output row
    // 4 wm + a and channels 32 (wn TW + t) + 8 g + 4 lh + {0..3}, g = 0..3.  LeakyReLU as max(v, slope v) (slope 1 =
    // none); the bias is fetched in one batch (a load per tile would drain the stores in flight every time).
    {
      // (opaque copies: hipcc otherwise hoists every address below out of the tile loop and spills it across the
      // main loop)
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
      const float sl = p.epilogue == LIC_EPI_LEAKY ? p.slope : 1.0f;
      const bool of32 = p.out_f32 != 0;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int oy = cur.oy0 + 4 * wmo + a, ox = cur.ox0 + lio;
        const bool rok = oy < p.Ho && ox < p.Wo;
        const long opix = rok ? ((long)cur.b * p.Ho + oy) * p.Wo + ox : 0;
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
              pk[2 * g] = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, bf16x2));
              pk[2 * g + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, bf16x2));
            }
            // lanes li and li + 32 exchange halves: each ends up with 8 consecutive channels -> 16-byte stores
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const u32x2 r0 = __builtin_amdgcn_permlane32_swap(pk[4 * s2], pk[4 * s2 + 2], false, false);
              const u32x2 r1 = __builtin_amdgcn_permlane32_swap(pk[4 * s2 + 1], pk[4 * s2 + 3], false, false);
              if (rok) {
                const hu32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                *reinterpret_cast<hu32x4*>(reinterpret_cast<bf16_t*>(p.out) + opix * p.out_ld + cb + 16 * s2 + 8 * lho) = o;
              }
            }
          }
        }
      }
    }
    if (it == 0) HALO_STAMP(3);
    if (!more) break;
    cur = nxt;
    tile = tnext;
  }
  // in flight: the (unused) weight sets of the four taps past the end, A slot 0, zero pieces into buffer 0
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+a"(af[0][0][0]), "+a"(af[0][0][1]), "+a"(af[0][1][0]), "+a"(af[0][1][1]), "+a"(af[0][2][0]),
                 "+a"(af[0][2][1]), "+a"(af[0][3][0]), "+a"(af[0][3][1]));
  lich_for_seq(std::make_integer_sequence<int, D - 1>{}, keep_set);
#ifdef LIC_HALO_ABLATE
  HALO_STAMP(4);
  if (tid == 0 && blockIdx.x < 1024)
    for (int i = 0; i < 10; ++i) g_halo_dbg[blockIdx.x * 10 + i] = st[i];
#endif
}
