// NOT BUILT INTO THE LIBRARY.  The per-wave software-pipeline variant of the sparse-conv forward / dgrad kernel as it was
// measured in round 5 (profiles/r05_conv_study.md, r05_conv_fat_ab*.txt; commit e8cc02a, where spconv_tiles.hip included it
// as csrc/spconv_fat.h and dispatched to it under EFG_CONV_FAT).  Bit-exact against every spconv test; within +-5 % of the
// stream-K tile kernel on every level, so the tile kernel stayed.  Kept here so that the study's code can be read.
// Sparse convolution forward / dgrad over a tile plan, third generation: ONE LONG SOFTWARE PIPELINE PER WAVE.
// (included by spconv_tiles.hip: same translation unit, same TileArgs, same plan)
//
// What the knock-out builds of conv_tile_kernel said (profiles/r05_conv_knock.txt): with every gather, weight load and LDS
// access removed the launch still takes 75 / 99 / 130 us on the 64 / 128 / 256-channel submanifold levels against
// 43 / 63 / 89 us of MFMA issue.  A level is ~50-100 (16 rows x 64 channels x 64 channels) steps per SIMD and the tile
// kernel spreads them over ~4 waves per SIMD x several units, so that a wave executes a handful of steps between a
// prologue of dependent plan loads behind a barrier and an epilogue of barriers, an LDS reduction by one wave and a
// share exchange through global memory: the fixed parts are a third of a wave's life and every wave of the chip is
// in them at the same time.  (Measured and dropped on the way, profiles/r05_conv_walk_ab.txt: one workgroup of
// KS = reduction-chunk waves per unit with a balanced two-barrier reduction, dispatched by the hardware or drawn from
// per-XCD queues -- 5-30 % SLOWER than the tile kernel: whole units are too coarse to balance 1024 SIMDs.)
//
// Here a wave is alone responsible for a contiguous range of the level's (n-slice, unit, active offset) items -- the
// same equal split stream-K makes, but per WAVE (1 or 2 per SIMD) and walked as one software pipeline:
//  * no workgroup cooperation: no barrier, no LDS reduction, no dealing of steps (a 256-thread workgroup is just four
//    independent waves);
//  * the NEXT unit's plan rows (masks, row numbers, neighbour block) are loaded during the current unit's steps, its
//    neighbour offsets go to the other half of a double-buffered wave-private LDS table, and its first gather is issued
//    before the current unit's last MFMAs: a unit boundary costs no dependent memory round trip;
//  * a unit cut by a range boundary is summed by whichever of its waves ARRIVES LAST (a ticket per unit): each wave
//    writes its partial tile through to memory and draws a ticket; the last one adds the partials in wave order -- a
//    fixed order whoever is last, so results are bit-reproducible -- and stores the rows.  Nobody waits for anybody:
//    no flags, no epochs, no bounded spin, no dependence on dispatch order, capturable into a HIP graph (the tickets
//    reset themselves).
// The step itself (wave-private A tile in LDS from coalesced 4-byte gathers, weights as MFMA fragments from L2, fp32
// v_mfma_f32_16x16x4_f32) is conv_tile_kernel's.
#pragma once

namespace efg {
namespace {

#ifndef EFG_FKNOCK
#define EFG_FKNOCK 0   // diagnostic A/B builds only: 1 no gather loads, 2 no weight loads, 4 no LDS stash, 8 no MFMAs
#endif
#ifndef EFG_FAT_DEEP
#define EFG_FAT_DEEP 0   // 1: the weights of a whole step requested a step ahead (128 weight registers: one wave per SIMD)
#endif
#ifndef EFG_FAT_WPE
#define EFG_FAT_WPE 2   // waves per SIMD asked of the register allocator (A/B builds: scripts/build_ab.sh)
#endif

// TileArgs fields used differently here: sk_prefix = the plan's item prefix for this R, sk_scratch = partial tiles
// [waves][2][R * NT * 256], sk_flags = tickets [uy * units], ux = units of the plan (incl. padding units), uy = n-slices.
// 16-byte write-through stores / L1-bypassing loads (sc1) of the partial tiles: agent-coherent without a fence (see
// conv_tile_kernel's stream-K note); the 4-byte forms of __hip_atomic_store are one fabric write each.  The loads come in
// fours with their wait inside the statement (the compiler does not count the loads of an asm statement).
__device__ __forceinline__ void st16_sc1(f32x4* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld16x4_sc1(const f32x4* p, f32x4& v0, f32x4& v1, f32x4& v2, f32x4& v3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(p)
      : "memory");
}

__device__ __forceinline__ unsigned walk_dummy_xcd() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }

template <int NT, int R>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EFG_FAT_WPE)))
conv_fat_kernel(TileArgs a) {
  constexpr int kAStr = kCKt + 2;
  constexpr int NBV = (R * 31 * 16 + 63) / 64;   // neighbour-block registers per lane (kvol <= 31)
  constexpr int kSlot = R * NT * 4 * 64;         // floats of a partial tile
  __shared__ __attribute__((aligned(16))) float a_tile[4][R * 16 * kAStr];   // wave-private A staging
  __shared__ int nb_tile[4][2][R * 32 * 16 + 16];                             // byte offsets of neighbour rows, [sub][k][16] (+ a dump slot)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* at0 = a_tile[wv];
  const int nchunk = (a.c16n * 16 + kCKt - 1) / kCKt;
  const int ne = R * a.kvol * 16;

  // ---- this wave's range of the item list ----
  // XCD x (= blockIdx.x & 7: the hardware deals workgroups round-robin) takes the x-th contiguous eighth of the waves, so
  // that an L2 sees one range of rows and one slice of the weights
  const unsigned nwg = gridDim.x, per8 = nwg >> 3, b = blockIdx.x;
  const unsigned pb = b < (per8 << 3) ? (b & 7u) * per8 + (b >> 3) : b;
  const int* __restrict__ P = a.sk_prefix;
  // (all range arithmetic in 32 bits: the host checked items x waves < 2^31; a 64-bit division is ~300 cycles and the
  // range ends, the ticket logic and the first unit need a dozen of them)
  // first round of the 64-ary search for this wave's first unit: its probes do not depend on the item count, so they are
  // requested together with it (one round trip instead of two)
  const unsigned step1 = (a.ux + 63) / 64;
  const unsigned pos1 = min((unsigned)(lane + 1) * step1, a.ux);
  const int probe1 = P[pos1];
  const unsigned C = (unsigned)P[a.ux], T = C * a.uy;
  const unsigned G = min(nwg * 4, T);
  const unsigned g = pb * 4 + (unsigned)wv;
  if (g >= G) return;
  auto lo_of = [&](unsigned gg) { return (gg * T) / G; };   // (gg * T < 2^32: checked by the host)
  auto wave_of = [&](unsigned x) {   // largest gg with lo_of(gg) <= x
    unsigned cand = (x * G) / T;
    while (cand + 1 < G && lo_of(cand + 1) <= x) ++cand;
    while (cand > 0 && lo_of(cand) > x) --cand;
    return cand;
  };
  auto unit_of = [&](unsigned x) {   // largest w with P[w] <= x (x < C): 64-ary search, wave-uniform
    unsigned lo = 0, hi = a.ux;
    bool first = true;
    while (hi - lo > 1) {
      const unsigned span = hi - lo, step = (span + 63) / 64;
      const unsigned pos = min(lo + (unsigned)(lane + 1) * step, hi);
      const int pv = first ? probe1 : P[min(pos, a.ux)];
      first = false;
      const bool le = pos < hi ? ((unsigned)pv <= x) : false;
      const int nle = __popcll(__ballot(le));
      const unsigned nlo = nle ? min(lo + (unsigned)nle * step, hi) : lo;
      const unsigned nhi = min(lo + (unsigned)(nle + 1) * step, hi);
      lo = nlo;
      hi = nhi;
    }
    return lo;
  };
  const unsigned my_lo = lo_of(g), my_hi = lo_of(g + 1);
  // debug (efg_spconv_fat_debug): per wave {start, end of the s_memtime clock, steps by active sub-tiles, units, cut units}
  long long* dbg = reinterpret_cast<long long*>(a.sk_fallbacks);
  const long long t_start = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
  int d_full = 0, d_half = 0, d_units = 0, d_cut = 0;
  long long c_store = 0, c_ticket = 0, c_sum = 0, c_rows = 0, c_open = 0, c_prol = 0;
  auto clk = [&]() { return dbg ? (long long)__builtin_amdgcn_s_memtime() : 0ll; };

  // ---- plan rows of a unit ----
  auto load_plan = [&](unsigned w, unsigned (&vm_o)[R], int& prow_o, int (&nbv)[NBV]) {
    const long long t0 = (long long)w * R;
#pragma unroll
    for (int s = 0; s < R; ++s) vm_o[s] = a.vm[(t0 + s) * 32 + (lane & 31)];
    prow_o = a.rows[t0 * 16 + (lane & (R * 16 - 1))];
    const int* src = a.nb + t0 * a.kvol * 16;
#pragma unroll
    for (int i = 0; i < NBV; ++i) nbv[i] = src[min(lane + i * 64, ne - 1)];
  };
  auto store_nbs = [&](int* tab, const int (&nbv)[NBV]) {
#pragma unroll
    for (int i = 0; i < NBV; ++i) {
      const int e = lane + i * 64;
      const int s = (R == 1) ? 0 : (e >= a.kvol * 16 ? 1 : 0);
      const int rem = e - s * (a.kvol * 16);
      // (entries past the block go to the dump slot: a select, not a branch per register)
      tab[e < ne ? s * 512 + rem : R * 512] = (int)((unsigned)max(nbv[i], 0) * (unsigned)a.cin * 4u);
    }
  };
  // the columns of rank [j0, j1) among the active columns `cols`
  auto select_cols = [&](unsigned cols, int j0, int j1) {
    unsigned sub = cols;
    for (int i = 0; i < j0; ++i) sub &= sub - 1;
    unsigned keep = 0;
    for (int i = j0; i < j1 && sub; ++i) {
      keep |= sub & (0u - sub);
      sub &= sub - 1;
    }
    return keep;
  };

  // ---- the step machinery: 4-byte gathers into a wave-private A tile, weights as MFMA fragments; every load of the
  // pipeline is unconditional and issued in the order its data is needed (the memory counter is in order: a wait for a
  // young load is a wait for every older one)
  f32x4 acc[R][NT];
  float pre[R * 16];
  unsigned pre_m[R];
  auto gather = [&](const unsigned (&vmq)[R], const int* nbs, int col, int ch, bool live) {
    const unsigned cc4 = (unsigned)(ch * kCKt + lane) * 4u;   // (cin is a multiple of 64 here)
#pragma unroll
    for (int s = 0; s < R; ++s) {
      pre_m[s] = live ? (unsigned)__builtin_amdgcn_readlane((int)vmq[s], col) : 0u;
      unsigned offs[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) offs[j] = (unsigned)nbs[s * 512 + col * 16 + j] + cc4;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (EFG_FKNOCK & 1) pre[s * 16 + j] = __int_as_float((int)offs[j]);
        else
        pre[s * 16 + j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) + offs[j]);
      }
    }
  };
  auto stash = [&]() {   // absent neighbours (and the rows of a dead step) are zeroed by a bit mask: no branch per row
    if (EFG_FKNOCK & 4) {
      at0[lane] = pre[0] + pre[R * 16 - 1];
      return;
    }
#pragma unroll
    for (int s = 0; s < R; ++s) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int keep = -(int)((pre_m[s] >> j) & 1u);
        at0[(s * 16 + j) * kAStr + lane] = __int_as_float(__float_as_int(pre[s * 16 + j]) & keep);
      }
    }
  };
  const unsigned bstep = (unsigned)a.np * 64u;
  auto b_off = [&](int col, int ch, int ntile) {
    const int k = a.flip ? (a.kvol - 1 - col) : col;
    const int c16_lo = ch * (kCKt / 16);
    const int m = lane & 15, kk = lane >> 4;
    return ((((unsigned)k * (unsigned)a.c16n + (unsigned)c16_lo) * (unsigned)a.np + (unsigned)(ntile * 16 + m)) * 16u + (unsigned)(kk * 4)) * 4u;
  };
  auto load_b = [&](float4 (&bq)[NT], unsigned boff0, int i) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (EFG_FKNOCK & 2) {
        const float f = __int_as_float((int)(boff0 + (unsigned)i * bstep + (unsigned)t * 1024u));
        bq[t] = make_float4(f, f, f, f);
      } else
      bq[t] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.wp) + boff0 + (unsigned)i * bstep + (unsigned)t * 1024u);
    }
  };
  auto read_a = [&](float (&af)[R * 4], int i) {
    const int m = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const float* ap = at0 + (s * 16 + m) * kAStr + i * 16 + kk;
      af[s * 4 + 0] = ap[0];
      af[s * 4 + 1] = ap[4];
      af[s * 4 + 2] = ap[8];
      af[s * 4 + 3] = ap[12];
    }
  };
  auto mfma_blk = [&](const float (&af)[R * 4], const float4 (&bq)[NT], unsigned m0, unsigned m1) {
#pragma unroll
    for (int s = 0; s < R; ++s) {
      if ((s == 0 ? m0 : m1) == 0) continue;   // wave-uniform: this sub-tile has no neighbour at the column
      if (EFG_FKNOCK & 8) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t][0] += (af[s * 4 + 0] + af[s * 4 + 1] + af[s * 4 + 2] + af[s * 4 + 3]) * (bq[t].x + bq[t].y + bq[t].z + bq[t].w);
        continue;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + 0], bq[t].x, acc[s][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + 1], bq[t].y, acc[s][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + 2], bq[t].z, acc[s][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + 3], bq[t].w, acc[s][t], 0, 0, 0);
    }
  };

  // ---- the end of a unit: rows stored, or a partial tile and a ticket ----
  auto finish_unit = [&](int prow_c, int ntile_c, unsigned base, int n, int j0, int j1, unsigned vid) {
    const bool whole = (j0 == 0 && j1 >= n);
    ++d_units;
    if (!whole) {
      ++d_cut;
      const long long k0 = clk();
      const unsigned g_first = wave_of(base), g_last = wave_of(base + n - 1);
      const unsigned slot = my_lo >= base ? 0u : 1u;   // the unit this wave's range STARTS in is its slot 0
      // lane-major: a lane's 32 (16) values are contiguous -- eight (four) 16-byte write-through stores per lane; 4-byte
      // write-through stores are one fabric write each
      f32x4* dst = reinterpret_cast<f32x4*>(a.sk_scratch + ((size_t)g * 2 + slot) * kSlot) + lane * (R * NT);
#pragma unroll
      for (int s = 0; s < R; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) st16_sc1(dst + s * NT + t, acc[s][t]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const long long k1 = clk();
      c_store += k1 - k0;
      int ticket = 0;
      if (lane == 0) ticket = atomicAdd(a.sk_flags + vid, 1);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      const long long k2 = clk();
      c_ticket += k2 - k1;
      if ((unsigned)ticket != g_last - g_first) return;   // somebody else arrives later and finishes the unit
      if (lane == 0) __hip_atomic_store(a.sk_flags + vid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the partials in wave order (this wave's own from its registers: the same bits it wrote)
      f32x4 tot[R][NT];
#pragma unroll
      for (int s = 0; s < R; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) tot[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (unsigned gp = g_first; gp <= g_last; ++gp) {
        if (gp == g) {
#pragma unroll
          for (int s = 0; s < R; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) tot[s][t] += acc[s][t];
        } else {
          const unsigned sl = lo_of(gp) >= base ? 0u : 1u;
          const f32x4* src = reinterpret_cast<const f32x4*>(a.sk_scratch + ((size_t)gp * 2 + sl) * kSlot) + lane * (R * NT);
          f32x4 v[R * NT];
          static_assert(NT == 4, "partial tiles are read four 16-byte pieces at a time");
#pragma unroll
          for (int q = 0; q < R; ++q) ld16x4_sc1(src + q * 4, v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
#pragma unroll
          for (int s = 0; s < R; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) tot[s][t] += v[s * NT + t];
        }
      }
#pragma unroll
      for (int s = 0; s < R; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = tot[s][t];
      c_sum += clk() - k2;
    }
    const long long k3 = clk();
    // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = (ntile_c + t) * 16 + (lane & 15);
      const float bias = (a.bias && co < a.cout) ? a.bias[co] : 0.0f;
#pragma unroll
      for (int s = 0; s < R; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = s * 16 + (lane >> 4) * 4 + r;
          const int row = __shfl(prow_c, j, 64);
          if (row >= 0 && co < a.cout) a.out[(long long)row * a.cout + co] = acc[s][t][r] + bias;
        }
    }
    c_rows += clk() - k3;
  };

  // ---- the walk ----
  // Two cursors move over the wave's range: the GATHER side (whose loads are issued a step ahead) and the COMPUTE side.
  // The gather side opens a unit (writes its neighbour table, selects its columns in range) when it gets there -- from plan
  // rows that were requested when the previous unit was opened -- and leaves the unit's description for the compute side.
  struct Unit {
    unsigned base;    // first item of the unit in the list
    int n, j0, j1;    // its items; the ranks this wave computes
    int prow;         // lane l: row l of the unit (-1: padding)
    int ntile;        // first 16-channel output tile of its slice
    unsigned vid;     // ticket index
    int steps;        // steps of this wave in it (>= 1: a unit without a step gets one dead step, so that it is finished)
  };
  // gather side
  unsigned vmr[R];
  int tsel = 0;
  int* tab = nb_tile[wv][0];
  unsigned rem = 0;      // columns of the unit the gather side has not left yet
  int c_g = 0, ch_g = 0;
  bool live_g = false;   // false: the unit's dead step
  Unit ug;               // the unit the gather side is in
  unsigned x_end_g = 0;    // first item after the gather side's unit (in this wave's range)
  unsigned yy = my_lo / C;
  unsigned w = unit_of(my_lo - yy * C);
  // prefetched plan rows of the unit after the gather side's
  unsigned vmr_n[R];
  int prow_n = -1;
  int nbv[NBV];
  int pn0 = 0, pn1 = 0;
  unsigned w_n = 0;
  unsigned yy_n = 0;
  bool more_n = false;

  auto request_next = [&]() {   // plan rows of the unit after (yy, w); past the end of the range: this unit's once more
    more_n = x_end_g < my_hi;
    w_n = w + 1;
    yy_n = yy;
    if (x_end_g - yy * C >= C) {   // the slice's items are used up: on to the next slice
      w_n = 0;
      yy_n = yy + 1;
    }
    const unsigned w_l = more_n ? w_n : w;
    load_plan(w_l, vmr_n, prow_n, nbv);
    pn0 = P[w_l];
    pn1 = P[w_l + 1];
  };
  auto open_unit = [&](int j0_new) {   // the gather side enters the prefetched unit
    const long long k0 = clk();
    store_nbs(nb_tile[wv][tsel ^ 1], nbv);
    tsel ^= 1;
    tab = nb_tile[wv][tsel];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned cols = 0;
#pragma unroll
    for (int s = 0; s < R; ++s) {
      vmr[s] = vmr_n[s];
      cols |= (unsigned)__builtin_amdgcn_readlane((int)vmr[s], 31);
    }
    w = w_n;
    yy = yy_n;
    ug.base = yy * C + (unsigned)pn0;
    ug.n = pn1 - pn0;
    ug.j0 = j0_new;
    ug.j1 = (int)min((unsigned)ug.n, my_hi - ug.base);
    ug.prow = prow_n;
    ug.ntile = (int)yy * NT;
    ug.vid = (unsigned)(yy * a.ux + w);
    rem = select_cols(cols, ug.j0, ug.j1);
    live_g = rem != 0;
    ug.steps = live_g ? __popc(rem) * nchunk : 1;
    c_g = live_g ? __ffs((int)rem) - 1 : 0;
    ch_g = 0;
    x_end_g = ug.base + ug.j1;
    request_next();
    c_open += clk() - k0;
  };

  // first unit: its plan rows synchronously
  {
    w_n = w;
    yy_n = yy;
    load_plan(w, vmr_n, prow_n, nbv);
    pn0 = P[w];
    pn1 = P[w + 1];
    const unsigned base0 = yy * C + (unsigned)pn0;
    open_unit((int)(my_lo - base0));
  }
  Unit uc = ug;   // compute side
  int steps_left = uc.steps;
  int c_c = c_g, ch_c = ch_g;
#pragma unroll
  for (int s = 0; s < R; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weights of a whole step (16 fragments, 64 registers) x 2: the NEXT step's are requested a full step ahead
  // (EFG_FAT_DEEP = 0: one set of 16 fragments; blocks 2, 3 of the current and 0, 1 of the next step are requested during a step)
  float4 Bq[EFG_FAT_DEEP ? 2 : 1][4][NT];
  int par = 0;
  gather(vmr, tab, c_g, ch_g, live_g);
  unsigned boff_c = b_off(c_c, ch_c, uc.ntile);
#pragma unroll
  for (int i = 0; i < (EFG_FAT_DEEP ? 4 : 2); ++i) load_b(Bq[0][i], boff_c, i);
  stash();
  unsigned cm0 = pre_m[0], cm1 = pre_m[R - 1];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  c_prol = clk() - t_start;
  while (true) {
    // ---- the gather side moves to the next step ----
    bool have_next = true, switched = false;
    if (live_g) ++ch_g;
    if (!live_g || ch_g >= nchunk) {
      ch_g = 0;
      rem &= rem - 1;
      if (live_g && rem) {
        c_g = __ffs((int)rem) - 1;
      } else if (more_n) {
        open_unit(0);
        switched = true;
      } else {
        have_next = false;
      }
    }
    // (no next step: the loads below repeat the current step's -- unconditional, nothing reads them)
    const int c_x = have_next ? c_g : c_c, ch_x = have_next ? ch_g : ch_c;
    const unsigned boff_x = b_off(c_x, ch_x, have_next ? ug.ntile : uc.ntile);

    // ---- step (c_c, ch_c) ----
    // 8 (4) groups of 16 MFMAs -- (16-channel block i, sub-tile s) -- each in ONE scheduling region together with a
    // portion of the step's other work, so that the scheduler can place those instructions in the shadow of the MFMAs
    // (a wave issues in order: a block of address arithmetic, LDS reads and loads between two MFMA blocks is time the
    // matrix pipe idles; measured with the MFMAs removed, that other work alone is 60-75 % of a step).  The portions:
    //   P0 / P1 the gather of sub-tile 0 / 1 of the NEXT step (needed first: at the end of this step)
    //   P2 .. P5 the weights of blocks 0 .. 3 of the NEXT step, into the other half of the weight registers: every load
    //   has at least half a step (2048 matrix-pipe cycles; an L2 hit under this load returns in ~1900) before it is needed
    // and the A fragments of block i + 1 are read during block i.  A sub-tile without a neighbour at the column is
    // skipped by picking another straight-line variant of the step (a branch around its MFMAs would end the region).
    {
      const bool live_x = have_next && live_g;
      const unsigned cc4 = (unsigned)(ch_x * kCKt + lane) * 4u;
#pragma unroll
      for (int s = 0; s < R; ++s) pre_m[s] = live_x ? (unsigned)__builtin_amdgcn_readlane((int)vmr[s], c_x) : 0u;
      const int* nbx = tab + c_x * 16;
      auto gather_rows = [&](auto sc, auto jc) {   // rows jc .. jc + 7 of sub-tile sc
        constexpr int s = decltype(sc)::value, j0 = decltype(jc)::value;
        if (s >= R) return;
        unsigned offs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) offs[j] = (unsigned)nbx[s * 512 + j0 + j] + cc4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (EFG_FKNOCK & 1) pre[s * 16 + j0 + j] = __int_as_float((int)offs[j]);
          else pre[s * 16 + j0 + j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) + offs[j]);
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I8 = std::integral_constant<int, 8>;
      auto read_as = [&](float (&af)[R * 4], int i, int s) {   // A fragment of block i, sub-tile s
        const int m = lane & 15, kk = lane >> 4;
        const float* ap = at0 + (s * 16 + m) * kAStr + i * 16 + kk;
        af[s * 4 + 0] = ap[0];
        af[s * 4 + 1] = ap[4];
        af[s * 4 + 2] = ap[8];
        af[s * 4 + 3] = ap[12];
      };
      auto mfma16 = [&](const float (&af)[R * 4], const float4 (&bq)[NT], int s) {
        if (EFG_FKNOCK & 8) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[s][t][0] += (af[s * 4 + 0] + af[s * 4 + 1] + af[s * 4 + 2] + af[s * 4 + 3]) * (bq[t].x + bq[t].y + bq[t].z + bq[t].w);
          return;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + 0], bq[t].x, acc[s][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + 1], bq[t].y, acc[s][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + 2], bq[t].z, acc[s][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s * 4 + 3], bq[t].w, acc[s][t], 0, 0, 0);
      };
#define EFG_SB __builtin_amdgcn_sched_barrier(0)
      // one step with the current weights in Bq[PC] and the next step's going to Bq[PC ^ 1]
      auto step = [&](auto pc) {
        constexpr int PC = EFG_FAT_DEEP ? decltype(pc)::value : 0, PN = EFG_FAT_DEEP ? (PC ^ 1) : 0;
        // portions of the other work.  Deep: the next step's gather first (it is needed first: at the end of this step), then
        // the next step's weights (needed from the start of the next step on).  Shallow: blocks 2 / 3 of this step ride with
        // the gather, blocks 0 / 1 of the next step come once their registers are free.
        auto P0 = [&]() {
          if (EFG_FAT_DEEP) { gather_rows(I0{}, I0{}); gather_rows(I0{}, I8{}); }
          else { load_b(Bq[0][2], boff_c, 2); gather_rows(I0{}, I0{}); }
        };
        auto P1 = [&]() {
          if (EFG_FAT_DEEP) { gather_rows(I1{}, I0{}); gather_rows(I1{}, I8{}); }
          else gather_rows(I0{}, I8{});
        };
        auto P2 = [&]() {
          if (EFG_FAT_DEEP) load_b(Bq[PN][0], boff_x, 0);
          else { load_b(Bq[0][3], boff_c, 3); gather_rows(I1{}, I0{}); }
        };
        auto P3 = [&]() {
          if (EFG_FAT_DEEP) load_b(Bq[PN][1], boff_x, 1);
          else gather_rows(I1{}, I8{});
        };
        auto P4 = [&]() {
          if (EFG_FAT_DEEP) load_b(Bq[PN][2], boff_x, 2);
          else load_b(Bq[0][0], boff_x, 0);
        };
        auto P5 = [&]() {
          if (EFG_FAT_DEEP) load_b(Bq[PN][3], boff_x, 3);
          else load_b(Bq[0][1], boff_x, 1);
        };
        float af0[R * 4], af1[R * 4];
        const bool a0 = cm0 != 0, a1 = R == 2 && cm1 != 0;
        if (a0 && a1) ++d_full; else if (a0 || a1) ++d_half;
        if (a0 && a1) {   // (R = 2) both sub-tiles: 8 regions
          constexpr int S1 = R - 1;
          read_as(af0, 0, 0); read_as(af0, 0, S1);
          EFG_SB; P0(); mfma16(af0, Bq[PC][0], 0);
          EFG_SB; P1(); read_as(af1, 1, 0); read_as(af1, 1, S1); mfma16(af0, Bq[PC][0], S1);
          EFG_SB; P2(); mfma16(af1, Bq[PC][1], 0);
          EFG_SB; P3(); read_as(af0, 2, 0); read_as(af0, 2, S1); mfma16(af1, Bq[PC][1], S1);
          EFG_SB; P4(); mfma16(af0, Bq[PC][2], 0);
          EFG_SB; P5(); read_as(af1, 3, 0); read_as(af1, 3, S1); mfma16(af0, Bq[PC][2], S1);
          EFG_SB; mfma16(af1, Bq[PC][3], 0);
          EFG_SB; mfma16(af1, Bq[PC][3], S1);
          EFG_SB;
        } else if (a0 || a1) {   // one sub-tile: 4 regions
          auto one = [&](auto sc) {
            constexpr int s = decltype(sc)::value < R ? decltype(sc)::value : 0;
            read_as(af0, 0, s);
            EFG_SB; P0(); P1(); read_as(af1, 1, s); mfma16(af0, Bq[PC][0], s);
            EFG_SB; P2(); P3(); read_as(af0, 2, s); mfma16(af1, Bq[PC][1], s);
            EFG_SB; P4(); P5(); read_as(af1, 3, s); mfma16(af0, Bq[PC][2], s);
            EFG_SB; mfma16(af1, Bq[PC][3], s);
            EFG_SB;
          };
          if (a0) one(I0{}); else one(I1{});
        } else {   // a dead step (a unit without offsets): the loads only
          P0(); P1(); P2(); P3(); P4(); P5();
          EFG_SB;
        }
      };
      if (!EFG_FAT_DEEP || par == 0) step(I0{}); else step(I1{});
      par ^= 1;
      boff_c = boff_x;
#undef EFG_SB
    }
    __builtin_amdgcn_wave_barrier();

    // ---- the compute side moves on ----
    if (--steps_left == 0) {
      finish_unit(uc.prow, uc.ntile, uc.base, uc.n, uc.j0, uc.j1, uc.vid);
      if (!have_next) break;
      uc = ug;   // (the gather side opened this unit during the step above: switched)
      steps_left = uc.steps;
#pragma unroll
      for (int s = 0; s < R; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    (void)switched;
    c_c = c_x;
    ch_c = ch_x;
    stash();
    cm0 = pre_m[0];
    cm1 = pre_m[R - 1];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (dbg && lane == 0) {
    long long* d = dbg + (size_t)g * 16;
    d[8] = c_prol;
    d[9] = c_store;
    d[10] = c_ticket;
    d[11] = c_sum;
    d[12] = c_rows;
    d[13] = c_open;
    d[0] = t_start;
    d[1] = (long long)__builtin_amdgcn_s_memtime();
    d[2] = d_full;
    d[3] = d_half;
    d[4] = d_units;
    d[5] = d_cut;
    d[6] = blockIdx.x;
    d[7] = (long long)walk_dummy_xcd();
  }
}

// scratch + tickets per (device, stream): kernels of one stream are serialised, so they can share them
struct FatState {
  float* scratch = nullptr;
  int* tickets = nullptr;
  size_t tickets_n = 0;
};
constexpr int kFatMaxWaves = 256 * 4 * 2;   // 2 waves per SIMD
void* g_fat_debug = nullptr;                // efg_spconv_fat_debug: [waves][8] int64 the kernel fills (timing study)
std::mutex g_fat_mu;
std::map<std::pair<int, hipStream_t>, FatState> g_fat_pool;

int fat_state_for_stream(hipStream_t stream, size_t tickets_needed, FatState** out) {
  int dev = 0;
  EFG_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_fat_mu);
  auto key = std::make_pair(dev, stream);
  FatState& st = g_fat_pool[key];
  if (!st.scratch) EFG_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st.scratch), (size_t)kFatMaxWaves * 2 * (2 * 4 * 256) * sizeof(float)));
  if (st.tickets_n < tickets_needed) {
    // (the old array may still be in use by queued launches of this stream: synchronise before freeing it)
    if (st.tickets) {
      EFG_HIP_TRY(hipStreamSynchronize(stream));
      EFG_HIP_TRY(hipFree(st.tickets));
    }
    size_t cap = 1 << 16;
    while (cap < tickets_needed) cap <<= 1;
    EFG_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st.tickets), cap * sizeof(int)));
    EFG_HIP_TRY(hipMemset(st.tickets, 0, cap * sizeof(int)));
    st.tickets_n = cap;
  }
  *out = &st;
  return EFG_OK;
}

// Shape: R = 2 sub-tiles per unit (weights loaded once for 32 rows) by default; EFG_FAT_R forces it (sweeps).
// EFG_FAT_WAVES (1 | 2): waves per SIMD of the launch.
int run_fat(TileArgs a, const PlanView& pv, hipStream_t stream) {
  static const int r_env = getenv("EFG_FAT_R") ? atoi(getenv("EFG_FAT_R")) : 2;
  static const int waves_env = getenv("EFG_FAT_WAVES") ? atoi(getenv("EFG_FAT_WAVES")) : 2;
  const int r = r_env >= 2 ? 2 : 1;
  a.ux = (unsigned)(a.n_tiles / r);
  a.uy = (unsigned)((a.np / 16 + 3) / 4);
  a.sk_prefix = r == 2 ? pv.pfx2 : pv.pfx1;
  FatState* st = nullptr;
  if (int rc = fat_state_for_stream(stream, (size_t)a.ux * a.uy, &st)) return rc;
  a.sk_scratch = st->scratch;
  a.sk_flags = st->tickets;
  a.sk_fallbacks = reinterpret_cast<int*>(g_fat_debug);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  const int wgs = std::min(cus * (waves_env >= 2 ? 2 : 1), kFatMaxWaves / 4);
  if (r == 2) hipLaunchKernelGGL((conv_fat_kernel<4, 2>), dim3(wgs), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_fat_kernel<4, 1>), dim3(wgs), dim3(256), 0, stream, a);
  return EFG_OK;
}

}  // namespace
}  // namespace efg
