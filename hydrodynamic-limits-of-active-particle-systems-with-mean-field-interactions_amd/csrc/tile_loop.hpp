// tile_loop.hpp -- MANY synchronous steps in ONE launch: tile_step's step with the tiles kept resident (included by
// aps_hip.hip after tile_step.hpp).
//
// Hot path replaced: the loop of ParticleSystem.run (PARTICLE_solver_CLASS.py:511-516), n iterations of
// compute_local_m_field (:216-246) + step_gillespie (:254-448) in the fixed-dt synchronous scheme of DESIGN.md 3.
//
// When the whole grid of tiles is resident at once (BASELINE config 2: 633 workgroups on 256 CUs) a step of tile_step is a
// ~14 us latency chain of which most is not the step: the launch, 37 KB of weight table staged into LDS again, the
// tile's cells and {W, S} read back, the results written out, the drain.  Here a workgroup keeps its tile for the whole
// call: table, cells and the frame's {W, S} stay in LDS (the field of the two halo sites either side is kept exactly too:
// every deposit in reach is added to them as well), and between two steps a tile waits only for what it needs of its
// neighbours -- the deposit lists of the tiles within the table's reach and three sites of cells from the tile on
// either side.  Those travel through global memory as 8-byte {tag, word} granules (one write-through store each, polled
// with L1-bypassing loads until the tag is this step's: cdna_hip_programming.md Guideline 16, form R2); records are
// double buffered by step parity, which is enough because a tile cannot run more than one step ahead of a tile it
// exchanges with.  No grid barrier, no flag, no fence.
//
// Safety: every wait is bounded (s_memrealtime); a workgroup that gives up raises a flag that every other wait watches,
// all workgroups leave, and the host repeats the call with one launch per step -- the loop writes the final state only
// into the buffer set of the other parity, so the inputs are still intact (after an even number of steps the host lets the
// two sets trade places).  Results are the same bits either way: the arithmetic and the random numbers are tile_step's.
#pragma once

constexpr int TL_SEG_MIN = 88;             // the four pooled deposit lists hold 4 (seg + 4) entries each; 88 at least (three workgroups per CU at config 2's 37 KB
                                           // table: <= 42 LDS granules of 1280 B), the host takes up to 256 when the residency the grid needs leaves room
constexpr int TL_NG = 2;                   // bucket groups whose first page a wave asks for ahead of time (config 2 has two)
#ifndef TL_MIN_WAVES
#define TL_MIN_WAVES 2                     /* waves per SIMD the register allocator must leave room for (tuning builds may ask for more) */
#endif
constexpr int TL_ABORT = 31;               // misc word: this workgroup leaves (a wait ran out, here or elsewhere)

struct LoopArgs {
    TileArgs a;                            // buffers of the first step's parity: *_in = [par] (read once), *_out = [par ^ 1] (final state)
    unsigned long long step0;              // index of the first step
    int nsteps;                            // any; the final state goes to the *_out buffers (the host swaps the two sets after an even count)
    uint32_t tag0;                         // the records written in iteration s carry the tag tag0 + s + 1 (never 0, never reused)
    int rec, drec;                         // granules per record; deposit slots of a record (a multiple of 16 above dcap)
    int seg;                               // sizes the pooled deposit lists: 4 (seg + 4) entries per class (a multiple of 4)
    unsigned long long *xrec;              // [2][E][ntile][rec] {tag << 32 | word}
    unsigned *abort_dev, *abort_host;      // raised by a workgroup whose wait ran out
    unsigned long long timeout_ticks;      // of the 100 MHz clock
#ifdef APS_LOOP_DEBUG
    uint32_t *dbg;                         // [nsteps][3][L]: cells after the step, occupancy | proposal << 8 seen by the step, particles on frame
#endif
};

struct TlLds { size_t seg, cells, cells2, props, occ, misc, plist, fw, fs, tab, total; };
__host__ __device__ inline TlLds tl_lds_layout(int tlen, int RS, int own, int K, int wbytes, int seg) {
    TlLds l;
    const size_t TS = 64 * (size_t)RS, ncell = (TS + 2) * K;
    auto up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    l.seg = 0;
    l.cells = l.seg + (size_t)FU_WAVES * 4 * (seg + 4) * sizeof(uint32_t);
    l.cells2 = up(l.cells + ncell * 4, 8);
    l.props = up(l.cells2 + (K == 1 ? 0 : ncell * 4), 8);
    l.occ = up(l.props + TS * K, 8);
    l.misc = up(l.occ + TS + 2, 8);
    l.plist = l.misc + 128;
    l.fw = up(l.plist + TS * (size_t)K * 8, 16);
    l.fs = l.fw + TS * wbytes;
    l.tab = up(l.fs + TS * wbytes, 16);
    l.total = l.tab + (size_t)ts_table_chunks(tlen, RS, own, wbytes) * 1024;
    return l;
}

typedef __attribute__((address_space(1))) unsigned long long tl_gu64;
typedef __attribute__((address_space(1))) unsigned tl_gu32;

__device__ __forceinline__ void tl_store_granule(unsigned long long *p, uint32_t tag, uint32_t word) {
    __hip_atomic_store((tl_gu64 *)p, ((unsigned long long)tag << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ONE write-through store
}
__device__ __forceinline__ unsigned long long tl_load_granule(const unsigned long long *p) {
    return __hip_atomic_load((tl_gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                      // bypasses L1
}

__device__ __forceinline__ void tl_lds_add(double *p, double v) { unsafeAtomicAdd(p, v); }       // ds_add_f64: sums are exact, the order is free
__device__ __forceinline__ void tl_lds_add(int *p, int v) { atomicAdd(p, v); }

template <int BC, int RS, bool K1, bool F32>
// (two waves per SIMD for every instance: the loop runs with exactly two workgroups per CU where the geometry is chosen for it, and at three the
// smaller frames spilled up to 200 bytes per lane to scratch)
__global__ __launch_bounds__(FU_THREADS, TL_MIN_WAVES) void tile_loop(const LoopArgs la, const void *__restrict__ table_v) {
    using W = typename TsField<F32>::w_t;
    using WS = typename TsField<F32>::ws_t;
    constexpr int SH = TsField<F32>::SH, WB = (int)sizeof(W);
    constexpr bool TAB_LDS = true;
    const TileArgs &a = la.a;
    const W *__restrict__ table_g = reinterpret_cast<const W *>(table_v);
    constexpr int TS = 64 * RS, NOLD = (TS + FU_THREADS - 1) / FU_THREADS;
    constexpr int NSLOT = 16, GB = FU_WAVES * 4;
    const int SEG = la.seg;
    extern __shared__ double lds[];
    const int L = a.L, K = K1 ? 1 : a.K, OWN = a.own;
    const TlLds lay = tl_lds_layout(a.tlen, RS, OWN, K, WB, la.seg);
    char *lds_c = reinterpret_cast<char *>(lds);
    uint32_t *seg_all = reinterpret_cast<uint32_t *>(lds_c + lay.seg);          // the pooled deposit lists: [4 classes][4 (seg + 4) entries]
    uint32_t *cellL = reinterpret_cast<uint32_t *>(lds_c + lay.cells);       // [(TS + 2) K]: frame positions -1 .. TS
    uint32_t *cellN = reinterpret_cast<uint32_t *>(lds_c + lay.cells2);      // K > 1: the cells after this step
    uint8_t *propL = reinterpret_cast<uint8_t *>(lds_c + lay.props);         // [TS K]
    uint8_t *occL = reinterpret_cast<uint8_t *>(lds_c + lay.occ);            // [TS + 2]
    int *misc = reinterpret_cast<int *>(lds_c + lay.misc);                   // by iteration parity: 0/1 deposits of this tile, 2/3 particles of the owned sites, 4/5 of the halo sites
    uint2 *plist = reinterpret_cast<uint2 *>(lds_c + lay.plist);
    W *fieldW = reinterpret_cast<W *>(lds_c + lay.fw), *fieldS = reinterpret_cast<W *>(lds_c + lay.fs);   // [TS] each: the frame's field, kept across the steps
    W *tab = reinterpret_cast<W *>(lds_c + lay.tab);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), e = blockIdx.y;
    const int tile = (int)blockIdx.x;
    const int own0 = tile * OWN, own_n = min(OWN, L - own0), nfr = own_n + 4;
    const int x0 = own0 - 2;
    // the frame clipped to the lattice AND to its valid positions (own_n + 4 sites: the field beyond them is never used).  With
    // the valid frame the "deposits of tile B reach tile A" relation is symmetric, (|A - B| - 1) OWN <= Rt + 2, which is what
    // the two record buffers rely on: two tiles that exchange are never more than one step apart
    const int x0c = max(x0, 0), x1c = min(x0 + nfr - 1, L - 1);
    const int Rt = a.tlen - 1;
    if (__hip_atomic_load((tl_gu32 *)la.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {        // the call was given up before this workgroup started
        if (t == 0) __hip_atomic_store(la.abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // test hook (APS_LOOP_TEST_STALL, abort_dev[1] = (tile + 1) << 16 | iteration, 0 = none): that tile takes only that many
    // iterations and leaves without its record -- what a workgroup that never became resident looks like to its neighbours
    int n_iter = la.nsteps;
    {
        const unsigned stall = la.abort_dev[1];
        if (stall && (int)(stall >> 16) - 1 == tile && e == 0) n_iter = min(n_iter, (int)(stall & 0xFFFFu));
    }
    const uint32_t *__restrict__ cell_e = a.cell_in + (size_t)e * L * K;
    const WS *__restrict__ ws_e = reinterpret_cast<const WS *>(a.ws_in) + (size_t)e * L;
    const uint32_t *__restrict__ dcnt_e = a.dcnt_in + (size_t)e * a.ntile;
    const uint32_t *__restrict__ dep_e = a.dep_in + (size_t)e * a.ntile * a.dcap;
    uint32_t tbase = 0;
    {
        typedef __attribute__((address_space(3))) W lds_w;
        tbase = (uint32_t)(size_t)(lds_w *)tab;
    }
    if (t < 32) misc[t] = 0;
    {   // the table, once per call (LDS-direct loads; the global copy is followed by zeros: the padded tail comes along)
        const int nchunk = ts_table_chunks(a.tlen, RS, OWN, WB);
        const char *srct = reinterpret_cast<const char *>(table_g) + lane * 16;
        for (int c = wave; c < nchunk; c += FU_WAVES) {
            const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tbase + (uint32_t)c * 1024u));
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(srct + c * 1024) : "memory");
        }
    }
    const double beta = a.beta[e];
    auto frame_site = [&](int i) -> int {
        if (i < -1 || i > nfr) return -1;
        int s = x0 + i;
        if (BC == 1) { s %= L; if (s < 0) s += L; return s; }
        return (s < 0 || s >= L) ? -1 : s;
    };
    const int ncell = (TS + 2) * K;
    for (int c = t; c < ncell; c += FU_THREADS) {              // the frame's cells (+1 site either side) as the call finds them
        const int s = frame_site(c / K - 1);
        const uint32_t v = cell_e[(unsigned)max(s, 0) * (unsigned)K + (unsigned)(c % K)];
        cellL[c] = s >= 0 ? v : CELL_EMPTY;
        if (!K1) cellN[c] = CELL_EMPTY;
    }
#pragma unroll
    for (int r = 0; r < NOLD; ++r) {                           // and its field
        const int xi = r * FU_THREADS + t;
        if (xi < TS) {
            const int s = xi < nfr ? frame_site(xi) : -1;
            WS o = ws_e[(unsigned)max(s, 0)];
            if (s < 0) { o.x = 0; o.y = 0; }
            fieldW[xi] = o.x; fieldS[xi] = o.y;
        }
    }
    for (int i = t; i < (TS * K + 3) / 4; i += FU_THREADS) reinterpret_cast<uint32_t *>(propL)[i] = 0u;
    // buckets (= tiles) whose deposits can reach the frame: one run of nbk buckets from b0 that may wrap around the torus
    int b0 = 0, nbk = 0;
    if (BC == 0) {
        b0 = max(0, x0c - Rt - 1) / OWN;
        nbk = min(L - 1, x1c + Rt + 1) / OWN - b0 + 1;
    } else {
        const int lo = x0 - Rt - 1, hi = x0 + nfr - 1 + Rt + 1;
        if (hi - lo + 1 >= L) { b0 = 0; nbk = a.ntile; }
        else {
            const int lom = ((lo % L) + L) % L, him = ((hi % L) + L) % L;
            const int blo = lom / OWN, bhi = him / OWN;
            b0 = blo;
            nbk = (lom <= him) ? bhi - blo + 1 : (a.ntile - blo) + bhi + 1;
            if (nbk > a.ntile) { b0 = 0; nbk = a.ntile; }
        }
    }
    const bool wall = BC == 0 && ((x0c + 1 <= Rt) || (L - x1c <= Rt));
    const bool mirror_ok = 2 * Rt + TS + OWN + 4 < L;
    const int sub = lane >> 4, slot = lane & (NSLOT - 1);
    const uint32_t tlen8 = (uint32_t)a.tlen << SH, L8 = (uint32_t)L << SH;
    uint32_t x8[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        int s = x0 + r * 64 + lane;
        if (BC == 1) { s %= L; if (s < 0) s += L; } else s = min(max(s, 0), L - 1);
        x8[r] = ((uint32_t)s + TS_BIAS) << SH;
    }
    // neighbours whose boundary cells this tile reads (-1: a wall)
    const int nb_l = tile > 0 ? tile - 1 : (BC == 1 ? a.ntile - 1 : -1), nb_r = tile + 1 < a.ntile ? tile + 1 : (BC == 1 ? 0 : -1);
    const size_t rec_e = (size_t)e * a.ntile;                  // records of this ensemble
    const int null_site = x0c;
    const uint32_t tb = tbase;
    uint32_t *dep_o = a.dep_out + ((size_t)e * a.ntile + tile) * a.dcap;
    uint32_t *cell_o = a.cell_out + (size_t)e * L * K;
    const int ngroups = (nbk + GB - 1) / GB;
    const int npass = (wall && mirror_ok) ? 2 : 1;
    // the frame's particles {pos | k << 16, cell} pooled into one list, and the occupancy of the frame sites: once from
    // the staged cells here, afterwards kept up to date by the hand-over of every iteration
    auto append = [&](bool occ, int pos, int k, uint32_t cw, int *counter) {
        const unsigned long long mm = __ballot(occ);
        const int cnt_u = __popcll(mm);
        int base = 0;
        if (lane == 0 && cnt_u) base = atomicAdd(counter, cnt_u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (occ) plist[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u))] = make_uint2((uint32_t)pos | ((uint32_t)k << 16), cw);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's table chunks have landed
    __syncthreads();                                           // cells staged
    for (int i = t; i < TS + 2; i += FU_THREADS) {
        int n = 0;
        for (int k = 0; k < K; ++k) n += cellL[i * K + k] != CELL_EMPTY;
        occL[i] = (uint8_t)n;
    }
    for (int c0 = 0; c0 < ncell; c0 += FU_THREADS) {           // uniform trip count
        const int c = c0 + t;
        const int pos = K1 ? c - 1 : c / K - 1, k = K1 ? 0 : c - (pos + 1) * K;
        const uint32_t cw = c < ncell ? cellL[c] : CELL_EMPTY;
        append(cw != CELL_EMPTY && pos >= 0 && pos < nfr, pos, k, cw, misc + 2);
    }
    __syncthreads();                                           // F of "iteration -1"

#ifdef APS_LOOP_STAMPS
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t0;
#define TLSTAMP(k) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t1_ - st_t0; st_t0 = t1_; }
#else
#define TLSTAMP(k)
#endif
    // what this wave asked for at the end of the previous iteration (in flight across barrier F and the random numbers):
    // page 0 of its buckets of the first TL_NG groups, and (last wave) the neighbours' boundary cells
    unsigned long long xg[TL_NG], xh = 0;
#pragma unroll
    for (int g = 0; g < TL_NG; ++g) xg[g] = 0;
    const int h_side = lane >> 5, h_i = lane & 31;             // lanes 0..3K-1: left neighbour's LAST three sites; 32..: right neighbour's FIRST three
    const int h_nb = h_side ? nb_r : nb_l;
    const bool h_act = wave == FU_WAVES - 1 && h_i < 3 * K && h_nb >= 0;
    auto bucket_of = [&](int j, bool &ok) -> int {
        const int bi = j * GB + sub * FU_WAVES + wave;
        ok = j < ngroups && bi < nbk;
        int b = b0 + (ok ? bi : 0);
        if (b >= a.ntile) b -= a.ntile;
        return b;
    };
    for (int it = 0; it < n_iter; ++it) {
        const unsigned long long step = la.step0 + (unsigned long long)it;
        const uint32_t tag_in = la.tag0 + (uint32_t)it, tag_out = tag_in + 1u;
        const bool first = it == 0, last = it + 1 == la.nsteps;
        const unsigned long long *rec_in = la.xrec + ((size_t)((it + 1) & 1) * a.E * a.ntile + rec_e) * la.rec;
        unsigned long long *rec_out = la.xrec + ((size_t)(it & 1) * a.E * a.ntile + rec_e + tile) * la.rec;
        int *dcount = misc + (it & 1), *pcount = misc + 2 + (it & 1);
        bool gave_up = false;
        // ------------------------------------------------------------ A  this thread's particle and its random numbers (while the records travel)
        const int n0 = *pcount;                                // the owned sites' particles (first iteration: the whole frame's)
        uint2 mine = make_uint2(0u, CELL_EMPTY);
        uint32_t rx[4] = {0u, 0u, 0u, 0u};
        if (t < n0) {
            mine = plist[t];
            philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), mine.y & CELL_ID, (uint32_t)(a.ens_base + e), a.seed_lo, a.seed_hi, rx);
        }
        TLSTAMP(2)
        // ------------------------------------------------------------ B  bounded waits: re-read until the granule carries this step's tag
        auto spin_check = [&](unsigned &spins, unsigned long long &t_w) -> bool {     // true: give up
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) != 0u) return false;
            const bool other = __hip_atomic_load((tl_gu32 *)la.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (!t_w) t_w = now;
            if (!other && now - t_w <= la.timeout_ticks) return false;
            if (!other && lane == 0) {
                __hip_atomic_store((tl_gu32 *)la.abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(la.abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            gave_up = true;
            return true;
        };
        auto wait_granule = [&](const unsigned long long *p, bool active) -> uint32_t {
            unsigned long long x = 0, t_w = 0;
            for (unsigned spins = 0;;) {
                if (active) x = tl_load_granule(p);
                if (!__ballot(active && (uint32_t)(x >> 32) != tag_in)) break;
                if (spin_check(spins, t_w)) break;
            }
            return (uint32_t)x;
        };
        if (!first) {                                          // everything asked for ahead: one round trip when the neighbours are done
            unsigned long long t_w = 0;
            for (unsigned spins = 0;;) {
                bool bad = false;
#pragma unroll
                for (int g = 0; g < TL_NG; ++g) {
                    bool ok;
                    const int b = bucket_of(g, ok);
                    if (ok && (uint32_t)(xg[g] >> 32) != tag_in) {
                        xg[g] = tl_load_granule(rec_in + (size_t)b * la.rec + slot);
                        bad |= (uint32_t)(xg[g] >> 32) != tag_in;
                    }
                }
                if (h_act && (uint32_t)(xh >> 32) != tag_in) {
                    xh = tl_load_granule(rec_in + (size_t)h_nb * la.rec + la.drec + (h_side ? 0 : 3 * K) + h_i);
                    bad |= (uint32_t)(xh >> 32) != tag_in;
                }
                if (!__ballot(bad)) break;
                if (spin_check(spins, t_w)) break;
            }
            // the neighbours' boundary cells: frame positions -1, 0, 1 and own_n + 2 .. own_n + 4 -- occupancy, and the
            // particles of the four halo sites join the list
            if (wave == FU_WAVES - 1 && !gave_up) {
                const uint32_t v = h_act ? (uint32_t)xh : CELL_EMPTY;
                const int hs = h_i / K, hk = h_i - hs * K, pos = (h_side ? own_n + 2 : -1) + hs;
                if (h_act) cellL[(h_side ? (own_n + 3) * K : 0) + h_i] = v;
                // occupancy of the six sites: K consecutive lanes per site
                const unsigned long long mo = __ballot(h_act && v != CELL_EMPTY);
                if (h_i < 3 * K && hk == 0 && h_nb >= 0) occL[pos + 1] = (uint8_t)__popcll((mo >> (h_side * 32 + hs * K)) & ((1ull << K) - 1ull));
                // (behind the owned sites' n0 entries; a counter of its own: n0 is still being read by slower waves)
                const bool hp = h_act && v != CELL_EMPTY && pos >= 0 && pos < nfr;
                const unsigned long long mh = __ballot(hp);
                if (hp) plist[n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u))] = make_uint2((uint32_t)pos | ((uint32_t)hk << 16), v);
                if (lane == 0) misc[4 + (it & 1)] = __popcll(mh);
            }
        }
        TLSTAMP(0)
        // ------------------------------------------------------------ 1  deposits of the previous step -> W, S of the frame
        W accP[RS], accM[RS], accF[RS], accWi[RS], accSi[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) accP[r] = accM[r] = accF[r] = accWi[r] = accSi[r] = 0;
        // The four waves POOL the deposits of their buckets into one list per class (P, M, F; I: image terms of a small box) --
        // one packed LDS atomic per wave and round hands out the slots -- and after a barrier take whole groups of four from the
        // lists in turn: every wave sweeps the same number of groups whatever its buckets held (a wave's own ~7 buckets vary
        // by +-16 %), and a list is padded once, not once per wave.  An entry that finds its list full is swept at once by
        // the wave that holds it.
        uint32_t *shl = seg_all;                               // [4][CAP]
        const int CAP = (FU_WAVES * 4 * (SEG + 4) / 4) & ~3;
        unsigned long long *shcnt = reinterpret_cast<unsigned long long *>(misc + 8 + 2 * (it & 1));   // packed list lengths: 16 bits each
        auto sweep_now = [&](unsigned long long m, const uint32_t word, const int cls) {   // overflow: one by one (never in practice)
            while (m) {
                const int src_lane = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)word, src_lane);
                if (BC == 0) {
                    if (cls == 0) ts_one_abs<RS, 0, F32>(e1, x8, tb, accP);
                    else if (cls == 1) ts_one_abs<RS, 0, F32>(e1, x8, tb, accM);
                    else if (cls == 2) ts_one_abs<RS, 1, F32>(e1, x8, tb, accF);
                    else ts_image_group<TAB_LDS, RS, F32>(make_uint4(e1, DEP_NULL | ((uint32_t)x0c + TS_BIAS), DEP_NULL | ((uint32_t)x0c + TS_BIAS), DEP_NULL | ((uint32_t)x0c + TS_BIAS)),
                                                          x8, tb, table_g, tlen8, L8, accWi, accSi);
                } else {
                    const uint4 q = make_uint4(e1, DEP_NULL | ((uint32_t)null_site + TS_BIAS), DEP_NULL | ((uint32_t)null_site + TS_BIAS), DEP_NULL | ((uint32_t)null_site + TS_BIAS));
                    if (cls == 0) ts_group<1, TAB_LDS, RS, 0, F32>(q, x8, tb, table_g, tlen8, L8, accP);
                    else if (cls == 1) ts_group<1, TAB_LDS, RS, 0, F32>(q, x8, tb, table_g, tlen8, L8, accM);
                    else ts_group<1, TAB_LDS, RS, 1, F32>(q, x8, tb, table_g, tlen8, L8, accF);
                }
            }
        };
#pragma unroll 1
        for (int j = 0; j < ngroups && !gave_up; ++j) {
            bool ok;
            const int b = bucket_of(j, ok);
            uint32_t cnt = 0u;
            if (first) { cnt = min(dcnt_e[(unsigned)b], (uint32_t)a.dcap); if (!ok) cnt = 0u; }   // the lists the previous launch left in the plain arrays
            const unsigned long long *rb = rec_in + (size_t)b * la.rec;                             // else: the records of the previous iteration
            bool active = ok;
#pragma unroll 1
            for (int page = 0;; ++page) {                      // a page of 16 slots per bucket at a time
                uint32_t en = DEP_NULL;
                bool valid = false, more = false;
                if (first) {
                    const uint32_t k0 = (uint32_t)page * NSLOT;
                    en = dep_e[(unsigned)b * (unsigned)a.dcap + min(k0 + slot, (uint32_t)a.dcap - 1u)];
                    valid = k0 + slot < cnt; more = k0 + NSLOT < cnt;
                } else if (__ballot(active)) {
                    if (page == 0 && j < TL_NG) {              // arrived above
                        unsigned long long x = xg[0];
#pragma unroll
                        for (int g = 1; g < TL_NG; ++g) if (j == g) x = xg[g];
                        en = (uint32_t)x;
                    } else {
                        en = wait_granule(rb + min(page * NSLOT + slot, la.drec - 1), active);
                        if (gave_up) break;
                    }
                    valid = active && en != DEP_NULL;
                    // a bucket has another page iff the last slot of this one is taken
                    const unsigned long long mv = __ballot(valid);
                    more = active && ((mv >> (sub * 16 + 15)) & 1ull) && (page + 1) * NSLOT < la.drec;
                }
                const uint32_t en_ = en + TS_BIAS;
                if (!wall) {                                   // (uniform) no image of any deposit reaches this frame: three classes, nothing else
                    const int cw = (int)((en_ >> 27) & 3u) - 1, cs = (int)(en_ >> 29) - 2;
                    const bool cP = valid && cw != 0 && cw == cs, cM = valid && cw != 0 && cw != cs, cF = valid && cw == 0;
                    const unsigned long long kP = __ballot(cP), kM = __ballot(cM), kF = __ballot(cF);
                    const unsigned long long add = (unsigned long long)__popcll(kP) | ((unsigned long long)__popcll(kM) << 16) | ((unsigned long long)__popcll(kF) << 32);
                    if (add) {
                        unsigned long long base = 0;
                        if (lane == 0) base = atomicAdd(shcnt, add);
                        const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
                        // (one uniform test per round whether any list runs over: only then the per-entry checks)
                        const bool room = (int)(blo & 0xFFFFu) + __popcll(kP) <= CAP && (int)(blo >> 16) + __popcll(kM) <= CAP && (int)(bhi & 0xFFFFu) + __popcll(kF) <= CAP;
#define TL_POOL(MASK, COND, WORD, CLS, BASE) { \
                        const int i_ = (int)(BASE) + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)((MASK) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(MASK), 0u)); \
                        if ((COND) && (room || i_ < CAP)) shl[(CLS) * CAP + i_] = (WORD); \
                        if (!room) { const unsigned long long o_ = __ballot((COND) && i_ >= CAP); if (o_) sweep_now(o_, (WORD), (CLS)); } }
                        TL_POOL(kP, cP, en_, 0, blo & 0xFFFFu) TL_POOL(kM, cM, en_, 1, blo >> 16) TL_POOL(kF, cF, en_, 2, bhi & 0xFFFFu)
#undef TL_POOL
                    }
                    active = more;
                    if (!__ballot(active)) break;
                    continue;
                }
                // by class; near a reflecting wall the image of a deposit is the same deposit at the mirrored site (small boxes: image list)
                const int dp = (int)(en & POS_MASK), cw = (int)((en_ >> 27) & 3u) - 1, cs = (int)(en_ >> 29) - 2;
                const bool img_l = valid && wall && (x0c + dp + 1 <= Rt), img_r = valid && wall && (2 * L - 1 - x1c - dp <= Rt);
                const bool cP = cw != 0 && cw == cs, cM = cw != 0 && cw != cs, cF = cw == 0;
                const bool im = img_l || img_r;
                const bool pl = valid && (mirror_ok || !im), mi = im && mirror_ok, ii = valid && im && !mirror_ok;
                const uint32_t mir = (en_ & ~POS_MASK) | (uint32_t)((int)TS_BIAS + (img_l ? -1 - dp : 2 * L - 1 - dp));
                const unsigned long long kP = __ballot(pl && cP), kM = __ballot(pl && cM), kF = __ballot(pl && cF);
                const unsigned long long gP = __ballot(mi && cP), gM = __ballot(mi && cM), gF = __ballot(mi && cF), kI = __ballot(ii);
                const unsigned long long add = (unsigned long long)(__popcll(kP) + __popcll(gP)) | ((unsigned long long)(__popcll(kM) + __popcll(gM)) << 16) |
                                               ((unsigned long long)(__popcll(kF) + __popcll(gF)) << 32) | ((unsigned long long)__popcll(kI) << 48);
                if (add) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(shcnt, add);
                    const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
#define TL_POOL(MASK, COND, WORD, CLS, BASE) { \
                    const int i_ = (int)(BASE) + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)((MASK) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(MASK), 0u)); \
                    if ((COND) && i_ < CAP) shl[(CLS) * CAP + i_] = (WORD); \
                    const unsigned long long o_ = __ballot((COND) && i_ >= CAP); \
                    if (o_) sweep_now(o_, (WORD), (CLS)); }
                    TL_POOL(kP, (pl && cP), en_, 0, blo & 0xFFFFu) TL_POOL(gP, (mi && cP), mir, 0, (blo & 0xFFFFu) + __popcll(kP))
                    TL_POOL(kM, (pl && cM), en_, 1, blo >> 16) TL_POOL(gM, (mi && cM), mir, 1, (blo >> 16) + __popcll(kM))
                    TL_POOL(kF, (pl && cF), en_, 2, bhi & 0xFFFFu) TL_POOL(gF, (mi && cF), mir, 2, (bhi & 0xFFFFu) + __popcll(kF))
                    TL_POOL(kI, ii, en_, 3, bhi >> 16)
#undef TL_POOL
                }
                active = more;
                if (!__ballot(active)) break;
            }
        }
        if (gave_up && lane == 0) misc[TL_ABORT] = 1;
        TLSTAMP(8)
        __syncthreads();                                       // S: the lists are complete
        TLSTAMP(9)
        if (misc[TL_ABORT]) return;                            // uniform: some wait ran out (here or in another workgroup)
        {
            const unsigned long long tot = *shcnt;
            const int nP = min((int)(tot & 0xFFFFu), CAP), nM = min((int)((tot >> 16) & 0xFFFFu), CAP), nF = min((int)((tot >> 32) & 0xFFFFu), CAP),
                      nI = min((int)(tot >> 48), CAP);
            const uint32_t nullw = DEP_NULL | ((uint32_t)null_site + TS_BIAS);
            int rot = wave;                                    // whole groups of four are dealt round-robin across the lists
#define TL_SHARED(CLS, N, CALL4) { \
            const int ng_ = ((N) + 3) >> 2; \
            const uint4 *l4 = reinterpret_cast<const uint4 *>(shl + (CLS) * CAP); \
            int g = rot & 3; \
            uint4 q = l4[min(g, CAP / 4 - 1)]; \
            _Pragma("unroll 1") for (; g < ng_; g += FU_WAVES) { \
                const uint4 qn = l4[min(g + FU_WAVES, CAP / 4 - 1)]; \
                if (4 * g + 3 >= (N)) {                        /* the list's last group: null deposits behind its end */ \
                    if (4 * g + 1 >= (N)) q.y = nullw; \
                    if (4 * g + 2 >= (N)) q.z = nullw; \
                    q.w = nullw; } \
                CALL4; \
                q = qn; } \
            rot = (rot - ng_) & 3; }
            if (BC == 1) {
                TL_SHARED(0, nP, (ts_group<1, TAB_LDS, RS, 0, F32>(q, x8, tb, table_g, tlen8, L8, accP)))
                TL_SHARED(1, nM, (ts_group<1, TAB_LDS, RS, 0, F32>(q, x8, tb, table_g, tlen8, L8, accM)))
                TL_SHARED(2, nF, (ts_group<1, TAB_LDS, RS, 1, F32>(q, x8, tb, table_g, tlen8, L8, accF)))
            } else {
                TL_SHARED(0, nP, (ts_group<0, TAB_LDS, RS, 0, F32>(q, x8, tb, table_g, tlen8, L8, accP)))
                TL_SHARED(1, nM, (ts_group<0, TAB_LDS, RS, 0, F32>(q, x8, tb, table_g, tlen8, L8, accM)))
                TL_SHARED(2, nF, (ts_group<0, TAB_LDS, RS, 1, F32>(q, x8, tb, table_g, tlen8, L8, accF)))
            }
#undef TL_SHARED
            if (nI) {
                const uint32_t nulli = DEP_NULL | ((uint32_t)x0c + TS_BIAS);
                const uint4 *l4 = reinterpret_cast<const uint4 *>(shl + 3 * CAP);
#pragma unroll 1
                for (int g = rot & 3; g < (nI + 3) >> 2; g += FU_WAVES) {
                    uint4 q = l4[g];
                    if (4 * g + 1 >= nI) q.y = nulli;
                    if (4 * g + 2 >= nI) q.z = nulli;
                    if (4 * g + 3 >= nI) q.w = nulli;
                    ts_image_group<TAB_LDS, RS, F32>(q, x8, tb, table_g, tlen8, L8, accWi, accSi);
                }
            }
        }
        TLSTAMP(10)
#pragma unroll
        for (int r = 0; r < RS; ++r) {                         // W = P + M, S = P - M + F (+ the image deposits), exact on the weight grid
            const W dw = (accP[r] + accM[r]) + accWi[r], ds = ((accP[r] - accM[r]) + accF[r]) + accSi[r];
            if (dw != 0) tl_lds_add(&fieldW[r * 64 + lane], dw);
            if (ds != 0) tl_lds_add(&fieldS[r * 64 + lane], ds);
        }
        TLSTAMP(11)
        __syncthreads();                                       // B: field, cells, occupancy and particle list of the frame complete
        TLSTAMP(7)
        // ------------------------------------------------------------ 2  proposals, a lane per particle
        {
            const Model M = *a.model;                          // uniform address: scalar loads, only now
            const int n_part = n0 + misc[4 + (it & 1)];
            auto propose_one = [&](const uint2 pc, const uint32_t (&x)[4]) {
                const int pos = (int)(pc.x & 0xFFFFu), k = (int)(pc.x >> 16);
                const int s = frame_site(pos);
                const bool anch = a.anchor ? a.anchor[s] != 0 : false;
                propL[pos * K + k] = decide_proposal(M, anch, s, (pc.y & CELL_PLUS) ? 1 : -1, (pc.y & CELL_BOUND) != 0,
                                                     clip_field((double)fieldS[pos], (double)fieldW[pos]), beta, occL[pos + 1], occL[pos], occL[pos + 2], x);
            };
            if (t < n_part) {
                if (t >= n0) {                                 // a halo particle that joined after the random numbers were drawn
                    mine = plist[t];
                    philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), mine.y & CELL_ID, (uint32_t)(a.ens_base + e), a.seed_lo, a.seed_hi, rx);
                }
                propose_one(mine, rx);
            }
            for (int j = FU_THREADS + t; j < n_part; j += FU_THREADS) {   // more than 256 particles on the frame
                const uint2 pc = plist[j];
                uint32_t x[4];
                philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), pc.y & CELL_ID, (uint32_t)(a.ens_base + e), a.seed_lo, a.seed_hi, x);
                propose_one(pc, x);
            }
        }
        __syncthreads();                                       // D
        TLSTAMP(3)
        // ------------------------------------------------------------ 3 + 4  exclusion, new cells of the owned sites, deposits
        auto rank_at = [&](int j, uint32_t id) -> int {
            int n = 0;
            for (int k = 0; k < K; ++k) {
                const uint32_t cl_ = cellL[j * K + k], cr_ = cellL[(j + 2) * K + k];
                const int el = propL[(j - 1) * K + k] & 7, er = propL[(j + 1) * K + k] & 7;
                n += (cl_ != CELL_EMPTY && (el == EV_RIGHT || el == EV_FWD) && (cl_ & CELL_ID) < id);
                n += (cr_ != CELL_EMPTY && er == EV_LEFT && (cr_ & CELL_ID) < id);
            }
            return n;
        };
        auto cap_at = [&](int j) -> int { const int c = K - (int)occL[j + 1]; return c < 1 ? 1 : (c > 32 ? 32 : c); };
        auto put_deposit = [&](int kd, uint32_t d) { if (last) dep_o[kd] = d; else tl_store_granule(rec_out + kd, tag_out, d); };
        auto log_exit = [&](const uint32_t c, const int s) {   // a particle leaves the system (ref :307-312, :427-446): exit log, its slot marked dead
            const TileRare R = *a.rare;
            const unsigned kx = atomicAdd(&R.n_exit[e], 1u);
            if ((int)kx < R.exit_cap) {
                double *row = R.exit_log + ((size_t)e * R.exit_cap + kx) * 3;
                row[0] = (double)step; row[1] = (double)s; row[2] = (double)(c & CELL_ID);
            }
            R.src[(size_t)e * R.Npad + R.slot_of[(size_t)e * R.N + (c & CELL_ID)]] =
                (uint32_t)s | DEAD_BIT | ((c & CELL_PLUS) ? SPIN_BIT : 0u) | ((c & CELL_BOUND) ? BOUND_BIT : 0u);
        };
        uint32_t newc[NOLD];                                   // K = 1: the owned sites' cells after this step
        int newn[NOLD];                                        // particles on the site after this step
#pragma unroll
        for (int r = 0; r < NOLD; ++r) {
            newc[r] = CELL_EMPTY; newn[r] = 0;
            const int xi = r * FU_THREADS + t;
            if (xi < 2 || xi >= 2 + own_n || xi >= TS) continue;
            const int s = frame_site(xi);
            if constexpr (K1) {
                // One cell per site: a hop is only ever proposed into a site that was empty at the start of the step
                // (channels(): open_l / open_r), so its capacity is 1 and the only rival is the particle on the far side of the
                // target -- the rule of DESIGN 3 (smaller id wins) on a window of five sites, read once.
                const uint32_t cm2 = cellL[xi - 1], cm1 = cellL[xi], c0 = cellL[xi + 1], cp1 = cellL[xi + 2], cp2 = cellL[xi + 3];
                const int pm2 = propL[xi - 2] & 7, pm1 = propL[xi - 1] & 7, p0 = propL[xi] & 7, pp1 = propL[xi + 1] & 7, pp2 = xi + 2 < TS ? propL[xi + 2] & 7 : 0;
                const bool from_l2 = cm2 != CELL_EMPTY && (pm2 == EV_RIGHT || pm2 == EV_FWD);     // the particle two sites left wants xi - 1
                const bool from_l1 = cm1 != CELL_EMPTY && (pm1 == EV_RIGHT || pm1 == EV_FWD);     // the left neighbour wants this site
                const bool from_r1 = cp1 != CELL_EMPTY && pp1 == EV_LEFT;                          // the right neighbour wants this site
                const bool from_r2 = cp2 != CELL_EMPTY && pp2 == EV_LEFT;                          // the particle two sites right wants xi + 1
                uint32_t c = c0;
                if (c != CELL_EMPTY) {
                    const int sgn = (c & CELL_PLUS) ? 1 : -1;
                    uint32_t d0 = 0, d1 = 0;
                    int nd = 0;
                    bool stays = true;
                    if (p0 == EV_LEFT || p0 == EV_RIGHT || p0 == EV_FWD) {
                        const bool left = p0 == EV_LEFT;
                        // rivals for the target: the other neighbour of the target site, if it proposes into it with a smaller id
                        const int rivals = left ? (int)(from_l2 && (cm2 & CELL_ID) < (c & CELL_ID)) + (int)(false)
                                                : (int)(from_r2 && (cp2 & CELL_ID) < (c & CELL_ID));
                        if (rivals < 1) {
                            stays = false;
                            d0 = deposit(s, -1, -sgn); d1 = deposit(frame_site(left ? xi - 1 : xi + 1), 1, sgn); nd = 2;
                        }
                    } else if (p0 == EV_BIND) c |= CELL_BOUND;
                    else if (p0 == EV_UNBIND) c &= ~CELL_BOUND;
                    else if (p0 == EV_FLIP) { c ^= CELL_PLUS; d0 = deposit(s, 0, -2 * sgn); nd = 1; }
                    // (EV_EXIT cannot be drawn here: with one cell per site the host takes this path only when no particle can leave --
                    //  the exit code costs this kernel 28 VGPRs and 2 % of config 2)
                    if (stays) { newc[r] = c; newn[r] = 1; }
                    if (nd) {
                        const int kd = atomicAdd(dcount, nd);
                        if (kd + nd <= a.dcap) { put_deposit(kd, d0); if (nd == 2) put_deposit(kd + 1, d1); }
                    }
                }
                // granted arrivals (capacity of this site: max(1, 1 - occupancy) = 1; a rival is the other neighbour with a smaller id)
                if (from_l1 && !(from_r1 && (cp1 & CELL_ID) < (cm1 & CELL_ID))) { newc[r] = cm1; newn[r] += 1; }
                if (from_r1 && !(from_l1 && (cm1 & CELL_ID) < (cp1 & CELL_ID))) { newc[r] = cp1; newn[r] += 1; }
                continue;
            }
            uint32_t *outN = cellN + (xi + 1) * K;
            int n_out = 0;
            auto emit = [&](uint32_t c) { if (K1) newc[r] = c; else outN[n_out] = c; ++n_out; };
            for (int k = 0; k < K; ++k) {
                uint32_t c = cellL[(xi + 1) * K + k];
                if (c == CELL_EMPTY) continue;
                const int ev = propL[xi * K + k] & 7, sgn = (c & CELL_PLUS) ? 1 : -1;
                uint32_t d0 = 0, d1 = 0;
                int nd = 0;
                bool stays = true;
                if (ev == EV_LEFT || ev == EV_RIGHT || ev == EV_FWD) {
                    const int j = ev == EV_LEFT ? xi - 1 : xi + 1;
                    if (rank_at(j, c & CELL_ID) < cap_at(j)) {
                        stays = false;
                        d0 = deposit(s, -1, -sgn); d1 = deposit(frame_site(j), 1, sgn); nd = 2;
                    }
                } else if (ev == EV_BIND) c |= CELL_BOUND;
                else if (ev == EV_UNBIND) c &= ~CELL_BOUND;
                else if (ev == EV_FLIP) { c ^= CELL_PLUS; d0 = deposit(s, 0, -2 * sgn); nd = 1; }
                else if (ev == EV_EXIT) { stays = false; log_exit(c, s); d0 = deposit(s, -1, -sgn); nd = 1; }
                if (stays) emit(c);
                if (nd) {
                    const int kd = atomicAdd(dcount, nd);
                    if (kd + nd <= a.dcap) { put_deposit(kd, d0); if (nd == 2) put_deposit(kd + 1, d1); }
                }
            }
            const int cap = cap_at(xi);
            for (int k = 0; k < K; ++k) {
                const uint32_t cl_ = cellL[xi * K + k], cr_ = cellL[(xi + 2) * K + k];
                const int el = propL[(xi - 1) * K + k] & 7, er = propL[(xi + 1) * K + k] & 7;
                if (cl_ != CELL_EMPTY && (el == EV_RIGHT || el == EV_FWD) && rank_at(xi, cl_ & CELL_ID) < cap) emit(cl_);
                if (cr_ != CELL_EMPTY && er == EV_LEFT && rank_at(xi, cr_ & CELL_ID) < cap) emit(cr_);
            }
            if (!K1) for (int k = n_out; k < K; ++k) outN[k] = CELL_EMPTY;
            newn[r] = n_out;
        }
        __syncthreads();                                       // E: nobody reads this step's cells, proposals, occupancy and list any more
        TLSTAMP(4)
        // ------------------------------------------------------------ 5  hand over: new cells, this tile's record (or the plain arrays),
        //                                                                 next iteration's list and occupancy of the owned sites
        const int count = min(*dcount, a.dcap);
        int *pnext = misc + 2 + ((it + 1) & 1);
#pragma unroll
        for (int r = 0; r < NOLD; ++r) {
            const int xi = r * FU_THREADS + t;
            const bool mine_site = !(xi < 2 || xi >= 2 + own_n || xi >= TS);
            const int io = xi - 2;
            for (int k = 0; k < K; ++k) {                      // (uniform trip count: the list's ballots)
                uint32_t c = CELL_EMPTY;
                if (mine_site) {
                    c = K1 ? newc[r] : cellN[(xi + 1) * K + k];
                    cellL[(xi + 1) * K + k] = c;
                    if (last) cell_o[(unsigned)frame_site(xi) * (unsigned)K + (unsigned)k] = c;
                    else {
                        if (io < 3) tl_store_granule(rec_out + la.drec + io * K + k, tag_out, c);
                        if (io >= own_n - 3) tl_store_granule(rec_out + la.drec + 3 * K + (io - (own_n - 3)) * K + k, tag_out, c);
                    }
                }
                if (!last) append(mine_site && c != CELL_EMPTY, xi, k, c, pnext);
            }
            if (mine_site) {
                occL[xi + 1] = (uint8_t)newn[r];
                if (last) {
                    WS f; f.x = fieldW[xi]; f.y = fieldS[xi];
                    reinterpret_cast<WS *>(a.ws_out)[(size_t)e * L + (unsigned)frame_site(xi)] = f;
                }
            }
        }
        if (!last && t < NSLOT) {                              // the rest of the record's last page: empty slots (at least one)
            const int idx = count + t;
            if (idx < (count / NSLOT + 1) * NSLOT && idx < la.drec) tl_store_granule(rec_out + idx, tag_out, DEP_NULL);
        }
        for (int i = t; i < (TS * K + 3) / 4; i += FU_THREADS) reinterpret_cast<uint32_t *>(propL)[i] = 0u;
        if (t == 0) {
            misc[(it + 1) & 1] = 0;                            // the next iteration's deposit counter (this one's is still being read)
            misc[2 + (it & 1)] = 0; misc[4 + (it & 1)] = 0;    // this iteration's particle counters: used again in two iterations
            misc[8 + 2 * (it & 1)] = 0; misc[9 + 2 * (it & 1)] = 0;   // and its list lengths
            if (last) {
                a.dcnt_out[(size_t)e * a.ntile + tile] = (uint32_t)count;
                if (tile == 0 && e == 0) a.stepw[(a.par + la.nsteps) & 1] = la.step0 + (unsigned long long)la.nsteps;   // (nobody reads the step words in here)
            }
        }
        if (!last) {                                           // ask for the next iteration's records now: in flight across the barrier and the random numbers
            const unsigned long long *rn = la.xrec + ((size_t)(it & 1) * a.E * a.ntile + rec_e) * la.rec;
#pragma unroll
            for (int g = 0; g < TL_NG; ++g) {
                bool ok;
                const int b = bucket_of(g, ok);
                xg[g] = ok ? tl_load_granule(rn + (size_t)b * la.rec + slot) : 0ull;
            }
            xh = h_act ? tl_load_granule(rn + (size_t)h_nb * la.rec + la.drec + (h_side ? 0 : 3 * K) + h_i) : 0ull;
        }
        TLSTAMP(5)
        __syncthreads();                                       // F: list and occupancy of the owned sites complete
    }
#ifdef APS_LOOP_STAMPS
    if (lane == 0 && tile < 512 && e == 0) {                   // [tile][wave][16]
        unsigned long long *o = a.rare->stamps + ((size_t)tile * FU_WAVES + wave) * 16;
        for (int k = 0; k < 12; ++k) o[k] = st_acc[k];
        o[6] = __builtin_amdgcn_s_memtime() - st_begin;
        o[1] = st_acc[8] + st_acc[9] + st_acc[10] + st_acc[11];
    }
#endif
}
