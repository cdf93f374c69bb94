// tile_step.hpp -- the whole synchronous step as ONE kernel over site tiles (included by aps_hip.hip, inside its
// anonymous namespace, after field_update's helpers).
//
// Hot path replaced: the body of ParticleSystem.run's loop (PARTICLE_solver_CLASS.py:511-516), i.e.
// compute_local_m_field (:216-246) + step_gillespie (:254-448) in the fixed-dt synchronous scheme of DESIGN.md 3.
//
// State is SITE-CENTRIC and double buffered by step parity (read [par], write [par ^ 1]):
//   cell [E][L][K] u32   one word per particle slot of a site: id (bits 0-29) | bound << 30 | plus << 31, or CELL_EMPTY
//                        (_build_occupancy, ref :248-252, is the number of non-empty cells of a site)
//   ws   [E][L] double2  {W = tot_conv, S = s_conv} (ref :224-238, unnormalised taps on the weight grid)
//   dep  [E][T][dcap], dcnt [E][T]   field changes ("deposits") made by the particles of tile t in the previous step
// A workgroup owns OWN = 64 RS - 4 consecutive sites and computes on a frame of 64 RS sites (2 halo sites on either
// side).  One step of a tile:
//   1  add every deposit of the previous step that reaches the frame to W, S of the frame sites (field_update's sweep)
//   2  proposals of every particle on the frame (rates -> Philox -> event), from the fresh field and the cells
//   3  exclusion: hops into a site are granted in increasing particle id while capacity lasts (the proposers of a site
//      sit on its two neighbours, so two halo sites decide every hop from or into an owned site)
//   4  new cells of the owned sites (stayers + granted arrivals), the tile's deposit list for the next step, exits
// Neighbouring tiles recompute the same halo proposals from the same inputs (exact arithmetic, counter-based random
// numbers), so no tile ever waits for another: the kernel boundary is the only synchronisation of a step, nothing is
// atomic in global memory except the exit log, and a site range of tiles can live on another GPU (halo = 3 sites of
// cells, 2 sites of ws and the deposit lists within the table's reach).
#pragma once

constexpr uint32_t CELL_EMPTY = 0xFFFFFFFFu, CELL_ID = 0x3FFFFFFFu, CELL_BOUND = 1u << 30, CELL_PLUS = 1u << 31;
constexpr int TS_CREG = 4;                 // rounds of 64 cell words a wave can stage through registers (general K)

// what only the rare paths (exits, diagnostic stamps) touch: kept out of the kernel's argument registers
struct TileRare { double *exit_log; unsigned *n_exit; uint32_t *src; const uint32_t *slot_of; unsigned long long *stamps; long long N; int exit_cap, Npad;
                  int rec_lo, rec_hi; };   // the rank's own tiles: exits in ghost tiles (stepped redundantly) are the owner's to record

struct TileArgs {
    int L, K, tlen, own, ntile, dcap, par, tile_lo, field_only, field_mode, ens_base, E, periodic;
    int dense_rt, dense_m;                             // tile_dense.hpp only: the deposits of the step go into the coefficient signals of the
    int *dense;                                        // convolution (ntt_conv.hpp) [E][2][2^dense_m], index = site + dense_rt, instead of lists
    uint32_t seed_lo, seed_hi;                         // Philox key
    const Model *model;                                // device copy of the rate parameters (read by the proposal phase)
    const TileRare *rare;
    const void *ws_in; void *ws_out;                   // [E][L] double2 {W, S}, or int2 in units of 2^-q (fp32 mode)
    const uint32_t *cell_in; uint32_t *cell_out;       // [E][L][K]
    const uint32_t *dcnt_in, *dep_in;                  // [E][ntile], [E][ntile][dcap]: written by the previous step
    uint32_t *dcnt_out, *dep_out;
    const long long *gpart_in; long long *gpart_out;   // [E][ntile][2] spin sum / live count per tile (global-field mode)
    unsigned long long *stepw;                         // [2] device step words (see propose_lattice)
    const double *beta; const uint8_t *anchor;
};

// slots per bucket and load, buckets per group: TAB_LDS 4 buckets x 16 slots per wave (16 buckets per group);
// table windowed: 2 buckets x 32 slots (8 per group), so that a group's window stays near 4096 + tile entries
template <bool TAB_LDS> struct TsGeom { static constexpr int SB = TAB_LDS ? 4 : 5, NSLOT = 1 << SB, GB = FU_WAVES * (64 >> SB); };

__host__ __device__ inline int ts_table_pad(int RS, int own) { return 64 * RS + own + 2; }
__host__ __device__ inline int ts_table_chunks(int tlen, int RS, int own, int wbytes) { return ((tlen + ts_table_pad(RS, own) + 1) * wbytes + 1023) / 1024; }   // 1 KB each
constexpr int TS_WH = 8;                   // windowed sweep: bucket offsets per group (buckets tile - off and tile + off, off in [jH, jH + H))
__host__ __device__ inline int ts_win_entries(int RS, int own) { return ((TS_WH + 1) * own + 64 * RS + 8 + 127) / 128 * 128; }
constexpr int TS_SHCAP = 320;              // windowed sweep: capacity of a shared class list (a round adds at most 256 entries); per side: half
constexpr int TS_SEG = 128;                // entries per deposit segment of a wave (four segments: P, M, F, image)
struct TsLds { size_t seg, cells, props, occ, misc, plist, tab, field, total; int Q; bool cells_in_regs; };
__host__ __device__ inline TsLds ts_lds_layout(int tlen, bool tab_lds, int RS, int own, int K, int wbytes = 8) {
    TsLds l;
    const size_t TS = 64 * (size_t)RS;
    const int ncell = (int)(TS + 2) * K;
    l.Q = (ncell + FU_WAVES - 1) / FU_WAVES;                     // cells per wave
    l.cells_in_regs = l.Q <= 64 * TS_CREG;
    l.seg = 0;
    l.cells = l.seg + (size_t)FU_WAVES * 4 * (TS_SEG + 4) * sizeof(uint32_t);
    l.props = l.cells + ((TS + 2) * K * 4 + 7) / 8 * 8;
    l.occ = l.props + (TS * K + 7) / 8 * 8;
    l.misc = l.occ + (TS + 2 + 7) / 8 * 8;
    l.plist = l.misc + 128;
    l.tab = (l.plist + (l.cells_in_regs ? TS * (size_t)K * 8 : 0) + 15) / 16 * 16;
    const size_t table = tab_lds ? (size_t)ts_table_chunks(tlen, RS, own, wbytes) * 1024 : (size_t)2 * ts_win_entries(RS, own) * wbytes;
    const size_t red = (size_t)FU_WAVES * TS * 2 * wbytes;
    l.field = l.tab + red;                                      // fresh {W, S} of the frame sites, behind the partial sums
    const size_t after = red + TS * sizeof(double2);
    l.total = l.tab + (table > after ? table : after);
    return l;
}

// Field arithmetic.  F32 = false: binary64 weights on the grid 2^-q (q = 51 - bits of the largest possible sum): the exact
// field every other formulation and the oracle use.  F32 = true (`fp32` flag of aps_params): the same construction on the
// coarser grid that lets every sum fit a 32-bit integer (q = 29 - bits): weights, W and S are int32 in units of 2^-q --
// float32-class accuracy (relative 1e-6 on m), still exact integer sums, hence still independent of summation order,
// tiling and GPU count; half the LDS bytes per table read and integer multiply-adds instead of f64 fma.
template <bool F32> struct TsField;
template <> struct TsField<false> { using w_t = double; using ws_t = double2; static constexpr int SH = 3; };
template <> struct TsField<true> { using w_t = int; using ws_t = int2; static constexpr int SH = 2; };

template <bool TAB_LDS, typename W>
__device__ __forceinline__ W ts_table_at(const W *__restrict__ table_g, uint32_t byte_addr) {
    if (TAB_LDS) {
        typedef __attribute__((address_space(3))) const W lds_cw;
        return *reinterpret_cast<lds_cw *>(byte_addr);        // byte_addr already includes the table's LDS offset
    }
    return *reinterpret_cast<const W *>(reinterpret_cast<const char *>(table_g) + byte_addr);
}

__device__ __forceinline__ void ts_acc(double &acc, double w, int c) { acc = fma(w, (double)c, acc); }
// |w| < 2^23 by construction of the table: v_mad_i32_i24, full rate.  Written as assembly: the compiler reassociates the
// integer sums of a group into multiplies + three-operand adds (30 instructions for 20 multiply-adds).  c is wave-uniform.
__device__ __forceinline__ void ts_acc(int &acc, int w, int c) { asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "s"(c)); }

// |a - b| + c with b wave-uniform (a deposit's site * 8 in a scalar register)
__device__ __forceinline__ uint32_t sad3s(uint32_t a, uint32_t b_uniform, uint32_t c) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
}

// weights of four wave-uniform deposits at this lane's RS sites.  VAR 0: interior (padded LDS table or window: no clamp),
// VAR 1: torus.  x8 / p8 / tlen8 / L8 are site numbers times the entry size (8 or 4 bytes).
template <int VAR, bool TAB_LDS, int RS, bool F32>
__device__ __forceinline__ void ts_weights(const uint32_t (&ent)[4], const uint32_t (&x8)[RS], const uint32_t tbase,
                                           const typename TsField<F32>::w_t *__restrict__ table_g, const uint32_t tlen8, const uint32_t L8,
                                           typename TsField<F32>::w_t (&w)[4][RS]) {
    using W = typename TsField<F32>::w_t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t p8 = (ent[k] & POS_MASK) << TsField<F32>::SH;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            if (VAR == 0) {
                const uint32_t d = sad3s(x8[r], p8, tbase);
                w[k][r] = ts_table_at<TAB_LDS, W>(table_g, TAB_LDS ? d : min(d, tlen8));
            } else {
                const uint32_t d8 = sad3s(x8[r], p8, 0u);
                w[k][r] = ts_table_at<TAB_LDS, W>(table_g, min(min(d8, L8 - d8), tlen8) + tbase);
            }
        }
    }
}

// four deposits of ONE class into ONE accumulator set: a deposit's (cW, cS) is (c, c) [class P: a plus particle came or
// went], (c, -c) [class M: a minus particle] or (0, c) [class F: a flip], so one multiply-add per table read does it.
// MODE 0: coefficient = cW (classes P, M), MODE 1: coefficient = cS (class F).  W = P + M, S = P - M + F, all exact.
template <int VAR, bool TAB_LDS, int RS, int MODE, bool F32>
__device__ __forceinline__ void ts_group(const uint4 q, const uint32_t (&x8)[RS], const uint32_t tbase,
                                         const typename TsField<F32>::w_t *__restrict__ table_g,
                                         const uint32_t tlen8, const uint32_t L8, typename TsField<F32>::w_t (&acc)[RS]) {
    const uint32_t ent[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)q.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.y),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)q.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.w)};
    typename TsField<F32>::w_t w[4][RS];
    ts_weights<VAR, TAB_LDS, RS, F32>(ent, x8, tbase, table_g, tlen8, L8, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = MODE == 0 ? (int)((ent[k] >> 27) & 3u) - 1 : (int)(ent[k] >> 29) - 2;
#pragma unroll
        for (int r = 0; r < RS; ++r) ts_acc(acc[r], w[k][r], c);
    }
}

// The same for four deposits that all lie on ONE side of the whole frame, table window in LDS (windowed sweep, groups j >= 1):
// the lane's RS sites are 64 apart, so their distances to a deposit are too -- one subtraction gives the address of the
// nearest row and the other rows are immediate offsets of the LDS reads (6 VALU per deposit x RS rows instead of 2 RS).
// DIR 0: deposits right of the frame, xb = x8[RS - 1] - window base;  DIR 1: left of it, xb = x8[0] + window base.
// Used with the 32-bit field only: binary64 rows 64 entries apart get merged into ds_read2st64_b64, half the LDS rate of
// separate ds_read_b64 (MI355X_MICROARCH.md, LDS table), and forcing separate reads (measured) still lost to the plain
// |x - p| form with one list per class -- the binary64 sweep is LDS bound, not VALU bound.
template <int RS, int MODE, bool F32, int DIR>
__device__ __forceinline__ void ts_group_dir(const uint4 q, const uint32_t xb, typename TsField<F32>::w_t (&acc)[RS]) {
    using W = typename TsField<F32>::w_t;
    typedef __attribute__((address_space(3))) const char lds_cc;
    typedef __attribute__((address_space(3))) const W lds_cw;
    constexpr int SH = TsField<F32>::SH;
    const uint32_t ent[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)q.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.y),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)q.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.w)};
    W w[4][RS];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t p8 = (ent[k] & POS_MASK) << SH;
        lds_cc *a0 = reinterpret_cast<lds_cc *>(DIR == 0 ? p8 - xb : xb - p8);
#pragma unroll
        for (int r = 0; r < RS; ++r) w[k][r] = *reinterpret_cast<lds_cw *>(a0 + ((DIR == 0 ? RS - 1 - r : r) * (64 << SH)));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = MODE == 0 ? (int)((ent[k] >> 27) & 3u) - 1 : (int)(ent[k] >> 29) - 2;
#pragma unroll
        for (int r = 0; r < RS; ++r) ts_acc(acc[r], w[k][r], c);
    }
}

// one deposit (the last one to three of a list: no padding to whole groups)
template <int RS, int MODE, bool F32, int DIR>
__device__ __forceinline__ void ts_one_dir(const uint32_t entry, const uint32_t xb, typename TsField<F32>::w_t (&acc)[RS]) {
    using W = typename TsField<F32>::w_t;
    typedef __attribute__((address_space(3))) const char lds_cc;
    typedef __attribute__((address_space(3))) const W lds_cw;
    constexpr int SH = TsField<F32>::SH;
    const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)entry);
    const uint32_t p8 = (ent & POS_MASK) << SH;
    W w[RS];
    lds_cc *a0 = reinterpret_cast<lds_cc *>(DIR == 0 ? p8 - xb : xb - p8);
#pragma unroll
    for (int r = 0; r < RS; ++r) w[r] = *reinterpret_cast<lds_cw *>(a0 + ((DIR == 0 ? RS - 1 - r : r) * (64 << SH)));
    const int c = MODE == 0 ? (int)((ent >> 27) & 3u) - 1 : (int)(ent >> 29) - 2;
#pragma unroll
    for (int r = 0; r < RS; ++r) ts_acc(acc[r], w[r], c);
}

template <int RS, int MODE, bool F32>
__device__ __forceinline__ void ts_one_abs(const uint32_t entry, const uint32_t (&x8)[RS], const uint32_t tbase, typename TsField<F32>::w_t (&acc)[RS]) {
    using W = typename TsField<F32>::w_t;
    const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)entry);
    const uint32_t p8 = (ent & POS_MASK) << TsField<F32>::SH;
    W w[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) w[r] = ts_table_at<true, W>(nullptr, sad3s(x8[r], p8, tbase));
    const int c = MODE == 0 ? (int)((ent >> 27) & 3u) - 1 : (int)(ent >> 29) - 2;
#pragma unroll
    for (int r = 0; r < RS; ++r) ts_acc(acc[r], w[r], c);
}

// Inside the kernel sites carry a bias so that the mirror images of deposits beyond a reflecting wall (sites -1 - p and
// 2L - 1 - p) are ordinary, non-negative site numbers: |x - p| is what the sweep computes, biased or not.
constexpr uint32_t TS_BIAS = 1u << 26;     // L <= 2^25 and reach <= L: every biased site stays below 2^27 (the entry's site field)

// small boxes only (reach comparable to L: a deposit's two wall images can both matter within one frame): direct term +
// folded image term per lane, W and S accumulated separately
template <bool TAB_LDS, int RS, bool F32>
__device__ __forceinline__ void ts_image_group(const uint4 q, const uint32_t (&x8)[RS], const uint32_t tbase,
                                               const typename TsField<F32>::w_t *__restrict__ table_g,
                                               const uint32_t tlen8, const uint32_t L8, typename TsField<F32>::w_t (&accW)[RS],
                                               typename TsField<F32>::w_t (&accS)[RS]) {
    using W = typename TsField<F32>::w_t;
    constexpr int SH = TsField<F32>::SH;
    const uint32_t ent[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)q.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.y),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)q.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.w)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t p8 = (ent[k] & POS_MASK) << SH;         // biased, like x8
        const int cw = (int)((ent[k] >> 27) & 3u) - 1, cs = (int)(ent[k] >> 29) - 2;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const uint32_t s8 = x8[r] + p8 + (1u << SH) - 2u * (TS_BIAS << SH);   // (x + p + 1) * entry size
            const W w = ts_table_at<TAB_LDS, W>(table_g, min(sad3s(x8[r], p8, 0u), tlen8) + tbase) +
                        ts_table_at<TAB_LDS, W>(table_g, min(min(s8, 2u * L8 - s8), tlen8) + tbase);
            ts_acc(accW[r], w, cw);
            ts_acc(accS[r], w, cs);
        }
    }
}

#ifdef APS_TS_WAVES                        /* tuning builds: pin the waves per SIMD the register allocator aims for */
#define TS_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(APS_TS_WAVES, APS_TS_WAVES)))
#else
#define TS_WAVES_ATTR
#endif
// K1: site capacity 1 (one cell per site): every loop over a site's cells disappears
template <int BC, bool TAB_LDS, int RS, bool K1, bool F32>
__global__ TS_WAVES_ATTR __launch_bounds__(FU_THREADS) void tile_step(const TileArgs a, const void *__restrict__ table_v) {
    using W = typename TsField<F32>::w_t;
    using WS = typename TsField<F32>::ws_t;
    constexpr int SH = TsField<F32>::SH, WB = (int)sizeof(W);
    const W *__restrict__ table_g = reinterpret_cast<const W *>(table_v);
    constexpr int TS = 64 * RS, NOLD = (TS + FU_THREADS - 1) / FU_THREADS;
    constexpr int SEG = TS_SEG;
    constexpr int SB = TsGeom<TAB_LDS>::SB, NSLOT = TsGeom<TAB_LDS>::NSLOT, GB = TsGeom<TAB_LDS>::GB;
    constexpr int NR = K1 ? ((TS + 2 + FU_WAVES - 1) / FU_WAVES + 63) / 64 : TS_CREG;   // register rounds of the wave's cell chunk
    extern __shared__ double lds[];
    const int L = a.L, K = K1 ? 1 : a.K, OWN = a.own;
    const TsLds lay = ts_lds_layout(a.tlen, TAB_LDS, RS, OWN, K, WB);
    char *lds_c = reinterpret_cast<char *>(lds);
    uint32_t *seg_all = reinterpret_cast<uint32_t *>(lds_c + lay.seg);
    uint32_t *cellL = reinterpret_cast<uint32_t *>(lds_c + lay.cells);       // [(TS + 2) K]: frame positions -1 .. TS
    uint8_t *propL = reinterpret_cast<uint8_t *>(lds_c + lay.props);         // [TS K]
    uint8_t *occL = reinterpret_cast<uint8_t *>(lds_c + lay.occ);            // [TS + 2]
    int *misc = reinterpret_cast<int *>(lds_c + lay.misc);                   // 0 deposits of this tile, 1 spin sum, 2 live count, 4/5 global sums
    W *tab = reinterpret_cast<W *>(lds_c + lay.tab);
    WS *red = reinterpret_cast<WS *>(tab);
    double2 *fieldL = reinterpret_cast<double2 *>(lds_c + lay.field);        // [TS]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), e = blockIdx.y;
    uint32_t *segP = seg_all + wave * 4 * (SEG + 4), *segM = segP + SEG + 4, *segF = segM + SEG + 4, *segI = segF + SEG + 4;
    uint2 *plist = reinterpret_cast<uint2 *>(lds_c + lay.plist);   // the frame's particles {pos | k << 16, cell}, pooled over the four waves
    const int tile = a.tile_lo + (int)blockIdx.x;
    const int own0 = tile * OWN, own_n = min(OWN, L - own0), nfr = own_n + 4;   // owned sites, valid frame positions
    const int x0 = own0 - 2;                                   // site of frame position 0 (may lie outside the lattice)
    const int x0c = max(x0, 0), x1c = min(x0 + TS - 1, L - 1); // the frame clipped to the lattice (range tests)
    const int Rt = a.tlen - 1;
    // per-ensemble bases (scalar); everything below indexes them with 32-bit offsets
    const uint32_t *__restrict__ cell_e = a.cell_in + (size_t)e * L * K;
    const WS *__restrict__ ws_e = reinterpret_cast<const WS *>(a.ws_in) + (size_t)e * L;
    const uint32_t *__restrict__ dcnt_e = a.dcnt_in + (size_t)e * a.ntile;
    const uint32_t *__restrict__ dep_e = a.dep_in + (size_t)e * a.ntile * a.dcap;
    uint32_t tbase = 0;
    {
        typedef __attribute__((address_space(3))) W lds_w;
        tbase = (uint32_t)(size_t)(lds_w *)tab;               // LDS byte offset of the table (or of the windows)
    }
    if (t < 32) misc[t] = 0;                                   // counters used before the first data barrier (particle list)
    __syncthreads();
#ifdef APS_STAMPS
    unsigned long long f_cnt = 0, f_stage = 0, f_copy = 0, f_proc = 0, f_part = 0, f_n = 0, t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long f_start = t0, r_start = __builtin_amdgcn_s_memrealtime();
#define TSTAMP(var) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); var += t1_ - t0; t0 = t1_; }
#else
#define TSTAMP(var)
#endif
    // ---------------------------------------------------------------- requests that depend on nothing
    // The table goes straight into LDS (LDS-direct loads, 1 KB per wave instruction, no registers); the global copy is
    // followed by zeros, so the padded tail comes along.  Written as inline assembly like the windows below (the compiler
    // would put an s_waitcnt vmcnt(0) in front of every later LDS read); the wait in front of the barrier orders it.
    if (TAB_LDS) {
        const int nchunk = ts_table_chunks(a.tlen, RS, OWN, WB);
        const char *srct = reinterpret_cast<const char *>(table_g) + lane * 16;
        for (int c = wave; c < nchunk; c += FU_WAVES) {
            const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tbase + (uint32_t)c * 1024u));
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(srct + c * 1024) : "memory");
        }
    }
    const unsigned long long step = a.stepw[a.par];
    const double beta = a.beta[e];
    if (blockIdx.x == 0 && e == 0 && t == 0 && !a.field_only) a.stepw[a.par ^ 1] = step + 1ull;   // nobody reads that word during this step
    // site of frame position i (-1 .. nfr), or -1: no such site (beyond a wall, or beyond the valid frame)
    auto frame_site = [&](int i) -> int {
        if (i < -1 || i > nfr) return -1;
        int s = x0 + i;
        if (BC == 1) { s %= L; if (s < 0) s += L; return s; }
        return (s < 0 || s >= L) ? -1 : s;
    };
    // The cells of the frame (+1 site either side), a contiguous chunk of Q cells per wave, through registers when a
    // chunk fits: the wave then compacts ITS particles right away (ballot + mbcnt, no barrier) and draws their Philox
    // numbers before the sweep.  All loads are unconditional (clamped addresses, values selected afterwards).
    const int ncell = (TS + 2) * K, Q = lay.Q;
    const int c_lo = wave * Q, c_hi = min(c_lo + Q, ncell);
    uint32_t creg[NR];
    int cpk[NR];
    const bool regs = K1 || lay.cells_in_regs;
    if (regs) {
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int c = c_lo + lane + 64 * u;
            const int pos = K1 ? c - 1 : c / K - 1, k = K1 ? 0 : c - (pos + 1) * K;
            const int s = c < c_hi ? frame_site(pos) : -1;
            creg[u] = cell_e[(unsigned)max(s, 0) * (unsigned)K + (unsigned)k];
            cpk[u] = s < 0 ? -2 : ((pos >= 0 && pos < nfr) ? (pos | (k << 16)) : -1);   // -2: no site, -1: only an occupancy neighbour
        }
    } else {
        for (int c = t; c < ncell; c += FU_THREADS) {
            const int s = frame_site(c / K - 1);
            const uint32_t v = cell_e[(unsigned)max(s, 0) * (unsigned)K + (unsigned)(c % K)];
            cellL[c] = s >= 0 ? v : CELL_EMPTY;
        }
    }
    // buckets (= tiles) whose deposits can reach the frame: one run of nbk buckets from b0 that may wrap around the torus
    int b0 = 0, nbk = 0;
    if (a.field_mode) {
        if (BC == 0) {
            b0 = max(0, x0c - Rt - 1) / OWN;
            nbk = min(L - 1, x1c + Rt + 1) / OWN - b0 + 1;
        } else {
            const int lo = x0 - Rt - 1, hi = x0 + TS - 1 + Rt + 1;
            if (hi - lo + 1 >= L) { b0 = 0; nbk = a.ntile; }
            else {
                const int lom = ((lo % L) + L) % L, him = ((hi % L) + L) % L;
                const int blo = lom / OWN, bhi = him / OWN;
                b0 = blo;
                nbk = (lom <= him) ? bhi - blo + 1 : (a.ntile - blo) + bhi + 1;
                if (nbk > a.ntile) { b0 = 0; nbk = a.ntile; }
            }
        }
    }
    const bool wall = BC == 0 && ((x0c + 1 <= Rt) || (L - x1c <= Rt));   // an image term can be non-zero
    // box much wider than the reach: at most one wall image of a deposit can matter to one frame, and where it is out of a
    // lane's reach the zero-padded table says so -- the image is then just another plain deposit (at the mirrored site)
    const bool mirror_ok = 2 * Rt + TS + OWN + 4 < L;
    const int sub = lane >> SB, slot = lane & (NSLOT - 1);
    const unsigned slot_c = (unsigned)min(slot, a.dcap - 1);
    uint32_t pre_cnt[FU_PRE], pre_ent[FU_PRE];
#pragma unroll
    for (int j = 0; j < FU_PRE; ++j) {
        const int bi = j * GB + sub * FU_WAVES + wave;
        const bool ok = bi < nbk;
        int b = b0 + (ok ? bi : 0);
        if (b >= a.ntile) b -= a.ntile;
        pre_cnt[j] = dcnt_e[(unsigned)b];                      // entries beyond the count are never looked at
        pre_ent[j] = dep_e[(unsigned)b * (unsigned)a.dcap + slot_c];
        if (!ok) pre_cnt[j] = 0u;
    }
    WS old[NOLD];
#pragma unroll
    for (int r = 0; r < NOLD; ++r) {
        const int xi = r * FU_THREADS + t;
        const int s = (xi < TS && xi < nfr) ? frame_site(xi) : -1;
        old[r] = ws_e[(unsigned)max(s, 0)];
        if (s < 0 || !a.field_mode) { old[r].x = 0; old[r].y = 0; }
    }
    // global-field mode (ref :219-221): the sums over all particles = sum of the tiles' parts
    long long gS = 0, gN = 0;
    if (!a.field_mode) {
        for (int i = t; i < a.ntile; i += FU_THREADS) { gS += a.gpart_in[((size_t)e * a.ntile + i) * 2]; gN += a.gpart_in[((size_t)e * a.ntile + i) * 2 + 1]; }
    }
    for (int i = t; i < (TS * K + 3) / 4; i += FU_THREADS) reinterpret_cast<uint32_t *>(propL)[i] = 0u;   // EV_NONE everywhere
    // the frame's particles: every wave appends the occupied cells of its chunk to ONE list (an LDS atomic per wave and
    // round), so that afterwards the proposals run a lane per particle over full wavefronts
    if (regs) {
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            if (cpk[u] == -2) creg[u] = CELL_EMPTY;
            const bool occ = creg[u] != CELL_EMPTY && cpk[u] >= 0;
            const unsigned long long mm = __ballot(occ);
            const int cnt_u = __popcll(mm);
            int base = 0;
            if (lane == 0 && cnt_u) base = atomicAdd(&misc[3], cnt_u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (occ) plist[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u))] = make_uint2((uint32_t)cpk[u], creg[u]);
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) { const int c = c_lo + lane + 64 * u; if (c < c_hi) cellL[c] = creg[u]; }
    }
    const uint32_t tlen8 = (uint32_t)a.tlen << SH, L8 = (uint32_t)L << SH;
    uint32_t x8[RS];
    W accP[RS], accM[RS], accF[RS], accWi[RS], accSi[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        int s = x0 + r * 64 + lane;
        if (BC == 1) { s %= L; if (s < 0) s += L; } else s = min(max(s, 0), L - 1);
        x8[r] = ((uint32_t)s + TS_BIAS) << SH; accP[r] = accM[r] = accF[r] = accWi[r] = accSi[r] = 0;
    }
    if (TAB_LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's table chunks have landed
    __syncthreads();                                           // table, cells and the particle list staged
    const int n_part = regs ? misc[3] : 0;
    uint2 mine = make_uint2(0u, CELL_EMPTY);
    uint32_t rx[4] = {0u, 0u, 0u, 0u};
    if (!a.field_only && t < n_part) {                         // this thread's particle: its random numbers now, ahead of the sweep
        mine = plist[t];
        philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), mine.y & CELL_ID, (uint32_t)(a.ens_base + e), a.seed_lo, a.seed_hi, rx);
    }
    TSTAMP(f_stage)
    // ---------------------------------------------------------------- 1  deposits of the previous step -> W, S of the frame
    const bool windowed = !TAB_LDS && BC == 0 && !wall;
    const uint32_t tb = TAB_LDS ? tbase : 0u;                  // table in global memory: byte offsets from its start
    const uint32_t win_lds = tbase;
    const int WIN = ts_win_entries(RS, OWN);
    const int null_site = x0c;
    int nP = 0, nM = 0, nF = 0, nI = 0;                        // entries waiting in this wave's four segments
    const uint4 *segP4 = reinterpret_cast<const uint4 *>(segP), *segM4 = reinterpret_cast<const uint4 *>(segM),
                *segF4 = reinterpret_cast<const uint4 *>(segF), *segI4 = reinterpret_cast<const uint4 *>(segI);
    auto flush = [&]() {                                       // sweep the frame with the segments' deposits, class by class
        if (lane < 8) {                                        // pad to whole groups of four, one group beyond (the sweep reads one group ahead)
            const uint32_t pad = DEP_NULL | ((uint32_t)null_site + TS_BIAS);   // a site whose distances stay in table range, coefficients 0
            segP[nP + lane] = pad; segM[nM + lane] = pad; segF[nF + lane] = pad; segI[nI + lane] = DEP_NULL | ((uint32_t)x0c + TS_BIAS);
        }
#define TS_SWEEP(SEG4, N, MODE, ACC) { \
        uint4 q = SEG4[0]; \
        _Pragma("unroll 1") for (int i = 0; i < ((N) + 3) >> 2; ++i) { \
            const uint4 qn = SEG4[i + 1];                      /* next group's entries: in flight during this group's gathers */ \
            if (BC == 1) ts_group<1, TAB_LDS, RS, MODE, F32>(q, x8, tb, table_g, tlen8, L8, ACC); \
            else ts_group<0, TAB_LDS, RS, MODE, F32>(q, x8, tb, table_g, tlen8, L8, ACC); \
            q = qn; } }
        TS_SWEEP(segP4, nP, 0, accP)
        TS_SWEEP(segM4, nM, 0, accM)
        TS_SWEEP(segF4, nF, 1, accF)
#undef TS_SWEEP
#pragma unroll 1
        for (int i = 0; i < (nI + 3) >> 2; ++i) ts_image_group<TAB_LDS, RS, F32>(segI4[i], x8, tb, table_g, tlen8, L8, accWi, accSi);
#ifdef APS_STAMPS
        f_n += nP + nM + nF + nI;
#endif
        nP = nM = nF = nI = 0;
    };
    const int ngroups = (nbk + GB - 1) / GB;
    if (!TAB_LDS && windowed) {
        // Table beyond LDS, interior tile.  The buckets at offsets -off and +off from this tile are at the same distances,
        // so they share one window of the table: group j = offsets [jH, jH + H) on both sides (16 buckets x 16 list slots =
        // one load per lane).  The four waves pool a group's deposits into SHARED lists per (side, class) -- a wave's buckets
        // are all on one side (waves 0, 2: right of the tile, 1, 3: left), one LDS atomic per wave -- and every wave then
        // takes whole groups of four from those lists: no per-wave padding, half the barriers.  From group 1 on every deposit
        // lies on one side of the whole frame (ts_group_dir).
        // Window j (distances [(jH - 1) OWN - 2, (jH + H) OWN + 2]) is double buffered and requested one group ahead by
        // LDS-direct loads; list words likewise.  Shared lists: double buffered; their packed lengths: triple buffered.
        constexpr int H = TS_WH;
        static_assert(2 * H * 16 == FU_THREADS, "one list load per lane: 2 H buckets x 16 slots = the workgroup");
        // the 32-bit field splits the lists by side (directional gathers from group 1 on); with the binary64 field the
        // gathers are LDS bound either way and the shorter lists only cost pipeline fills: one list per class
        constexpr int NSIDE = F32 ? 2 : 1, CAP = TS_SHCAP / NSIDE;
        uint32_t *shl = seg_all;                               // [2][NSIDE][3][CAP], over the per-wave segments (unused here)
        static_assert(2 * 3 * TS_SHCAP <= FU_WAVES * 4 * (TS_SEG + 4), "the shared lists fit the segments' space");
        int *scnt = misc + 8;                                  // [3][3]: per side the lengths nP | nM << 10 | nF << 20, "a bucket has more than 16" flag
        const int side_w = F32 ? (wave & 1) : 0;               // wave & 1 = bsel & 1: this wave's buckets are tile - off (1) or tile + off (0)
        const int side = max(tile - b0, b0 + nbk - 1 - tile);
        const int ngr = side / H + 1;
        const int bsel = (lane >> 4) * FU_WAVES + wave, sl16 = lane & 15;
        const unsigned sl16c = (unsigned)min(sl16, a.dcap - 1);
        auto bucket_of = [&](int j, bool &okb) -> int {
            const int off = j * H + (bsel >> 1);
            const bool neg = (bsel & 1) != 0;
            const int b = neg ? tile - off : tile + off;
            okb = off <= side && !(neg && off == 0) && b >= b0 && b < b0 + nbk;
            return okb ? b : tile;
        };
        // distances of the frame to the deposits of buckets tile +- off, off in [jH, jH + H) (deposit sites lie within one site of their bucket)
        auto win_dmin = [&](int j) -> int { return max(0, min((j * H - 1) * OWN - 2, j * H * OWN + 2 - TS)); };
        auto stage_w = [&](int j) {
            const int dmin = win_dmin(j), span = max((j * H + H) * OWN + 2, (j * H + H - 1) * OWN + TS - 2) - dmin;
            const char *srcw = reinterpret_cast<const char *>(table_g + dmin) + lane * 16;
            const uint32_t dstw = win_lds + (uint32_t)((j & 1) * WIN) * (uint32_t)WB;
            for (int c = wave; c * (1024 / WB) <= span; c += FU_WAVES) {
                const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dstw + (uint32_t)c * 1024u));
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(srcw + c * 1024) : "memory");
            }
        };
        stage_w(0);
        bool okb;
        int b = bucket_of(0, okb);
        uint32_t cnt = dcnt_e[(unsigned)b], ent = dep_e[(unsigned)b * (unsigned)a.dcap + sl16c];
        if (!okb) cnt = 0u;
        for (int j = 0; j < ngr; ++j) {
            TSTAMP(f_copy)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's chunks of window j and its list words have landed
            asm volatile("" : "+v"(cnt), "+v"(ent));
            cnt = min(cnt, (uint32_t)a.dcap);
            const int jj = j & 1, j3 = j % 3;
            {   // pool this wave's entries of group j
                const uint32_t en_ = ent + TS_BIAS;
                const bool valid = (uint32_t)sl16 < cnt;
                const int cw = (int)((en_ >> 27) & 3u) - 1, cs = (int)(en_ >> 29) - 2;
                const bool cP = valid && cw != 0 && cw == cs, cM = valid && cw != 0 && cw != cs, cF = valid && cw == 0;
                const unsigned long long mP = __ballot(cP), mM = __ballot(cM), mF = __ballot(cF);
                const int add = __popcll(mP) | (__popcll(mM) << 10) | (__popcll(mF) << 20);
                int base = 0;
                if (lane == 0 && add) base = atomicAdd(&scnt[j3 * 3 + side_w], add);
                base = __builtin_amdgcn_readfirstlane(base);
                if (__ballot((uint32_t)sl16 + 16u <= cnt && cnt > 16u) && lane == 0) atomicOr(&scnt[j3 * 3 + 2], 1);
                uint32_t *lst = shl + (size_t)(jj * NSIDE + side_w) * 3 * CAP;
#define TS_POOL(COND, MASK, SHIFT, CLS) if (COND) lst[(CLS) * CAP + ((base >> (SHIFT)) & 1023) + \
                    __builtin_amdgcn_mbcnt_hi((uint32_t)((MASK) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(MASK), 0u))] = en_;
                TS_POOL(cP, mP, 0, 0) TS_POOL(cM, mM, 10, 1) TS_POOL(cF, mF, 20, 2)
#undef TS_POOL
            }
            __syncthreads();                                   // lists j complete, window j landed everywhere, everyone is done with group j - 1
            if (t < 3) scnt[((j + 2) % 3) * 3 + t] = 0;        // nobody touches that triple before the next barrier
            const uint32_t wbase = win_lds + (uint32_t)(jj * WIN) * (uint32_t)WB - ((uint32_t)win_dmin(j) << SH);
            const uint32_t cnt_now = cnt;
            const int b_now = b;
            if (j + 1 < ngr) {                                 // next window and next list words: in flight during this sweep
                stage_w(j + 1);
                b = bucket_of(j + 1, okb);
                cnt = dcnt_e[(unsigned)b]; ent = dep_e[(unsigned)b * (unsigned)a.dcap + sl16c];
                if (!okb) cnt = 0u;
            }
            TSTAMP(f_cnt)
            const int packedR = scnt[j3 * 3], packedL = scnt[j3 * 3 + 1], more = scnt[j3 * 3 + 2];
            int rot = wave;                                    // whole groups of four are dealt round-robin across the lists
#define TS_SHARED(SIDE, CLS, PACKED, CALL4, CALL1) { \
                const int n_ = ((PACKED) >> (10 * (CLS))) & 1023, nfull = n_ >> 2, rem = n_ & 3; \
                const uint4 *l4 = reinterpret_cast<const uint4 *>(shl + ((size_t)(jj * NSIDE + (SIDE)) * 3 + (CLS)) * CAP); \
                int g = rot & 3; \
                uint4 q = l4[min(g, CAP / 4 - 1)]; \
                _Pragma("unroll 1") for (; g < nfull; g += FU_WAVES) { \
                    const uint4 qn = l4[min(g + FU_WAVES, CAP / 4 - 1)]; \
                    CALL4; \
                    q = qn; } \
                if (g == nfull && rem) {                       /* the list's last one to three entries: one by one, by the wave whose turn it is */ \
                    { const uint32_t q1 = q.x; CALL1; } \
                    if (rem > 1) { const uint32_t q1 = q.y; CALL1; } \
                    if (rem > 2) { const uint32_t q1 = q.z; CALL1; } } \
                rot = (rot - nfull - (rem ? 1 : 0)) & 3; }
            if (!F32 || j == 0) {                              // group 0 = the tile itself and its neighbours: deposits inside the frame, |x - p|
                TS_SHARED(0, 0, packedR, (ts_group<0, true, RS, 0, F32>(q, x8, wbase, table_g, tlen8, L8, accP)), (ts_one_abs<RS, 0, F32>(q1, x8, wbase, accP)))
                if (F32) TS_SHARED(1, 0, packedL, (ts_group<0, true, RS, 0, F32>(q, x8, wbase, table_g, tlen8, L8, accP)), (ts_one_abs<RS, 0, F32>(q1, x8, wbase, accP)))
                TS_SHARED(0, 1, packedR, (ts_group<0, true, RS, 0, F32>(q, x8, wbase, table_g, tlen8, L8, accM)), (ts_one_abs<RS, 0, F32>(q1, x8, wbase, accM)))
                if (F32) TS_SHARED(1, 1, packedL, (ts_group<0, true, RS, 0, F32>(q, x8, wbase, table_g, tlen8, L8, accM)), (ts_one_abs<RS, 0, F32>(q1, x8, wbase, accM)))
                TS_SHARED(0, 2, packedR, (ts_group<0, true, RS, 1, F32>(q, x8, wbase, table_g, tlen8, L8, accF)), (ts_one_abs<RS, 1, F32>(q1, x8, wbase, accF)))
                if (F32) TS_SHARED(1, 2, packedL, (ts_group<0, true, RS, 1, F32>(q, x8, wbase, table_g, tlen8, L8, accF)), (ts_one_abs<RS, 1, F32>(q1, x8, wbase, accF)))
            } else if constexpr (F32) {
                const uint32_t xbR = x8[RS - 1] - wbase, xbL = x8[0] + wbase;
                TS_SHARED(0, 0, packedR, (ts_group_dir<RS, 0, F32, 0>(q, xbR, accP)), (ts_one_dir<RS, 0, F32, 0>(q1, xbR, accP)))
                TS_SHARED(1, 0, packedL, (ts_group_dir<RS, 0, F32, 1>(q, xbL, accP)), (ts_one_dir<RS, 0, F32, 1>(q1, xbL, accP)))
                TS_SHARED(0, 1, packedR, (ts_group_dir<RS, 0, F32, 0>(q, xbR, accM)), (ts_one_dir<RS, 0, F32, 0>(q1, xbR, accM)))
                TS_SHARED(1, 1, packedL, (ts_group_dir<RS, 0, F32, 1>(q, xbL, accM)), (ts_one_dir<RS, 0, F32, 1>(q1, xbL, accM)))
                TS_SHARED(0, 2, packedR, (ts_group_dir<RS, 1, F32, 0>(q, xbR, accF)), (ts_one_dir<RS, 1, F32, 0>(q1, xbR, accF)))
                TS_SHARED(1, 2, packedL, (ts_group_dir<RS, 1, F32, 1>(q, xbL, accF)), (ts_one_dir<RS, 1, F32, 1>(q1, xbL, accF)))
            }
#undef TS_SHARED
            if (more) {                                        // rare: a bucket of this group holds more than 16 deposits -> one by one
                for (uint32_t k0 = 16; __ballot(k0 < cnt_now); k0 += 16) {
                    const bool valid = k0 + (uint32_t)sl16 < cnt_now;
                    const uint32_t en = dep_e[(unsigned)b_now * (unsigned)a.dcap + min(k0 + (uint32_t)sl16, (uint32_t)a.dcap - 1u)] + TS_BIAS;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (also waits for the next window: harmless here)
                    unsigned long long mm = __ballot(valid);
                    while (mm) {
                        const int src_lane = __builtin_ctzll(mm);
                        mm &= mm - 1;
                        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)en, src_lane);
                        const int cw = (int)((e1 >> 27) & 3u) - 1, cs = (int)(e1 >> 29) - 2;
                        if (cw == 0) ts_one_abs<RS, 1, F32>(e1, x8, wbase, accF);
                        else if (cw == cs) ts_one_abs<RS, 0, F32>(e1, x8, wbase, accP);
                        else ts_one_abs<RS, 0, F32>(e1, x8, wbase, accM);
                    }
                }
            }
            TSTAMP(f_proc)
        }
    } else {
    for (int j = 0; j < ngroups; ++j) {
        const int bi = j * GB + sub * FU_WAVES + wave;
        const bool ok = bi < nbk;
        int b = b0 + (ok ? bi : 0);
        if (b >= a.ntile) b -= a.ntile;
        uint32_t cnt, ent;
        if (j < FU_PRE) { cnt = j == 0 ? pre_cnt[0] : pre_cnt[FU_PRE - 1]; ent = j == 0 ? pre_ent[0] : pre_ent[FU_PRE - 1]; }
        else {
            cnt = dcnt_e[(unsigned)b];
            ent = dep_e[(unsigned)b * (unsigned)a.dcap + slot_c];
            if (!ok) cnt = 0u;
        }
        cnt = min(cnt, (uint32_t)a.dcap);
        // NSLOT slots of each of the wave's buckets: compact the valid ones into the wave's segments by class.  Near a
        // reflecting wall the image of a deposit is the same deposit at the mirrored site (small boxes: image segment).
#define TS_PUT(SEGX, NX, COND, WORD) { const unsigned long long m_ = __ballot(COND); \
            if (COND) SEGX[NX + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u))] = (WORD); \
            NX += __popcll(m_); }
#define TS_FULL() (nP + 64 > SEG - 4 || nM + 64 > SEG - 4 || nF + 64 > SEG - 4 || nI + 64 > SEG - 4)
#define TS_ROUND(EN, K0) { \
            const uint32_t en_ = (EN) + TS_BIAS;               /* site field biased (no carry: sites < 2^25) */ \
            const bool valid = (K0) + slot < cnt; \
            const int dp = (int)((EN) & POS_MASK), cw = (int)((en_ >> 27) & 3u) - 1, cs = (int)(en_ >> 29) - 2; \
            const bool img_l = valid && wall && (x0c + dp + 1 <= Rt), img_r = valid && wall && (2 * L - 1 - x1c - dp <= Rt); \
            const bool cP = cw != 0 && cw == cs, cM = cw != 0 && cw != cs, cF = cw == 0; \
            const bool pl = valid && (mirror_ok || !(img_l || img_r)); \
            if (TS_FULL()) flush(); \
            TS_PUT(segP, nP, (pl && cP), en_) TS_PUT(segM, nM, (pl && cM), en_) TS_PUT(segF, nF, (pl && cF), en_) \
            if (wall && mirror_ok) { \
                const bool im = img_l || img_r; \
                const uint32_t mir = (en_ & ~POS_MASK) | (uint32_t)((int)TS_BIAS + (img_l ? -1 - dp : 2 * L - 1 - dp)); \
                if (TS_FULL()) flush(); \
                TS_PUT(segP, nP, (im && cP), mir) TS_PUT(segM, nM, (im && cM), mir) TS_PUT(segF, nF, (im && cF), mir) \
            } else if (wall) TS_PUT(segI, nI, (valid && (img_l || img_r)), en_) }
        for (uint32_t k0 = 0;; k0 += NSLOT) {
            TS_ROUND(ent, k0)
            if (!__ballot(k0 + NSLOT < cnt)) break;
            ent = dep_e[(unsigned)b * (unsigned)a.dcap + min(k0 + NSLOT + slot, (uint32_t)a.dcap - 1u)];
        }
#undef TS_ROUND
#undef TS_FULL
#undef TS_PUT
    }
    }
    TSTAMP(f_copy)
    if (TAB_LDS || !windowed) flush();                         // (the shared lists of the windowed sweep live in the segments' space)
    TSTAMP(f_proc)
    __syncthreads();                                           // every wave is done with the table: its space takes the partial sums
#pragma unroll
    for (int r = 0; r < RS; ++r)                               // W = P + M, S = P - M + F (+ the image deposits), exact on the weight grid
        { WS v; v.x = (accP[r] + accM[r]) + accWi[r]; v.y = ((accP[r] - accM[r]) + accF[r]) + accSi[r]; red[(size_t)wave * TS + r * 64 + lane] = v; }
    // occupancy of the frame sites (-1 .. TS) from the staged cells
    for (int i = t; i < TS + 2; i += FU_THREADS) {
        int n = 0;
        for (int k = 0; k < K; ++k) n += cellL[i * K + k] != CELL_EMPTY;
        occL[i] = (uint8_t)n;
    }
    if (!a.field_mode) {                                       // workgroup sums of the global-field parts
        const long long s1 = wave_sum(gS), s2 = wave_sum(gN);
        if (lane == 0) { atomicAdd(&misc[4 + 0], (int)s1); atomicAdd(&misc[4 + 1], (int)s2); }
    }
    __syncthreads();
    // ---------------------------------------------------------------- 2  fresh field of the frame sites
#pragma unroll
    for (int r = 0; r < NOLD; ++r) {
        const int xi = r * FU_THREADS + t;
        if (xi < TS && xi < nfr) {
            WS f = old[r];
#pragma unroll
            for (int w = 0; w < FU_WAVES; ++w) { const WS pth = red[(size_t)w * TS + xi]; f.x += pth.x; f.y += pth.y; }
            // S / W does not care about the unit (2^-q in the integer field): the proposals read doubles either way
            fieldL[xi] = a.field_mode ? make_double2((double)f.x, (double)f.y) : make_double2((double)misc[5], (double)misc[4]);
            if (a.field_mode && xi >= 2 && xi < 2 + own_n) reinterpret_cast<WS *>(a.ws_out)[(size_t)e * L + (unsigned)frame_site(xi)] = f;
        }
    }
    if (a.field_only) return;                                  // flush of the pending deposits only (observation)
    __syncthreads();
    // ---------------------------------------------------------------- 2b proposals, a lane per particle of this wave
    {
        const Model M = *a.model;                              // uniform address: scalar loads, only now
        auto propose_one = [&](const uint2 pc, const uint32_t (&x)[4]) {
            const int pos = (int)(pc.x & 0xFFFFu), k = (int)(pc.x >> 16);
            const int s = frame_site(pos);
            const double2 f = fieldL[pos];
            const bool anch = a.anchor ? a.anchor[s] != 0 : false;
            propL[pos * K + k] = decide_proposal(M, anch, s, (pc.y & CELL_PLUS) ? 1 : -1, (pc.y & CELL_BOUND) != 0, clip_field(f.y, f.x), beta,
                                                 occL[pos + 1], occL[pos], occL[pos + 2], x);
        };
        if (regs) {
            if (t < n_part) propose_one(mine, rx);
            for (int j = FU_THREADS + t; j < n_part; j += FU_THREADS) {   // more than 256 particles on the frame (K > 1)
                const uint2 pc = plist[j];
                uint32_t x[4];
                philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), pc.y & CELL_ID, (uint32_t)(a.ens_base + e), a.seed_lo, a.seed_hi, x);
                propose_one(pc, x);
            }
        } else {
            for (int c = c_lo + lane; c < c_hi; c += 64) {
                const uint32_t cw = cellL[c];
                const int pos = c / K - 1;
                if (cw == CELL_EMPTY || pos < 0 || pos >= nfr) continue;
                uint32_t x[4];
                philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), cw & CELL_ID, (uint32_t)(a.ens_base + e), a.seed_lo, a.seed_hi, x);
                propose_one(make_uint2((uint32_t)pos | ((uint32_t)(c - (pos + 1) * K) << 16), cw), x);
            }
        }
    }
    __syncthreads();
    TSTAMP(f_part)
    // ---------------------------------------------------------------- 3 + 4  exclusion, new cells of the owned sites, deposits
    // number of proposers of frame site j (0 < j < nfr - 1) with an id below `id`: they sit on j - 1 (moving right) and j + 1 (moving left)
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
    uint32_t *dep_o = a.dep_out + ((size_t)e * a.ntile + tile) * a.dcap;
    uint32_t *cell_o = a.cell_out + (size_t)e * L * K;
    int my_spin = 0, my_live = 0;
    auto emit_deposits = [&](const int nd, const uint32_t d0, const uint32_t d1) {          // the deposits of one event: into the tile's list
        if (!nd || !a.field_mode) return;
        const int kd = atomicAdd(&misc[0], nd);
        if (kd + nd <= a.dcap) { dep_o[kd] = d0; if (nd == 2) dep_o[kd + 1] = d1; }
    };
    // the event of the particle `c` on site s (frame position xi): returns whether it is still on the site afterwards (c updated);
    // hop_granted: the exclusion rule let its hop to frame position j through
    auto own_event = [&](uint32_t &c, const int ev, const int s, const int j, const bool hop_granted) -> bool {
        const int sgn = (c & CELL_PLUS) ? 1 : -1;
        uint32_t d0 = 0, d1 = 0;
        int nd = 0;
        bool stays = true;
        if (ev == EV_LEFT || ev == EV_RIGHT || ev == EV_FWD) {
            if (hop_granted) { stays = false; d0 = deposit(s, -1, -sgn); d1 = deposit(frame_site(j), 1, sgn); nd = 2; }
        } else if (ev == EV_BIND) c |= CELL_BOUND;
        else if (ev == EV_UNBIND) c &= ~CELL_BOUND;
        else if (ev == EV_FLIP) { c ^= CELL_PLUS; d0 = deposit(s, 0, -2 * sgn); nd = 1; }
        else if (ev == EV_EXIT) {
            stays = false;
            const TileRare R = *a.rare;
            if (tile >= R.rec_lo && tile < R.rec_hi) {
                const unsigned kx = atomicAdd(&R.n_exit[e], 1u);
                if ((int)kx < R.exit_cap) {
                    double *row = R.exit_log + ((size_t)e * R.exit_cap + kx) * 3;
                    row[0] = (double)step; row[1] = (double)s; row[2] = (double)(c & CELL_ID);
                }
                R.src[(size_t)e * R.Npad + R.slot_of[(size_t)e * R.N + (c & CELL_ID)]] =
                    (uint32_t)s | DEAD_BIT | ((c & CELL_PLUS) ? SPIN_BIT : 0u) | ((c & CELL_BOUND) ? BOUND_BIT : 0u);
            }
            d0 = deposit(s, -1, -sgn); nd = 1;
        }
        emit_deposits(nd, d0, d1);
        return stays;
    };
#pragma unroll
    for (int r = 0; r < NOLD; ++r) {
        const int xi = r * FU_THREADS + t;
        if (xi < 2 || xi >= 2 + own_n || xi >= TS) continue;
        const int s = frame_site(xi);
        uint32_t *out = cell_o + (unsigned)s * (unsigned)K;
        if constexpr (K1) {
            // One cell per site: a hop is only ever proposed into a site that was empty at the start of the step (channels(): open_l /
            // open_r), so its capacity is 1 and the only rival is the particle on the far side of the target -- the rule of DESIGN 3
            // (smaller id wins) on a window of five sites, read once (tile_loop's K = 1 path; 10 LDS reads instead of ~24)
            const uint32_t cm2 = cellL[xi - 1], cm1 = cellL[xi], c0 = cellL[xi + 1], cp1 = cellL[xi + 2], cp2 = cellL[xi + 3];
            const int pm2 = propL[xi - 2] & 7, pm1 = propL[xi - 1] & 7, p0 = propL[xi] & 7, pp1 = propL[xi + 1] & 7, pp2 = xi + 2 < TS ? propL[xi + 2] & 7 : 0;
            const bool from_l2 = cm2 != CELL_EMPTY && (pm2 == EV_RIGHT || pm2 == EV_FWD);     // the particle two sites left wants xi - 1
            const bool from_l1 = cm1 != CELL_EMPTY && (pm1 == EV_RIGHT || pm1 == EV_FWD);     // the left neighbour wants this site
            const bool from_r1 = cp1 != CELL_EMPTY && pp1 == EV_LEFT;                          // the right neighbour wants this site
            const bool from_r2 = cp2 != CELL_EMPTY && pp2 == EV_LEFT;                          // the particle two sites right wants xi + 1
            uint32_t newc = CELL_EMPTY;
            if (c0 != CELL_EMPTY) {
                uint32_t c = c0;
                const bool left = p0 == EV_LEFT;
                const bool granted = left ? !(from_l2 && (cm2 & CELL_ID) < (c & CELL_ID)) : !(from_r2 && (cp2 & CELL_ID) < (c & CELL_ID));
                if (own_event(c, p0, s, left ? xi - 1 : xi + 1, granted)) newc = c;
            }
            if (from_l1 && !(from_r1 && (cp1 & CELL_ID) < (cm1 & CELL_ID))) newc = cm1;        // granted arrivals (this site was empty)
            if (from_r1 && !(from_l1 && (cm1 & CELL_ID) < (cp1 & CELL_ID))) newc = cp1;
            out[0] = newc;
            if (newc != CELL_EMPTY) { my_spin += (newc & CELL_PLUS) ? 1 : -1; my_live += 1; }
            continue;
        }
        int n_out = 0;
        for (int k = 0; k < K; ++k) {                          // the particles on this site: stay (possibly changed) or leave
            uint32_t c = cellL[(xi + 1) * K + k];
            if (c == CELL_EMPTY) continue;
            const int ev = propL[xi * K + k] & 7;
            const bool hop = ev == EV_LEFT || ev == EV_RIGHT || ev == EV_FWD;
            const int j = ev == EV_LEFT ? xi - 1 : xi + 1;
            const bool granted = hop && rank_at(j, c & CELL_ID) < cap_at(j);
            if (own_event(c, ev, s, j, granted)) { out[n_out++] = c; my_spin += (c & CELL_PLUS) ? 1 : -1; my_live += 1; }
        }
        const int cap = cap_at(xi);
        for (int k = 0; k < K; ++k) {                          // granted arrivals from the left and right neighbour
            const uint32_t cl_ = cellL[xi * K + k], cr_ = cellL[(xi + 2) * K + k];
            const int el = propL[(xi - 1) * K + k] & 7, er = propL[(xi + 1) * K + k] & 7;
            if (cl_ != CELL_EMPTY && (el == EV_RIGHT || el == EV_FWD) && rank_at(xi, cl_ & CELL_ID) < cap) {
                out[n_out++] = cl_; my_spin += (cl_ & CELL_PLUS) ? 1 : -1; my_live += 1;
            }
            if (cr_ != CELL_EMPTY && er == EV_LEFT && rank_at(xi, cr_ & CELL_ID) < cap) {
                out[n_out++] = cr_; my_spin += (cr_ & CELL_PLUS) ? 1 : -1; my_live += 1;
            }
        }
        for (int k = n_out; k < K; ++k) out[k] = CELL_EMPTY;
    }
    if (!a.field_mode) {
        const long long s1 = wave_sum((long long)my_spin), s2 = wave_sum((long long)my_live);
        if (lane == 0) { atomicAdd(&misc[1], (int)s1); atomicAdd(&misc[2], (int)s2); }
    }
    __syncthreads();
    if (t == 0) {
        a.dcnt_out[(size_t)e * a.ntile + tile] = (uint32_t)min(misc[0], a.dcap);
        if (!a.field_mode) {
            a.gpart_out[((size_t)e * a.ntile + tile) * 2] = misc[1];
            a.gpart_out[((size_t)e * a.ntile + tile) * 2 + 1] = misc[2];
        }
    }
#ifdef APS_STAMPS
    TSTAMP(f_cnt)
    if (t == 0 && blockIdx.x < 4096) {
        unsigned long long *o = a.rare->stamps + (size_t)blockIdx.x * 8;
        o[0] = f_cnt; o[1] = f_stage; o[2] = f_copy; o[3] = f_proc; o[4] = __builtin_amdgcn_s_memtime() - f_start;
        o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = r_start; o[7] = f_part;
    }
#endif
}
