// aps_hip.hip -- MI355X (gfx950) implementation of the C ABI in include/aps.h.
//
// Hot path replaced: ParticleSystem.run's loop body of the reference, i.e.
//   compute_local_m_field (PARTICLE_solver_CLASS.py:216-246)  +  step_gillespie (:254-448),
// restated as a fixed-dt synchronous stepper (DESIGN.md "The synchronous scheme").
//
// Per step three launches:
//   pair_propose  all-pairs tile kernel: per target particle S = sum sigma_j w(d_ij), W = sum w(d_ij),
//                 occupancy of the neighbouring sites; epilogue = rates -> Philox draw -> proposal byte
//   claim         hop proposals register at their target site (tiny per-site lists)
//   apply         index-ordered arbitration of the hops, state update, per-tile position bounds
//
// Data layout in HBM (one handle, E ensembles, Npad slots each, slot order = internal order):
//   src  [E][Npad] u32   packed particle: bits 0-26 site, 27 spin(+), 28 bound, 29 dead
//   orig [E][Npad] u32   original particle index of the slot (Philox counter, tie-break, output order)
//   prop [world][E][SH] u8  proposals; rank r owns block r (exchanged between ranks once per step)
//   bounds [E][Npad/64] int2  min/max live site per 64-slot tile (tile culling)
//   pcnt [E][L] u32, plist [E][L][2K] u32   per-site proposer lists for the commit
//
// Arithmetic: binary64 throughout; the weight table lives on the grid 2^-q so every partial sum is
// exact and the result does not depend on summation order (any tiling, any number of GPUs, CPU oracle).
// Compile with -ffp-contract=off: the rate/probability code is a fixed sequence of IEEE operations.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "aps.h"
#include "aps_common.hpp"

namespace {

constexpr uint32_t POS_MASK = 0x07FFFFFFu;
constexpr uint32_t SPIN_BIT = 1u << 27;
constexpr uint32_t BOUND_BIT = 1u << 28;
constexpr uint32_t DEAD_BIT = 1u << 29;
constexpr uint32_t AWAY_BIT = 1u << 30;       // site-sharded handles: the particle currently lives on another rank's sites
constexpr uint32_t DEAD_P8 = POS_MASK << 3;   // far-away site (x8) every distance test rejects
constexpr int TILE = 64;                      // slots per tile = one wavefront of targets
#ifndef APS_WAVES
#define APS_WAVES 4
#endif
constexpr int WAVES = APS_WAVES;              // waves per workgroup; they split the source tiles of one target tile
constexpr int NTHREADS = TILE * WAVES;
constexpr int MAX_SPLIT = 16;                 // shares a target tile's source list can be cut into
constexpr int PLAN_CAP = 192;                 // planned source tiles per target tile (more -> in-kernel scan)

enum { EV_NONE = 0, EV_LEFT = 1, EV_RIGHT = 2, EV_FWD = 3, EV_BIND = 4, EV_UNBIND = 5, EV_EXIT = 6, EV_FLIP = 7 };
enum { V_FAST = 0, V_GENERIC = 1, V_MIRROR = 2 };

struct PairArgs {
    Model m;
    const uint32_t *src;      // [E][Npad] packed state (epilogue reads the target's own word)
    const uint32_t *orig;     // [E][Npad]
    const long long *gsum;    // [E][2] sum of spins, number alive (global-field mode) of the current state
    long long *gsum_next;     // [E][2] accumulator apply() fills for the next state (cleared by propose())
    const double *beta;       // [E]
    const uint8_t *anchor;    // [L] or nullptr
    uint8_t *prop;            // [world][E][SH]
    uint32_t *pcnt;           // [2][E][L] per-site proposer counters; propose() of step n clears the half of step n+1
    double *accW, *accS;      // [split][E][Npad] per-share partial sums (plain stores; propose() adds the shares)
    unsigned *occ;            // [split][E][Npad] packed neighbour-site occupancies: own | left << 10 | right << 20
    int split;                // shares per target tile
    uint32_t *plist;          // [E][L][2K] per-site proposer lists (used here only when fuse_claim)
    int parity;               // which half of pcnt[2][E][L] belongs to this step
    int fuse_claim;           // single GPU: register the hop proposals right here instead of in claim()
    double *S_out, *W_out;    // optional [E][Npad]
    int *occ4_out;            // optional [E][Npad][4]
    unsigned long long *stamps;   // diagnostic build only (APS_STAMPS): per-workgroup phase cycle totals
    const uint32_t *plan_n;   // [E][ntiles] number of planned source tiles (> PLAN_CAP: overflow)
    int tlen, Npad, SH, E, ntiles, tile_lo, tile_cnt;
    uint32_t step_lo, step_hi;
    int write_prop;
};

// ------------------------------------------------------------------------------------ inner loops
// d = |a - b| + c in ONE VALU op; c carries the LDS byte offset of the table, so the result is directly the
// LDS address of w(|dp|).
__device__ __forceinline__ uint32_t sad3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <bool TAB_LDS>
__device__ __forceinline__ double table_at(const double *__restrict__ table_g, uint32_t byte_addr) {
    if (TAB_LDS) {
        typedef __attribute__((address_space(3))) const double lds_cdouble;
        return *reinterpret_cast<lds_cdouble *>(byte_addr);   // byte_addr already includes the table's LDS offset
    }
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(table_g) + byte_addr);
}

#ifndef APS_G
#define APS_G 8
#endif
#ifndef APS_TA_EVERY
#define APS_TA_EVERY 0                        // 0: all table gathers from LDS
#endif
constexpr int G = APS_G;                      // sources per group = G/4 broadcast ds_read_b128
constexpr int NACC = 2;                       // independent accumulator chains per sum (sums are exact: order-free)
constexpr int RT = 2;                         // target tiles per work item in pair_accumulate (each lane owns RT targets)

// G sources (already broadcast into registers: the same value in every lane) against this lane's target.
// Per pair: v_sad_u32 -> ds_read_b64 (table gather) -> v_add_f64.  The tile's sources are spin-partitioned
// (plus first), so a group usually has ONE sign and adds into a single accumulator set (P = sum over plus
// sources, M = sum over minus sources; W = P + M and S = P - M are exact on the weight grid).  `cut` = number
// of leading plus sources in the group: cut >= G all plus, cut <= 0 all minus, else the mixed group.
//   V_FAST    every pair of the tile block is inside the table's reach and farther than one site: no clamp
//   V_GENERIC clamp to the zero entry behind the table; track the smallest distance for the occupancy branch
//   V_MIRROR  additionally the reflected images (distance p_i + p_j + 1 mod 2L)
template <int BC, int VAR, bool TAB_LDS>
__device__ __forceinline__ void group_weights(const uint32_t (&p8)[G], const uint32_t pi8, const uint32_t tbase,
                                              const double *__restrict__ table_g, const uint32_t tlen8, const uint32_t L8,
                                              double (&wt)[G], int &c0, int &cl, int &cr) {
    uint32_t near = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        if (VAR == V_FAST) {
#if defined(APS_ABL_NOLDS)          /* timing-only ablation (wrong results): no table read at all */
            wt[k] = (double)sad3(pi8, p8[k], tbase);
#else
            wt[k] = table_at<TAB_LDS>(table_g, sad3(pi8, p8[k], tbase));
#endif
        } else {
            const uint32_t d8 = sad3(pi8, p8[k], 0u);
            const uint32_t t8 = (BC == 1) ? min(d8, L8 - d8) : d8;      // circular distance on the torus
            wt[k] = table_at<TAB_LDS>(table_g, min(t8, tlen8) + tbase);
            if (VAR == V_MIRROR) {
                const uint32_t s8 = pi8 + p8[k] + 8u;
                wt[k] += table_at<TAB_LDS>(table_g, min(min(s8, 2u * L8 - s8), tlen8) + tbase);
            }
            near = min(near, t8);
        }
    }
    if (VAR != V_FAST && near <= 8u) {                       // rare: same or neighbouring site -> occupancy
#pragma unroll
        for (int k = 0; k < G; ++k) {
            if (p8[k] == DEAD_P8) continue;
            const int dlt = (int)p8[k] - (int)pi8;
            c0 += (dlt == 0);
            cr += (dlt == 8) | (BC == 1 && dlt == 8 - (int)L8);
            cl += (dlt == -8) | (BC == 1 && dlt == (int)L8 - 8);
        }
    }
}

template <int BC, int VAR, bool TAB_LDS, int R>
__device__ __forceinline__ void group_accumulate(const uint32_t (&p8)[G], const int cut, const uint32_t (&pi8)[R], const uint32_t tbase,
                                                 const double *__restrict__ table_g, const uint32_t tlen8, const uint32_t L8,
                                                 double (&accP)[R][NACC], double (&accM)[R][NACC], int (&cnt)[R][3]) {
    double wt[R][G];
#pragma unroll
    for (int r = 0; r < R; ++r)
        group_weights<BC, VAR, TAB_LDS>(p8, pi8[r], tbase, table_g, tlen8, L8, wt[r], cnt[r][0], cnt[r][1], cnt[r][2]);
    if (cut >= G) {
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r) accP[r][k % NACC] += wt[r][k];
    } else if (cut <= 0) {
#pragma unroll
        for (int k = 0; k < G; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r) accM[r][k % NACC] += wt[r][k];
    } else {                                                 // at most one such group per tile
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const double sp = k < cut ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                accP[r][k % NACC] = fma(wt[r][k], sp, accP[r][k % NACC]);
                accM[r][k % NACC] = fma(wt[r][k], 1.0 - sp, accM[r][k % NACC]);
            }
        }
    }
}

// Which loop variant (or none: -1) does the block (targets in [tlo,thi]) x (source tile info sb) need?
// R = reach of the table in sites (>= 1 so that the occupancy of neighbouring sites is always seen).
// `margin` (sites) makes the answer conservative for states whose tile bounds moved by up to margin/2 each
// since the tile infos were taken (stale-tolerant plans): inclusion tests widen, the FAST test narrows.
template <int BC>
__device__ __forceinline__ int tile_variant(int tlo, int thi, const int4 sb, int R, int Rtab, int L, bool allow_fast,
                                            int margin = 0) {
    if (sb.w <= 0) return -1;                                // no live particle in the tile
    const int slo = sb.x, shi = sb.y;
    if (BC == 0) {
        const int Rm = R + margin;
        const bool direct = (slo <= thi + Rm) && (shi >= tlo - Rm);
        const bool mirror = (tlo + slo + 1 <= Rm) || (2 * L - 1 - thi - shi <= Rm);
        if (mirror) return V_MIRROR;
        if (!direct) return -1;
        const int far = max(thi - slo, shi - tlo);           // largest |p_i - p_j| in the block
        const int gap = slo > thi ? slo - thi : (tlo > shi ? tlo - shi : 0);
        return (allow_fast && far + margin <= Rtab && gap - margin >= 2 && sb.z == 0) ? V_FAST : V_GENERIC;
    }
    const int lo = slo - thi, hi = shi - tlo;                // range of p_j - p_i
    const bool direct = (hi - lo >= L) || (lo <= R && hi >= -R) || (lo - L <= R && hi - L >= -R) ||
                        (lo + L <= R && hi + L >= -R);
    return direct ? V_GENERIC : -1;
}

// One source tile against this lane's target.  `word` is the tile's source word of THIS lane (lane = source:
// site*8 | spin-plus in bit 0, or the far sentinel for dead slots), fetched earlier by one coalesced vector
// load.  The wave parks the 64 sites in its private LDS ring slot -- plus spins first (ballot + mbcnt
// compaction) -- and reads them back 4 at a time with broadcast ds_read_b128, so every lane holds every source.
template <int BC, bool TAB_LDS, int R>
__device__ __forceinline__ void process_tile(const uint32_t var, const uint32_t word, uint32_t *ring, const uint32_t (&pi8)[R],
                                             const uint32_t tbase, const double *__restrict__ table_g, const uint32_t tlen8,
                                             const uint32_t L8, double (&accP)[R][NACC], double (&accM)[R][NACC], int (&cnt)[R][3]) {
    const bool plus = (word & 1u) != 0u;
    const uint64_t pm = __ballot(plus);
    const int nplus = __popcll(pm);
    const int lane = threadIdx.x & 63;
    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
    ring[plus ? below : nplus + (lane - below)] = word & ~7u;
    const uint4 *ring4 = reinterpret_cast<const uint4 *>(ring);
#pragma unroll 1
    for (int g = 0; g < TILE; g += G) {
        uint32_t p8[G];
#pragma unroll
        for (int k = 0; k < G / 4; ++k) {
            const uint4 v = ring4[g / 4 + k];                // uniform address: LDS broadcast
            p8[4 * k] = v.x; p8[4 * k + 1] = v.y; p8[4 * k + 2] = v.z; p8[4 * k + 3] = v.w;
        }
        const int cut = nplus - g;
        if (var == V_FAST) group_accumulate<BC, V_FAST, TAB_LDS, R>(p8, cut, pi8, tbase, table_g, tlen8, L8, accP, accM, cnt);
        else if (var == V_GENERIC) group_accumulate<BC, V_GENERIC, TAB_LDS, R>(p8, cut, pi8, tbase, table_g, tlen8, L8, accP, accM, cnt);
        else group_accumulate<BC, V_MIRROR, TAB_LDS, R>(p8, cut, pi8, tbase, table_g, tlen8, L8, accP, accM, cnt);
    }
}

// Accumulation of one work item by ONE wave: R consecutive target tiles (each lane owns R targets, one per
// tile) against share `q` of `split` of the source tiles of that target group.  Planned path: the wave's list
// entries are fetched with one vector load (lane k = k-th entry of the share) and the source tiles are
// streamed one tile ahead (vmcnt), so no memory latency is exposed in the steady state.  Overflow path (list
// longer than PLAN_CAP, e.g. unsorted particles): the wave scans the tile infos itself, 64 at a time.
// `tb` = merged info of the target group.  No barrier anywhere.
template <int BC, bool TAB_LDS, int R>
__device__ __forceinline__ unsigned accumulate_item(const uint32_t *__restrict__ entries, const int pn, const int q, const int split,
                                                    const uint32_t *__restrict__ sp8_e, const int4 *__restrict__ tinfo_e,
                                                    const int ntiles, const int4 tb, const uint32_t (&pi8)[R], const uint32_t tbase,
                                                    uint32_t *ring, const double *__restrict__ table_g, const int tlen, const int L,
                                                    double (&outW)[R], double (&outS)[R], int (&cnt)[R][3]) {
    const uint32_t tlen8 = (uint32_t)tlen << 3, L8 = (uint32_t)L << 3;
    const int lane = threadIdx.x & 63;
    double accP[R][NACC], accM[R][NACC];                     // sums of weights over plus / minus sources
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < NACC; ++k) accP[r][k] = accM[r][k] = 0.0;
    unsigned done = 0;
    if (pn <= PLAN_CAP) {
        const int n_mine = pn > q ? (pn - q + split - 1) / split : 0;    // entries of this share
        for (int c = 0; c < n_mine; c += TILE) {
            const int n = min(TILE, n_mine - c);
            const uint32_t my_ent = lane < n ? entries[q + (size_t)(c + lane) * split] : 0u;
            uint32_t ent = (uint32_t)__builtin_amdgcn_readlane((int)my_ent, 0);
            uint32_t nxt = sp8_e[(size_t)(ent & 0x0FFFFFFFu) * TILE + lane];
            for (int k = 0; k < n; ++k) {
                const uint32_t cur = nxt, var = ent >> 28;
                if (k + 1 < n) {                             // next tile's sources are in flight during this tile
                    ent = (uint32_t)__builtin_amdgcn_readlane((int)my_ent, k + 1);
                    nxt = sp8_e[(size_t)(ent & 0x0FFFFFFFu) * TILE + lane];
                }
                process_tile<BC, TAB_LDS, R>(var, cur, ring, pi8, tbase, table_g, tlen8, L8, accP, accM, cnt);
                ++done;
            }
        }
    } else {
        const int Rr = tlen > 1 ? tlen - 1 : 1, Rtab = tlen - 1;
        unsigned seen = 0;
        for (int base = 0; base < ntiles; base += 64) {
            const int jt = base + lane;
            const int var = jt < ntiles ? tile_variant<BC>(tb.x, tb.y, tinfo_e[jt], Rr, Rtab, L, tb.z == 0) : -1;
            unsigned long long m = __ballot(var >= 0);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                if ((int)(seen++ % (unsigned)split) != q) continue;
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane(var, b);
                process_tile<BC, TAB_LDS, R>(v, sp8_e[(size_t)(base + b) * TILE + lane], ring, pi8, tbase, table_g, tlen8, L8,
                                             accP, accM, cnt);
                ++done;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 1; k < NACC; ++k) { accP[r][0] += accP[r][k]; accM[r][0] += accM[r][k]; }
        outW[r] = accP[r][0] + accM[r][0];                   // exact on the weight grid
        outS[r] = accP[r][0] - accM[r][0];
    }
    return done;
}

// merged tile info of R consecutive tiles (bounds over the live particles, any-dead flag, live count)
template <int R>
__device__ __forceinline__ int4 merged_info(const int4 *__restrict__ tinfo_e, const int first_tile) {
    int4 m = tinfo_e[first_tile];
#pragma unroll
    for (int r = 1; r < R; ++r) {
        const int4 t = tinfo_e[first_tile + r];
        m.x = min(m.x, t.x); m.y = max(m.y, t.y); m.z |= t.z; m.w += t.w;   // empty tiles carry (INT_MAX, -1)
    }
    return m;
}

// Weight table global -> LDS with 8 independent loads in flight per thread (a plain loop waits for each load).
__device__ __forceinline__ void stage_table(double *lds, const double *__restrict__ table_g, const int tlen) {
    constexpr int U = 8;
    for (int base = threadIdx.x; base <= tlen; base += NTHREADS * U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = base + u * NTHREADS; v[u] = i <= tlen ? table_g[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = base + u * NTHREADS; if (i <= tlen) lds[i] = v[u]; }
    }
}

// LDS holds only the weight table (tlen + 1 doubles) when it fits
__host__ __device__ inline size_t lds_table_bytes(int tlen, bool tab_lds) { return tab_lds ? ((size_t)tlen + 2) / 2 * 2 * sizeof(double) : 0; }
__host__ __device__ inline size_t lds_total_bytes(int tlen, bool tab_lds) { return lds_table_bytes(tlen, tab_lds) + (size_t)WAVES * TILE * sizeof(uint32_t); }

// Epilogue shared by both formulations: m = clip(S/W) -> rates (ref :261-351) -> Philox draw -> proposal byte
// (event code | (free capacity of the hop target - 1) << 3).  c0/cl/cr = occupancy of the own / left / right site.
// The draw itself: x = Philox4x32-10 output of (step, particle id, ensemble) under the handle's key.
__device__ inline uint8_t decide_proposal(const Model &M, bool anch, int p, int spin, bool bound, double mloc, double beta,
                                          int c0, int cl, int cr, const uint32_t (&x)[4]) {
    const Channels c = channels(M, anch, p, spin, bound, mloc, beta, c0, cl, cr);
    const double u0 = ((double)(x[0] >> 5) * 67108864.0 + (double)(x[1] >> 6)) * 0x1.0p-53;
    const double u1 = (double)x[2] * 0x1.0p-32, u2 = (double)x[3] * 0x1.0p-32;
    const double p_fire = 1.0 - aps_exp(-(c.total * M.dt));
    if (!(u0 < p_fire)) return EV_NONE;
    const double v = u1 * c.total;
    const double e_diff = c.diff, e_act = e_diff + c.act, e_bind = e_act + c.bind,
                 e_unbind = e_bind + c.unbind, e_exit = e_unbind + c.leave;
    int ev = EV_NONE, occ_t = 0;
    if (v < e_diff) {
        if (c.left + c.right > 0.0) {
            if (u2 < c.left / (c.left + c.right)) { ev = EV_LEFT; occ_t = cl; }
            else { ev = EV_RIGHT; occ_t = cr; }
        }
    } else if (v < e_act) { ev = EV_FWD; occ_t = cr; }
    else if (v < e_bind) ev = EV_BIND;
    else if (v < e_unbind) ev = EV_UNBIND;
    else if (v < e_exit) ev = EV_EXIT;
    else ev = EV_FLIP;
    int cap = M.K - occ_t;                                   // free capacity of the hop target at step start
    cap = cap < 1 ? 1 : (cap > 32 ? 32 : cap);
    return (uint8_t)(ev | ((cap - 1) << 3));
}

__device__ inline double clip_field(double accS, double accW) {
    double mloc = 0.0;
    if (accW > 0.0) { mloc = accS / accW; mloc = mloc > 1.0 ? 1.0 : (mloc < -1.0 ? -1.0 : mloc); }
    return mloc;
}

__device__ inline uint8_t draw_proposal(const Model &M, bool anch, int p, int spin, bool bound, double accS, double accW,
                                        double beta, int c0, int cl, int cr, uint32_t step_lo, uint32_t step_hi,
                                        uint32_t orig, int ens) {
    uint32_t x[4];
    philox4x32_10(step_lo, step_hi, orig, (uint32_t)ens, M.seed_lo, M.seed_hi, x);
    return decide_proposal(M, anch, p, spin, bound, clip_field(accS, accW), beta, c0, cl, cr, x);
}

// plan<BC>: one wave per target group (RT consecutive tiles) lists the source tiles that can matter (tile index
// | loop variant << 28) into plan[e][group][0..PLAN_CAP) and their number into plan_n[e][group] (a count >
// PLAN_CAP means "scan in-kernel").
// Correct for any particle order; short lists when the slots are site-sorted.
struct PlanArgs { int L, tlen, ntiles, tile_lo, tile_cnt, margin; uint32_t *plan, *plan_n; };

template <int BC>
__global__ __launch_bounds__(256) void plan_tiles(const PlanArgs a, const int4 *__restrict__ tinfo_all) {
    const int e = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int gg = blockIdx.x * 4 + (threadIdx.x >> 6);       // target group within this rank's shard
    if (gg >= a.tile_cnt / RT) return;
    const int group = a.tile_lo / RT + gg;                    // global group index
    const int4 *__restrict__ tinfo_e = tinfo_all + (size_t)e * a.ntiles;
    const int4 tb = merged_info<RT>(tinfo_e, group * RT);
    uint32_t *out = a.plan + ((size_t)e * (a.ntiles / RT) + group) * PLAN_CAP;
    const int R = a.tlen > 1 ? a.tlen - 1 : 1, Rtab = a.tlen - 1;
    unsigned n = 0;
    if (tb.w > 0) {
        constexpr int U = 8;                                  // independent tile-info loads in flight per lane
        for (int base = 0; base < a.ntiles; base += 64 * U) {
            int4 info[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jt = base + u * 64 + lane;
                info[u] = jt < a.ntiles ? tinfo_e[jt] : make_int4(0, -1, 1, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jt = base + u * 64 + lane;
                const int var = jt < a.ntiles ? tile_variant<BC>(tb.x, tb.y, info[u], R, Rtab, a.L, tb.z == 0, a.margin) : -1;
                const unsigned long long m = __ballot(var >= 0);
                if (var >= 0) {
                    const unsigned k = n + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                    if (k < (unsigned)PLAN_CAP) out[k] = (uint32_t)jt | ((uint32_t)var << 28);
                }
                n += (unsigned)__popcll(m);
            }
        }
    }
    if (lane == 0) a.plan_n[(size_t)e * (a.ntiles / RT) + group] = n;
}

// pair_accumulate: the dominant kernel.  Persistent workgroups of 4 INDEPENDENT waves that share only the
// weight table in LDS.  Wave w evaluates the work items w, w + nwaves, w + 2 nwaves, ...  Item = (ensemble, target tile, share q of `split`): 64 targets (lane = target) against every
// split-th source tile of the target tile's list.  The partial sums go to the per-particle accumulators with
// atomics: sums of weights are exact on the 2^-q grid, so the result does not depend on the arrival order.
// The read-only arrays are separate `const __restrict__` kernel parameters (not struct members) so the
// compiler can prove the wave-uniform reads are never clobbered and emits scalar loads.
template <int BC, bool TAB_LDS>
__global__ __launch_bounds__(NTHREADS) void pair_accumulate(const PairArgs a, const uint32_t *__restrict__ sp8_all,
                                                           const int4 *__restrict__ tinfo_all,
                                                           const double *__restrict__ table_g,
                                                           const uint32_t *__restrict__ plan_all,
                                                           const uint32_t *__restrict__ plan_n_all) {
    extern __shared__ double lds[];
    const Model &M = a.m;
    const int lane = threadIdx.x & (TILE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t tbase = 0;
    if (TAB_LDS) {
        stage_table(lds, table_g, a.tlen);
        typedef __attribute__((address_space(3))) double lds_double;
        tbase = (uint32_t)(size_t)(lds_double *)lds;          // LDS byte offset of the table
        __syncthreads();
    }
    uint32_t *ring = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + lds_table_bytes(a.tlen, TAB_LDS)) + wave * TILE;
    const unsigned ngroups = (unsigned)(a.tile_cnt / RT);     // target groups of this rank's shard
    const unsigned total_items = (unsigned)a.E * ngroups * (unsigned)a.split;
    const unsigned nwaves = gridDim.x * WAVES;
#ifdef APS_STAMPS
    unsigned long long t_fetch = 0, t_acc = 0, t_epi = 0, t_items = 0, t0 = __builtin_amdgcn_s_memtime(), t_start = t0;
    const unsigned long long r_start = __builtin_amdgcn_s_memrealtime();
#define STAMP(var) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); var += t1_ - t0; t0 = t1_; }
#else
#define STAMP(var)
#endif
    unsigned item = blockIdx.x * WAVES + wave;                // static striding: wave w takes items w, w + nwaves, ...
    while (item < total_items) {
        const int q = (int)(item % (unsigned)a.split);
        const unsigned gix = item / (unsigned)a.split;
        const int e = (int)(gix / ngroups);
        const int group = a.tile_lo / RT + (int)(gix % ngroups);
        const uint32_t *__restrict__ sp8_e = sp8_all + (size_t)e * a.Npad;
        const int4 *__restrict__ tinfo_e = tinfo_all + (size_t)e * a.ntiles;
        const size_t slot0 = (size_t)group * RT * TILE + lane;
        uint32_t pi8[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const uint32_t w = sp8_e[slot0 + (size_t)r * TILE] & ~7u;
            pi8[r] = w == DEAD_P8 ? 0u : w;                  // dead targets: any in-range site
        }
        const int4 tb = merged_info<RT>(tinfo_e, group * RT);
        const size_t pidx = (size_t)e * (a.ntiles / RT) + group;
        const int pn = (int)plan_n_all[pidx];
        STAMP(t_fetch)
        double accW[RT], accS[RT];
        int cnt[RT][3];
#pragma unroll
        for (int r = 0; r < RT; ++r) { accW[r] = accS[r] = 0.0; cnt[r][0] = cnt[r][1] = cnt[r][2] = 0; }
        if (tb.w > 0)
            accumulate_item<BC, TAB_LDS, RT>(plan_all + pidx * PLAN_CAP, pn, q, a.split, sp8_e, tinfo_e, a.ntiles, tb, pi8, tbase,
                                             ring, table_g, a.tlen, M.L, accW, accS, cnt);
        STAMP(t_acc)
#ifdef APS_ABL_NOSTORE
        if (accW[0] == -1.0)
#endif
#pragma unroll
        for (int r = 0; r < RT; ++r) {   // every (group, share) item is evaluated exactly once per launch: plain stores
            const size_t o = ((size_t)q * a.E + e) * a.Npad + slot0 + (size_t)r * TILE;
            a.accW[o] = accW[r]; a.accS[o] = accS[r];
            a.occ[o] = (unsigned)cnt[r][0] | ((unsigned)cnt[r][1] << 10) | ((unsigned)cnt[r][2] << 20);
        }
        item += nwaves;                                       // items cost about the same; a single dequeue counter
        STAMP(t_epi)                                          // would saturate at ~88 grabs/us (measured)
#ifdef APS_STAMPS
        t_items += 1;
#endif
    }
#ifdef APS_STAMPS
    if (threadIdx.x == 0 && a.stamps) {
        unsigned long long *o = a.stamps + (size_t)blockIdx.x * 8;
        o[0] = t_fetch; o[1] = t_acc; o[2] = t_epi; o[3] = t_items; o[4] = __builtin_amdgcn_s_memtime() - t_start;
        o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = r_start;
    }
#endif
}

// propose: per-particle epilogue of the step (rates -> Philox draw -> proposal byte), one thread per slot of
// this rank's shard.  Consumes and clears the accumulators; also clears the per-site proposer counters for
// the commit that follows and resets the work counter.
__global__ __launch_bounds__(256) void propose(const PairArgs a) {
    const Model &M = a.m;
    const int e = blockIdx.y;
    {   // the counters of the NEXT step (other parity) are idle now: clear them
        uint32_t *other = a.pcnt + ((size_t)(a.parity ^ 1) * a.E + e) * M.L;
        const size_t total = (size_t)M.L, nthreads = (size_t)gridDim.x * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += nthreads) other[i] = 0u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.gsum_next[2 * e] = 0; a.gsum_next[2 * e + 1] = 0; }   // apply() accumulates
    const size_t slot = (size_t)a.tile_lo * TILE + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (size_t)(a.tile_lo + a.tile_cnt) * TILE) return;
    const size_t o = (size_t)e * a.Npad + slot;
    double accW = 0.0, accS = 0.0;
    unsigned packed = 0;
    for (int q = 0; q < a.split; ++q) {                       // exact sums: the order of the shares is irrelevant
        const size_t oq = ((size_t)q * a.E + e) * a.Npad + slot;
        accW += a.accW[oq]; accS += a.accS[oq]; packed += a.occ[oq];
    }
    int c0 = packed & 1023, cl = (packed >> 10) & 1023, cr = (packed >> 20) & 1023;
    const uint32_t me = a.src[o];
    const bool live = !(me & DEAD_BIT);
    const int p = (int)(me & POS_MASK);
    const int spin = (me & SPIN_BIT) ? 1 : -1;
    const bool bound = (me & BOUND_BIT) != 0;
    if (M.field_mode == 0) {                                  // global mean field (ref :219-221)
        accS = (double)a.gsum[2 * e]; accW = (double)a.gsum[2 * e + 1];
    }
    if (!live) { accS = 0.0; accW = 0.0; c0 = cl = cr = 0; }
    const bool wall_l = !M.periodic && p == 0, wall_r = !M.periodic && p == M.L - 1;
    if (a.S_out) {
        a.S_out[o] = accS;
        a.W_out[o] = accW;
        int *oo = a.occ4_out + o * 4;
        const int o_l = wall_l ? c0 : cl, o_r = wall_r ? c0 : cr;
        oo[0] = c0; oo[1] = live ? (spin > 0 ? o_r : c0) : 0; oo[2] = live ? o_l : 0; oo[3] = live ? o_r : 0;
    }
    if (!a.write_prop) return;
    uint8_t code = EV_NONE;
    if (live)
        code = draw_proposal(M, a.anchor ? a.anchor[p] != 0 : false, p, spin, bound, accS, accW, a.beta[e], c0, cl, cr,
                             a.step_lo, a.step_hi, a.orig[o], M.ens_base + e);
    const int r = (int)(slot / a.SH);
    a.prop[((size_t)r * a.E + e) * a.SH + (slot - (size_t)r * a.SH)] = code;
    if (a.fuse_claim) {                                       // same as claim(): join the target site's proposer list
        const int ev = code & 7;
        if (ev == EV_LEFT || ev == EV_RIGHT || ev == EV_FWD) {
            int s = (ev == EV_LEFT) ? p - 1 : p + 1;
            if (M.periodic) s = s < 0 ? s + M.L : (s >= M.L ? s - M.L : s);
            const size_t site = (size_t)e * M.L + s;
            const uint32_t k = atomicAdd(&a.pcnt[(size_t)a.parity * a.E * M.L + site], 1u);
            if (k < 2u * M.K) a.plist[site * 2 * M.K + k] = a.orig[o];
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct CommitArgs {
    Model m;
    uint32_t *src; const uint32_t *orig; const uint8_t *prop;
    uint32_t *sp8; int4 *tinfo;
    uint32_t *pcnt, *plist; long long *gsum;
    double *exit_log; unsigned *n_exit; int exit_cap;
    int Npad, SH, E, ntiles, parity;
    double step_as_double;
    // lattice formulation (all null / 0 in the all-pairs formulation)
    const unsigned long long *stepw;     // device step words: the step index is stepw[parity] (graph replays)
    uint32_t *occ_site;                  // [E][L] particles per site
    uint32_t *dcnt, *dep;                // [E][nb] deposit counters, [E][nb][dcap] deposits of this step
    int bshift, nb, dcap;
};

// A deposit = one change of the lattice field caused by an accepted event: add cW * w(x - site) to W(x) and
// cS * w(x - site) to S(x) for every site x in reach.  hop of a spin-s particle: (-1, -s) at the old site and
// (+1, +s) at the new one; flip s -> -s: (0, -2s); exit: (-1, -s).  Packed: site | (cW + 1) << 27 | (cS + 2) << 29.
__device__ __host__ inline uint32_t deposit(int site, int cw, int cs) {
    return (uint32_t)site | ((uint32_t)(cw + 1) << 27) | ((uint32_t)(cs + 2) << 29);
}
constexpr uint32_t DEP_NULL = (1u << 27) | (2u << 29);        // cW = cS = 0: contributes nothing

__device__ inline int hop_target(const Model &M, int p, int ev) {
    int s = (ev == EV_LEFT) ? p - 1 : p + 1;
    if (M.periodic) s = s < 0 ? s + M.L : (s >= M.L ? s - M.L : s);
    return s;
}

__global__ __launch_bounds__(256) void claim(const CommitArgs a) {
    const int e = blockIdx.y;
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (size_t)a.Npad) return;
    const int r = (int)(slot / a.SH);
    const uint8_t code = a.prop[((size_t)r * a.E + e) * a.SH + (slot - (size_t)r * a.SH)];
    const int ev = code & 7;
    if (ev != EV_LEFT && ev != EV_RIGHT && ev != EV_FWD) return;
    const uint32_t me = a.src[(size_t)e * a.Npad + slot];
    const int s = hop_target(a.m, (int)(me & POS_MASK), ev);
    const size_t site = (size_t)e * a.m.L + s;
    const uint32_t k = atomicAdd(&a.pcnt[(size_t)a.parity * a.E * a.m.L + site], 1u);
    if (k < 2u * a.m.K) a.plist[site * 2 * a.m.K + k] = a.orig[(size_t)e * a.Npad + slot];
}

__global__ __launch_bounds__(256) void apply(const CommitArgs a) {
    const int e = blockIdx.y;
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // Npad is a multiple of 256
    const int r = (int)(slot / a.SH);
    const uint8_t code = a.prop[((size_t)r * a.E + e) * a.SH + (slot - (size_t)r * a.SH)];
    const int ev = code & 7;
    uint32_t me = a.src[(size_t)e * a.Npad + slot];
    const uint32_t my_orig = a.orig[(size_t)e * a.Npad + slot];
    int p = (int)(me & POS_MASK);
    if (!(me & DEAD_BIT)) {
        const int p_old = p, sgn = (me & SPIN_BIT) ? 1 : -1;
        uint32_t d0 = 0, d1 = 0;                              // lattice deposits of this particle
        int nd = 0;
        if (ev == EV_LEFT || ev == EV_RIGHT || ev == EV_FWD) {
            const int s = hop_target(a.m, p, ev);
            const size_t site = (size_t)e * a.m.L + s;
            const uint32_t n = min(a.pcnt[(size_t)a.parity * a.E * a.m.L + site], 2u * a.m.K);
            int rank = 0;                                     // proposers to s with a smaller particle index
            for (uint32_t k = 0; k < n; ++k) rank += a.plist[site * 2 * a.m.K + k] < my_orig;
            if (rank < (code >> 3) + 1) {
                p = s; me = (me & ~POS_MASK) | (uint32_t)s;
                if (a.occ_site) {
                    atomicAdd(&a.occ_site[site], 1u);
                    atomicSub(&a.occ_site[(size_t)e * a.m.L + p_old], 1u);
                    d0 = deposit(p_old, -1, -sgn); d1 = deposit(s, 1, sgn); nd = 2;
                }
            }
        } else if (ev == EV_BIND) me |= BOUND_BIT;
        else if (ev == EV_UNBIND) me &= ~BOUND_BIT;
        else if (ev == EV_FLIP) { me ^= SPIN_BIT; d0 = deposit(p, 0, -2 * sgn); nd = 1; }
        else if (ev == EV_EXIT) {
            me |= DEAD_BIT;
            const unsigned k = atomicAdd(&a.n_exit[e], 1u);
            if ((int)k < a.exit_cap) {
                double *row = a.exit_log + ((size_t)e * a.exit_cap + k) * 3;
                row[0] = a.stepw ? (double)a.stepw[a.parity] : a.step_as_double; row[1] = (double)p; row[2] = (double)my_orig;
            }
            if (a.occ_site) atomicSub(&a.occ_site[(size_t)e * a.m.L + p], 1u);
            d0 = deposit(p, -1, -sgn); nd = 1;
        }
        if (ev != EV_NONE) a.src[(size_t)e * a.Npad + slot] = me;
        if (a.dep && nd) {                                    // binned by the OLD site: <= 2 per particle, so a bucket
            const size_t b = (size_t)e * a.nb + (size_t)(p_old >> a.bshift);   // of B sites never exceeds 2 K B
            const uint32_t k = atomicAdd(&a.dcnt[b], (uint32_t)nd);
            if (k + (uint32_t)nd <= (uint32_t)a.dcap) {
                a.dep[b * a.dcap + k] = d0;
                if (nd == 2) a.dep[b * a.dcap + k + 1] = d1;
            }
        }
    }
    // pre-decoded source pair for the next all-pairs pass, per-tile (= per-wave) info, global spin sums
    const bool live = !(me & DEAD_BIT);
    if (ev != EV_NONE) a.sp8[(size_t)e * a.Npad + slot] = live ? (((uint32_t)p << 3) | ((me & SPIN_BIT) ? 1u : 0u)) : DEAD_P8;
    int lo = live ? p : 0x7fffffff, hi = live ? p : -1;
    int ssum = live ? ((me & SPIN_BIT) ? 1 : -1) : 0, cnt = live ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off));
        ssum += __shfl_xor(ssum, off); cnt += __shfl_xor(cnt, off);
    }
    if ((threadIdx.x & 63) == 0) {
        a.tinfo[(size_t)e * a.ntiles + slot / TILE] = make_int4(lo, hi, cnt < TILE ? 1 : 0, cnt);
        if (a.m.field_mode == 0 && cnt) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&a.gsum[2 * e]), (unsigned long long)(long long)ssum);
            atomicAdd(reinterpret_cast<unsigned long long *>(&a.gsum[2 * e + 1]), (unsigned long long)cnt);
        }
    }
}

// ================================================================================ lattice formulation
// The reference's own algorithm (histogram -> Gaussian smoothing -> read back at the particle sites, ref
// :216-246, :261-301), kept INCREMENTALLY: the smoothed histograms W(x) = tot_conv, S(x) = s_conv live on the L
// sites and every accepted event adds its change (a "deposit") to the sites in reach.  All values sit on the
// weight grid 2^-q, so add/subtract sequences are exact and W, S equal a from-scratch recomputation bit for
// bit after any number of steps (tests compare with the oracle's recomputation and with the all-pairs kernel).
// Per step:  propose_lattice (gather W,S,occupancy at the particle -> rates -> Philox -> proposal)
//            [all-gather + claim when sharded]  apply (commit; occupancy and deposits)  field_update.
struct LatticeArgs {
    Model m;
    const uint32_t *src, *orig;
    const double2 *ws;                   // [E][L] {W, S}
    const uint32_t *occ_site;            // [E][L]
    const long long *gsum; long long *gsum_next;
    const double *beta; const uint8_t *anchor;
    uint8_t *prop; uint32_t *pcnt, *plist, *dcnt;
    unsigned long long *stepw;           // [2]: step n reads stepw[n & 1] and writes n + 1 to the other word
    double *S_out, *W_out; int *occ4_out;
    int par, fuse_claim, write_prop, Npad, SH, E, tile_lo, tile_cnt, nb;
};

__global__ __launch_bounds__(256) void propose_lattice(const LatticeArgs a) {
    const Model &M = a.m;
    const int e = blockIdx.y;
    const unsigned long long step = a.stepw[a.par];
    if (a.write_prop) {   // idle buffers of the coming commit: the other parity's site counters, the deposit counters
        uint32_t *other = a.pcnt + ((size_t)(a.par ^ 1) * a.E + e) * M.L;
        const size_t nthreads = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (size_t i = t0; i < (size_t)M.L; i += nthreads) other[i] = 0u;
        if (a.dcnt) for (size_t i = t0; i < (size_t)a.nb; i += nthreads) a.dcnt[(size_t)e * a.nb + i] = 0u;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.gsum_next[2 * e] = 0; a.gsum_next[2 * e + 1] = 0;
            if (e == 0) a.stepw[a.par ^ 1] = step + 1ull;     // nobody reads that word during this step
        }
    }
    const size_t slot = (size_t)a.tile_lo * TILE + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= (size_t)(a.tile_lo + a.tile_cnt) * TILE) return;
    const size_t o = (size_t)e * a.Npad + slot;
    const uint32_t me = a.src[o];
    const bool live = !(me & DEAD_BIT);
    const int p = (int)(me & POS_MASK), L = M.L;
    const int spin = (me & SPIN_BIT) ? 1 : -1;
    const bool bound = (me & BOUND_BIT) != 0;
    double accW = 0.0, accS = 0.0;
    int c0 = 0, cl = 0, cr = 0;
    if (live) {
        const uint32_t *occ = a.occ_site + (size_t)e * L;
        int l = p - 1, r = p + 1;
        if (M.periodic) { l = l < 0 ? l + L : l; r = r >= L ? r - L : r; }
        c0 = (int)occ[p];
        cl = l >= 0 ? (int)occ[l] : 0;
        cr = r < L ? (int)occ[r] : 0;
        if (M.field_mode == 0) { accS = (double)a.gsum[2 * e]; accW = (double)a.gsum[2 * e + 1]; }
        else { const double2 f = a.ws[(size_t)e * L + p]; accW = f.x; accS = f.y; }
    }
    const bool wall_l = !M.periodic && p == 0, wall_r = !M.periodic && p == L - 1;
    if (a.S_out) {
        a.S_out[o] = accS;
        a.W_out[o] = accW;
        int *oo = a.occ4_out + o * 4;
        const int o_l = wall_l ? c0 : cl, o_r = wall_r ? c0 : cr;
        oo[0] = c0; oo[1] = live ? (spin > 0 ? o_r : c0) : 0; oo[2] = live ? o_l : 0; oo[3] = live ? o_r : 0;
    }
    if (!a.write_prop) return;
    uint8_t code = EV_NONE;
    if (live)
        code = draw_proposal(M, a.anchor ? a.anchor[p] != 0 : false, p, spin, bound, accS, accW, a.beta[e], c0, cl, cr,
                             (uint32_t)step, (uint32_t)(step >> 32), a.orig[o], M.ens_base + e);
    const int r = (int)(slot / a.SH);
    a.prop[((size_t)r * a.E + e) * a.SH + (slot - (size_t)r * a.SH)] = code;
    if (a.fuse_claim) {                                       // same as claim(): join the target site's proposer list
        const int ev = code & 7;
        if (ev == EV_LEFT || ev == EV_RIGHT || ev == EV_FWD) {
            int s = (ev == EV_LEFT) ? p - 1 : p + 1;
            if (M.periodic) s = s < 0 ? s + L : (s >= L ? s - L : s);
            const size_t site = (size_t)e * L + s;
            const uint32_t k = atomicAdd(&a.pcnt[(size_t)a.par * a.E * L + site], 1u);
            if (k < 2u * M.K) a.plist[site * 2 * M.K + k] = a.orig[o];
        }
    }
}

// field_update: one workgroup owns SITES = 64 * RS consecutive sites and adds every deposit in reach to them.
// The deposits sit in per-bucket lists (bucket = B consecutive sites of the depositing particle's old position).
// The four waves SPLIT THE BUCKETS in reach (dealt round-robin): a wave fetches 8 list slots of 8 buckets per load
// (lane = (bucket, slot)), together with the buckets' counters, compacts the valid ones into its own LDS segment (ballot + mbcnt)
// and sweeps ALL the workgroup's sites with them (lane = RS sites, 64 apart).  A deposit is wave-uniform (scalar
// registers), so the 64 lanes read 64 CONSECUTIVE table entries -- a conflict-free ds_read_b64 -- and issue two f64
// fma's (W and S); the LDS copy of the table is zero-padded so that interior workgroups need no range clamp.  The
// four partial sums per site are added through LDS (exact on the weight grid, any order) and each site is
// updated by exactly one thread: plain read-modify-write, no atomics.  Counters, list slots, table and the sites'
// old values are all requested up front: the kernel pays ONE memory round trip before it computes.
constexpr int FU_THREADS = 256, FU_WAVES = FU_THREADS / 64, FU_SEG = 256, FU_SEG_WIN = 128, FU_WIN_BUCKETS = 16, FU_TREG = 16, FU_PRE = 2;
struct FieldUpdArgs { int L, tlen, bshift, nb, dcap; double2 *ws; const uint32_t *dcnt, *dep; unsigned long long *stamps; };

__host__ __device__ inline int fu_table_pad(int RS, int bshift) { return 64 * RS + (1 << bshift) + 2; }
// table beyond LDS: entries of one window (the distances a group of FU_WIN_BUCKETS buckets can have to the tile), in
// chunks of 128 entries = one LDS-direct load of a wave
__host__ __device__ inline int fu_win_entries(int RS, int bshift) { return ((FU_WIN_BUCKETS << bshift) + 64 * RS + 8 + 127) / 128 * 128; }
__host__ __device__ inline size_t fu_lds_bytes(int tlen, bool tab_lds, int RS, int bshift) {
    // the whole (padded) table when it fits; otherwise two windows of it (double buffer, interior tiles)
    const size_t table = tab_lds ? ((size_t)tlen + 2 + fu_table_pad(RS, bshift)) / 2 * 2 * sizeof(double)
                                 : (size_t)2 * fu_win_entries(RS, bshift) * sizeof(double);
    const size_t red = (size_t)FU_WAVES * 64 * RS * sizeof(double2);          // reuses the table's space after the sweep
    return (size_t)FU_WAVES * 2 * ((tab_lds ? FU_SEG : FU_SEG_WIN) + 4) * sizeof(uint32_t) + (table > red ? table : red);
}

// VAR: 0 = interior (no image term, padded table: no clamp), 1 = torus, 2 = reflecting wall in reach
template <int VAR, bool TAB_LDS, int RS>
__device__ __forceinline__ void fu_group(const uint4 q, const uint32_t (&x8)[RS], const uint32_t tbase, const double *__restrict__ table_g,
                                         const uint32_t tlen8, const uint32_t L8, double (&accW)[RS], double (&accS)[RS]) {
    const uint32_t ent[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)q.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.y),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)q.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)q.w)};
    double w[4][RS];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                            // wave-uniform deposit: decode on the scalar unit
        const uint32_t p8 = (ent[k] & POS_MASK) << 3;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            if (VAR == 0) {
                const uint32_t d = sad3(x8[r], p8, tbase);
                w[k][r] = table_at<TAB_LDS>(table_g, TAB_LDS ? d : min(d, tlen8));
            } else if (VAR == 1) {
                const uint32_t d8 = sad3(x8[r], p8, 0u);
                w[k][r] = table_at<TAB_LDS>(table_g, min(min(d8, L8 - d8), tlen8) + tbase);
            } else {
                const uint32_t s8 = x8[r] + p8 + 8u;
                w[k][r] = table_at<TAB_LDS>(table_g, min(sad3(x8[r], p8, 0u), tlen8) + tbase) +
                          table_at<TAB_LDS>(table_g, min(min(s8, 2u * L8 - s8), tlen8) + tbase);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double cw = (double)((int)((ent[k] >> 27) & 3u) - 1), cs = (double)((int)(ent[k] >> 29) - 2);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            accW[r] = fma(w[k][r], cw, accW[r]);             // cw, cs in {-2..2}: exact on the weight grid
            accS[r] = fma(w[k][r], cs, accS[r]);
        }
    }
}

template <int BC, bool TAB_LDS, int RS>
__global__ __launch_bounds__(FU_THREADS) void field_update(const FieldUpdArgs a, const double *__restrict__ table_g) {
    constexpr int SITES = 64 * RS, NOLD = (SITES + FU_THREADS - 1) / FU_THREADS;
    extern __shared__ double lds[];
    const int tpad = TAB_LDS ? a.tlen + fu_table_pad(RS, a.bshift) : 0;        // last LDS table index (zeros beyond tlen - 1)
    constexpr int SEG = TAB_LDS ? FU_SEG : FU_SEG_WIN;
    uint32_t *seg_all = reinterpret_cast<uint32_t *>(lds);                   // [FU_WAVES][2][SEG + 4] deposits, then the table
    double *tab = reinterpret_cast<double *>(seg_all + FU_WAVES * 2 * (SEG + 4));
    double2 *red = reinterpret_cast<double2 *>(tab);          // [FU_WAVES][SITES] partial sums, over the table once it is no longer read
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), e = blockIdx.y;
    uint32_t *seg = seg_all + wave * 2 * (SEG + 4), *segi = seg + SEG + 4;   // plain deposits / deposits with an image term
    // table beyond LDS, reflecting walls: the tiles next to a wall gather from global memory and carry image deposits --
    // several times the work of an interior tile -- so they are dealt first, from both ends inwards
    const int tile = (BC == 0 && !TAB_LDS) ? ((blockIdx.x & 1) ? (int)gridDim.x - 1 - (int)(blockIdx.x >> 1) : (int)(blockIdx.x >> 1)) : (int)blockIdx.x;
    const int L = a.L, x0 = tile * SITES, x1 = min(x0 + SITES - 1, L - 1);
    const int Rt = a.tlen - 1;                                // largest distance with a non-zero weight
#ifdef APS_STAMPS
    unsigned long long f_cnt = 0, f_stage = 0, f_copy = 0, f_proc = 0, f_n = 0, t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long f_start = t0, r_start = __builtin_amdgcn_s_memrealtime();
#define FSTAMP(var) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); var += t1_ - t0; t0 = t1_; }
#else
#define FSTAMP(var)
#endif
    // buckets whose deposits can reach [x0, x1] (deposit sites lie within one site of their bucket), as one run of
    // `nbk` buckets starting at `b0` that may wrap around the torus
    int b0, nbk;
    if (BC == 0) {
        b0 = max(0, x0 - Rt - 1) >> a.bshift;
        nbk = (min(L - 1, x1 + Rt + 1) >> a.bshift) - b0 + 1;
    } else {
        const int lo = x0 - Rt - 1, hi = x1 + Rt + 1;
        if (hi - lo + 1 >= L) { b0 = 0; nbk = a.nb; }
        else {
            const int blo = (lo < 0 ? lo + L : lo) >> a.bshift, bhi = (hi >= L ? hi - L : hi) >> a.bshift;
            b0 = blo;
            nbk = (lo >= 0 && hi < L) ? bhi - blo + 1 : (a.nb - blo) + bhi + 1;
            if (nbk > a.nb) { b0 = 0; nbk = a.nb; }
        }
    }
    const bool wall = BC == 0 && ((x0 + 1 <= Rt) || (L - x1 <= Rt));   // an image term can be non-zero
    // Requests that depend on nothing: counters and first list slots of the first FU_PRE bucket groups (a group =
    // 32 buckets, 8 per wave, 8 slots each), the table, the sites' old values.
    // lane = (bucket, list slot) of a load: 8 buckets x 8 slots per wave (groups of 32 buckets); with the table windowed,
    // 4 buckets x 16 slots (groups of 16: half the window, so more workgroups per CU overlap staging and sweeping)
    constexpr int SB = TAB_LDS ? 3 : 4, NSLOT = 1 << SB, GB = FU_WAVES * (64 >> SB);
    static_assert(TAB_LDS || GB == FU_WIN_BUCKETS, "window size");
    const int sub = lane >> SB, slot = lane & (NSLOT - 1);
    uint32_t pre_cnt[FU_PRE], pre_ent[FU_PRE];
#pragma unroll
    for (int j = 0; j < FU_PRE; ++j) {
        const int bi = j * GB + sub * FU_WAVES + wave;          // buckets dealt to the waves round-robin: near-wall ones cost more
        int b = b0 + bi;
        if (b >= a.nb) b -= a.nb;
        const bool ok = bi < nbk;
        pre_cnt[j] = ok ? a.dcnt[(size_t)e * a.nb + b] : 0u;
        pre_ent[j] = (ok && slot < a.dcap) ? a.dep[((size_t)e * a.nb + b) * a.dcap + slot] : DEP_NULL;
    }
    double tv[FU_TREG];
    if (TAB_LDS) {
#pragma unroll
        for (int u = 0; u < FU_TREG; ++u) { const int i = t + u * FU_THREADS; tv[u] = i < a.tlen ? table_g[i] : 0.0; }
    }
    double2 old[NOLD];
#pragma unroll
    for (int r = 0; r < NOLD; ++r) {
        const int xi = r * FU_THREADS + t, x = x0 + xi;
        old[r] = (xi < SITES && x < L) ? a.ws[(size_t)e * L + x] : make_double2(0.0, 0.0);
    }
    uint32_t tbase = 0;
    if (TAB_LDS) {
        typedef __attribute__((address_space(3))) double lds_double;
        tbase = (uint32_t)(size_t)(lds_double *)tab;
#pragma unroll
        for (int u = 0; u < FU_TREG; ++u) { const int i = t + u * FU_THREADS; if (i <= tpad) tab[i] = tv[u]; }
        for (int i = t + FU_TREG * FU_THREADS; i <= tpad; i += FU_THREADS) tab[i] = i < a.tlen ? table_g[i] : 0.0;   // beyond 4096 entries
    }
    const uint32_t tlen8 = (uint32_t)a.tlen << 3, L8 = (uint32_t)L << 3;
    uint32_t x8[RS];
    double accW[RS], accS[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) { x8[r] = (uint32_t)min(x0 + r * 64 + lane, L - 1) << 3; accW[r] = accS[r] = 0.0; }
    __syncthreads();                                          // table staged
    FSTAMP(f_stage)
    // Table too large for LDS: interior tiles keep, per group of GB = 16 buckets, the window of the table that the group's
    // distances to this tile can touch (<= GB B + tile + 2 entries) in LDS and gather from that -- the sweep then reads LDS
    // instead of sending 512 bytes per (deposit, 64-site tile) to L2.  The windows are double buffered and filled by
    // LDS-direct loads (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no registers), issued one group
    // ahead, so the sweep of group j hides the round trip of window j + 1.  The table in global memory is followed by
    // zeros (see the allocation), so a window may run past its end.  Wall tiles and the torus keep the global path.
    const bool windowed = !TAB_LDS && BC == 0 && !wall;
    uint32_t win_base = 0;
    if (!TAB_LDS) {
        typedef __attribute__((address_space(3))) double lds_double;
        win_base = (uint32_t)(size_t)(lds_double *)tab;
    }
    const uint32_t win_lds = win_base;
    const int WIN = fu_win_entries(RS, a.bshift);
    int null_site = x0;                                       // site of the padding deposits (coefficients 0)
    int dmin_next = 0, null_next = x0;
    auto stage = [&](int j) {                                 // request window j into buffer j & 1
        const int bs = b0 + j * GB, be = min(bs + GB, b0 + nbk);                        // buckets [bs, be) of this group
        const int sA = max(0, (bs << a.bshift) - 1), sB = min(L - 1, be << a.bshift);   // their deposits' sites (one beyond each end)
        const int dmin = max(0, max(x0 - sB, sA - x1)), dmax = max(x1 - sA, sB - x0), span = dmax - dmin;
        // Written as inline assembly on purpose: the compiler orders every later LDS read behind a tracked LDS-direct load
        // (s_waitcnt vmcnt(0) in front of the first ds_read of the sweep), which would serialise exactly what this hides.
        // The explicit s_waitcnt vmcnt(0) + barrier at the top of the next group is the ordering that is needed.
        const double *srcw = table_g + dmin + lane * 2;
        const uint32_t dstw = win_lds + (uint32_t)((j & 1) * WIN) * 8u;
        for (int c = wave; c * 128 <= span; c += FU_WAVES) {
            const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dstw + (uint32_t)c * 1024u));
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(srcw + c * 128) : "memory");
        }
        dmin_next = dmin;
        null_next = min(max(x0, sA), sB);                     // its distances to the tile lie inside [dmin, dmax]
    };
    int nseg = 0, nimg = 0;                                   // entries waiting in this wave's two segments
    const uint4 *seg4 = reinterpret_cast<const uint4 *>(seg), *segi4 = reinterpret_cast<const uint4 *>(segi);
    auto flush = [&]() {                                      // sweep all SITES with the segments' deposits
        if (lane < 4) { seg[nseg + lane] = DEP_NULL | (uint32_t)null_site; segi[nimg + lane] = DEP_NULL | (uint32_t)x0; }   // pad to groups of four (a site whose distances stay in table range)
        const int n4 = (nseg + 3) >> 2, ni4 = (nimg + 3) >> 2;
#pragma unroll 1
        for (int i = 0; i < n4; ++i) {
            const uint4 q = seg4[i];                          // uniform address: LDS broadcast
            if (BC == 1) fu_group<1, TAB_LDS, RS>(q, x8, tbase, table_g, tlen8, L8, accW, accS);
            else if (!TAB_LDS && windowed) fu_group<0, true, RS>(q, x8, win_base, table_g, tlen8, L8, accW, accS);
            else fu_group<0, TAB_LDS, RS>(q, x8, tbase, table_g, tlen8, L8, accW, accS);
        }
#pragma unroll 1
        for (int i = 0; i < ni4; ++i) fu_group<2, TAB_LDS, RS>(segi4[i], x8, tbase, table_g, tlen8, L8, accW, accS);
#ifdef APS_STAMPS
        f_n += nseg + nimg;
#endif
        nseg = nimg = 0;
    };
    const int ngroups = (nbk + GB - 1) / GB;
    if (!TAB_LDS && windowed) stage(0);
    uint32_t nx_cnt = 0u, nx_ent = DEP_NULL;
    for (int j = 0; j < ngroups; ++j) {
        const int bi = j * GB + sub * FU_WAVES + wave;          // buckets dealt to the waves round-robin: near-wall ones cost more
        int b = b0 + bi;
        if (b >= a.nb) b -= a.nb;
        const bool ok = bi < nbk;
        uint32_t cnt, ent;                                    // requested before the window is staged: one round trip for both
        if (j < FU_PRE) { cnt = j == 0 ? pre_cnt[0] : pre_cnt[FU_PRE - 1]; ent = j == 0 ? pre_ent[0] : pre_ent[FU_PRE - 1]; }
        else if (!TAB_LDS) { cnt = nx_cnt; ent = nx_ent; }    // requested during the previous group's sweep
        else {
            cnt = ok ? a.dcnt[(size_t)e * a.nb + b] : 0u;
            ent = (ok && slot < a.dcap) ? a.dep[((size_t)e * a.nb + b) * a.dcap + slot] : DEP_NULL;
        }
        if (!TAB_LDS) {
            FSTAMP(f_copy)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's chunks of window j have landed (and the list loads)
            asm volatile("" : "+v"(cnt), "+v"(ent));          // the list words count as arrived: no wait on them once window j + 1 is in flight
        }
        if (!TAB_LDS && windowed) {
            __syncthreads();                                  // ... everyone's; and every wave is done sweeping window j - 1
            win_base = win_lds + (uint32_t)((j & 1) * WIN) * 8u - ((uint32_t)dmin_next << 3);
            null_site = null_next;
            if (j + 1 < ngroups) stage(j + 1);                // lands during this group's sweep
            FSTAMP(f_cnt)
        }
        if (!TAB_LDS && j + 1 >= FU_PRE && j + 1 < ngroups) { // many groups per tile here: the next group's list words too
            const int bi1 = (j + 1) * GB + sub * FU_WAVES + wave;
            int b1 = b0 + bi1;
            if (b1 >= a.nb) b1 -= a.nb;
            const bool ok1 = bi1 < nbk;
            nx_cnt = ok1 ? a.dcnt[(size_t)e * a.nb + b1] : 0u;
            nx_ent = (ok1 && slot < a.dcap) ? a.dep[((size_t)e * a.nb + b1) * a.dcap + slot] : DEP_NULL;
        }
        cnt = min(cnt, (uint32_t)a.dcap);
        // NSLOT slots of each of the wave's buckets: compact the valid ones into the wave's segments.  Near a reflecting
        // wall only some deposits have an image in reach: x + p + 1 <= Rt or 2L - 1 - x - p <= Rt.
#define FU_ROUND(EN, K0) { \
            const bool valid = (K0) + slot < cnt; \
            const int dp = (int)((EN) & POS_MASK); \
            const bool img = valid && wall && ((x0 + dp + 1 <= Rt) || (2 * L - 1 - x1 - dp <= Rt)); \
            const unsigned long long m = __ballot(valid && !img), mi = __ballot(img); \
            if (nseg + 64 > SEG || nimg + 64 > SEG) flush(); \
            if (valid && !img) seg[nseg + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (EN); \
            if (img) segi[nimg + __builtin_amdgcn_mbcnt_hi((uint32_t)(mi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mi, 0u))] = (EN); \
            nseg += __popcll(m); nimg += __popcll(mi); }
        if (TAB_LDS) {
            for (uint32_t k0 = 0;; k0 += NSLOT) {
                FU_ROUND(ent, k0)
                if (!__ballot(k0 + NSLOT < cnt)) break;       // no bucket of this wave has more
                ent = (k0 + NSLOT + slot < cnt) ? a.dep[((size_t)e * a.nb + b) * a.dcap + k0 + NSLOT + slot] : DEP_NULL;
            }
        } else {
            // first round peeled: inside a loop the compiler waits (vmcnt(0)) for the reload of `ent` before every use,
            // and such a wait also waits for the next window, which is in flight on purpose
            FU_ROUND(ent, 0u)
            for (uint32_t k0 = NSLOT; __ballot(k0 < cnt); k0 += NSLOT) {
                const uint32_t en = (k0 + slot < cnt) ? a.dep[((size_t)e * a.nb + b) * a.dcap + k0 + slot] : DEP_NULL;
                FU_ROUND(en, k0)
            }
        }
#undef FU_ROUND
        if (!TAB_LDS && windowed) { FSTAMP(f_copy) flush(); FSTAMP(f_proc) }   // this group's deposits against this group's window
    }
    FSTAMP(f_copy)
    flush();
    FSTAMP(f_proc)
    __syncthreads();                                          // every wave is done with the table: its space takes the partial sums
#pragma unroll
    for (int r = 0; r < RS; ++r) red[(size_t)wave * SITES + r * 64 + lane] = make_double2(accW[r], accS[r]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NOLD; ++r) {
        const int xi = r * FU_THREADS + t, x = x0 + xi;
        if (xi < SITES && x < L) {
            double2 f = old[r];
#pragma unroll
            for (int w = 0; w < FU_WAVES; ++w) { const double2 pth = red[(size_t)w * SITES + xi]; f.x += pth.x; f.y += pth.y; }
            a.ws[(size_t)e * L + x] = f;
        }
    }
#ifdef APS_STAMPS
    FSTAMP(f_cnt)
    if (t == 0 && a.stamps && blockIdx.x < 4096) {
        unsigned long long *o = a.stamps + (size_t)blockIdx.x * 8;
        o[0] = f_cnt; o[1] = f_stage; o[2] = f_copy; o[3] = f_proc; o[4] = __builtin_amdgcn_s_memtime() - f_start;
        o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = r_start; o[7] = f_n;
    }
#endif
}

// Rate vectors of step_gillespie (ref :254-352) for caller-supplied particles, m-field and site histograms:
// out[c][i], c = diff, act, flip, bind, unbind, exit, left, right, total.
struct RatesArgs {
    Model m; double beta; const int32_t *pos; const int8_t *sigma; const uint8_t *bound; const double *m_field;
    const int32_t *occ; const uint8_t *anchor; double *out; long long n;
};

__global__ __launch_bounds__(256) void rates_kernel(const RatesArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const Model &M = a.m;
    const int p = a.pos[i], L = M.L;
    int l = p - 1, r = p + 1;
    if (M.periodic) { l = l < 0 ? l + L : l; r = r >= L ? r - L : r; }
    else { l = l < 0 ? 0 : l; r = r > L - 1 ? L - 1 : r; }
    const Channels c = channels(M, a.anchor ? a.anchor[p] != 0 : false, p, a.sigma[i], a.bound[i] != 0, a.m_field[p], a.beta,
                                a.occ[p], a.occ[l], a.occ[r]);
    const double v[9] = {c.diff, c.act, c.flip, c.bind, c.unbind, c.leave, c.left, c.right, c.total};
#pragma unroll
    for (int k = 0; k < 9; ++k) a.out[(long long)k * a.n + i] = v[k];
}

// m-field on lattice sites: the same accumulation with the targets being sites instead of particles.
struct FieldArgs {
    Model m; const long long *gsum;
    double *m_out; int tlen, ntiles, e;
    double2 *ws_out;          // optional [L]: the raw sums {W, S} (from-scratch build of the lattice field)
};

template <int BC, bool TAB_LDS>
__global__ __launch_bounds__(NTHREADS) void field_sites(const FieldArgs a, const uint32_t *__restrict__ sp8,
                                                       const int4 *__restrict__ tinfo,
                                                       const double *__restrict__ table_g) {
    extern __shared__ double lds[];
    const Model &M = a.m;
    const int lane = threadIdx.x & (TILE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t tbase = 0;
    if (TAB_LDS) {
        stage_table(lds, table_g, a.tlen);
        typedef __attribute__((address_space(3))) double lds_double;
        tbase = (uint32_t)(size_t)(lds_double *)lds;
        __syncthreads();
    }
    uint32_t *ring = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds) + lds_table_bytes(a.tlen, TAB_LDS)) + wave * TILE;
    const int tlo = (blockIdx.x * WAVES + wave) * TILE;       // one wave = 64 consecutive sites
    if (tlo >= M.L) return;
    const int thi = min(tlo + TILE - 1, M.L - 1);
    const int x = tlo + lane;
    const uint32_t pi8 = (uint32_t)min(x, M.L - 1) << 3;
    const uint32_t pi8v[1] = {pi8};
    double accWv[1] = {0.0}, accSv[1] = {0.0};
    int cnt[1][3] = {{0, 0, 0}};
    accumulate_item<BC, TAB_LDS, 1>(nullptr, PLAN_CAP + 1, 0, 1, sp8, tinfo, a.ntiles, make_int4(tlo, thi, 0, TILE), pi8v, tbase, ring,
                                    table_g, a.tlen, M.L, accWv, accSv, cnt);
    double accW = accWv[0], accS = accSv[0];
    if (x >= M.L) return;
    if (a.ws_out) a.ws_out[x] = make_double2(accW, accS);
    if (!a.m_out) return;
    if (M.field_mode == 0) { accS = (double)a.gsum[2 * a.e]; accW = (double)a.gsum[2 * a.e + 1]; }
    double mloc = 0.0;
    if (accW > 0.0) { mloc = accS / accW; mloc = mloc > 1.0 ? 1.0 : (mloc < -1.0 ? -1.0 : mloc); }
    a.m_out[x] = mloc;
}

// Driver-side observables: the integer sums from which the sweep drivers' statistics are built
// (compute_v_eff_and_window, compute_blocking_probability, compute_mean_magnetizatoin, compute_D_eff_active,
// PARTICLE_solver_BIOLOGY_EXCLUSION_sweep_beta.py:123-229, :316-319, :500-525), taken on the device so that a sweep
// needs no M x L density arrays.  All outputs are exact integers; the float formulas stay on the host.
struct ScalarArgs {                          // blockIdx.y = ensemble index within the call (arrays are strided by it)
    const uint32_t *src;                     // [n][Npad]
    const uint32_t *ref;                     // [n][Npad] packed reference state by slot, or null
    const uint8_t *ref_ok;                   // [n] reference marked for this ensemble (null: all, if ref)
    const uint8_t *block_table;              // [(K+1)*(K+1)]: is a right neighbour holding (plus, minus) particles "blocking"?
    uint32_t *cnt_pm;                        // [n][L] per-site counts plus | minus << 16 (built by count_sites)
    long long *out;                          // [n][16]
    const int *lo_hi;                        // [n][2] site range of the range count, or null (lo, hi below)
    int Npad, L, K, x_wall, lo, hi;
};
enum { SC_N = 0, SC_SPIN, SC_POS, SC_WALL, SC_MAXPOS, SC_RANGE, SC_ATTEMPT, SC_BLOCKED, SC_DISP, SC_DISP2, SC_NDISP, SC_COUNT };

__global__ __launch_bounds__(256) void count_sites(const uint32_t *__restrict__ src, uint32_t *cnt_pm, int Npad, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Npad) return;
    const uint32_t w = src[(size_t)blockIdx.y * Npad + i];
    if (w & (DEAD_BIT | AWAY_BIT)) return;
    atomicAdd(&cnt_pm[(size_t)blockIdx.y * L + (w & POS_MASK)], (w & SPIN_BIT) ? 1u : 65536u);
}

__device__ inline long long wave_sum(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void observe_scalars(const ScalarArgs a) {
    long long v[SC_COUNT] = {0, 0, 0, 0, -1, 0, 0, 0, 0, 0, 0};
    const int en = blockIdx.y;
    const uint32_t *src = a.src + (size_t)en * a.Npad, *cnt_pm = a.cnt_pm + (size_t)en * a.L;
    const uint32_t *ref = (a.ref && (!a.ref_ok || a.ref_ok[en])) ? a.ref + (size_t)en * a.Npad : nullptr;
    const int lo = a.lo_hi ? a.lo_hi[2 * en] : a.lo, hi = a.lo_hi ? a.lo_hi[2 * en + 1] : a.hi;
    long long *out = a.out + (size_t)en * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.Npad; i += gridDim.x * blockDim.x) {
        const uint32_t w = src[i];
        if (w & (DEAD_BIT | AWAY_BIT)) continue;
        const int p = (int)(w & POS_MASK);
        const bool plus = (w & SPIN_BIT) != 0;
        v[SC_N] += 1; v[SC_SPIN] += plus ? 1 : -1; v[SC_POS] += p;
        v[SC_WALL] += p >= a.x_wall;
        v[SC_MAXPOS] = max(v[SC_MAXPOS], (long long)p);
        v[SC_RANGE] += (p >= lo && p <= hi);
        if (plus && p < a.L - 1) {                            // ref :197-229: movers on sites 0..L-2, blocked by the right neighbour
            const uint32_t c = cnt_pm[p + 1];
            v[SC_ATTEMPT] += 1;
            v[SC_BLOCKED] += a.block_table[(c & 0xFFFFu) * (a.K + 1) + (c >> 16)];
        }
        if (ref) {
            const uint32_t r = ref[i];
            if (!(r & DEAD_BIT)) { const long long d = (long long)p - (long long)(r & POS_MASK); v[SC_DISP] += d; v[SC_DISP2] += d * d; v[SC_NDISP] += 1; }
        }
    }
#pragma unroll
    for (int k = 0; k < SC_COUNT; ++k) {
        if (k == SC_MAXPOS) {
            long long m = v[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
            if ((threadIdx.x & 63) == 0) atomicMax(&out[k], m);
        } else {
            const long long s = wave_sum(v[k]);
            if ((threadIdx.x & 63) == 0 && s) atomicAdd(reinterpret_cast<unsigned long long *>(&out[k]), (unsigned long long)s);
        }
    }
}

#include "tile_step.hpp"
#include "tile_dense.hpp"
#include "tile_loop.hpp"
#include "ntt_conv.hpp"

// fp32 mode: the field as the tile kernel keeps it (int32, units of 2^-q) <-> the binary64 view of the other kernels; exact both ways
__global__ __launch_bounds__(256) void ws_to_int(const double2 *__restrict__ in, int2 *__restrict__ out, size_t n, double up) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const double2 v = in[i]; out[i] = make_int2((int)(v.x * up), (int)(v.y * up)); }
}
__global__ __launch_bounds__(256) void ws_to_double(const int2 *__restrict__ in, double2 *__restrict__ out, size_t n, double down) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int2 v = in[i]; out[i] = make_double2((double)v.x * down, (double)v.y * down); }
}

// halo of a site-sharded tiles handle: its segments (ranges of the cell / ws / dcnt / dep arrays) <-> ONE packed message per
// neighbour.  Every transport (ncclSend / ncclRecv, device copy between two handles, host bytes) moves packed messages
// that this one kernel packs and unpacks.  All segment sizes are multiples of 4 bytes.
struct HaloSegD { unsigned long long arr_off, msg_off; unsigned bytes; int array; };
constexpr int HALO_BPS = 32;                 // workgroups per segment
__global__ __launch_bounds__(256) void halo_move(const HaloSegD *__restrict__ segs, char *a0, char *a1, char *a2, char *a3, char *msg, int unpack) {
    const HaloSegD g = segs[blockIdx.x / HALO_BPS];
    char *arr = (g.array == 0 ? a0 : g.array == 1 ? a1 : g.array == 2 ? a2 : a3) + g.arr_off;
    char *m = msg + g.msg_off;
    const unsigned words = g.bytes >> 2;
    for (unsigned i = (blockIdx.x % HALO_BPS) * 256 + threadIdx.x; i < words; i += HALO_BPS * 256) {
        if (unpack) reinterpret_cast<uint32_t *>(arr)[i] = reinterpret_cast<const uint32_t *>(m)[i];
        else reinterpret_cast<uint32_t *>(m)[i] = reinterpret_cast<const uint32_t *>(arr)[i];
    }
}

// ---- the same halo by PEER STORES (aps_ipc_export / aps_ipc_connect): a rank writes its packed block straight into the
// landing buffer of its neighbour rank -- device memory of the neighbour's process, mapped here through a HIP IPC handle
// (over xGMI when the neighbour is another GPU) -- followed by ONE arrival word carrying the exchange's tag; the neighbour's
// pull kernel waits for the tag and unpacks.  No RCCL kernel, no host call per exchange, nothing but two small launches on
// the handle's stream.  Landing buffers and arrival words are double buffered by exchange parity: a rank can only start
// exchange m after it has pulled exchange m - 1, which its neighbour pushed after pulling m - 2 from the buffer m will reuse.
struct HaloPushArgs {
    const HaloSegD *segs[2];              // [side]: this rank's first / last block
    int nseg[2];
    char *dst[2];                         // the neighbour's landing buffer for that block (peer memory)
    unsigned long long *flag[2];          // its arrival word
    unsigned *done;                       // [2] local completion counters (zero between launches)
    unsigned tag;
};
__global__ __launch_bounds__(256) void halo_push(const HaloPushArgs a, char *a0, char *a1, char *a2, char *a3) {
    const int nb0 = a.nseg[0] * HALO_BPS;
    const int side = (int)blockIdx.x < nb0 ? 0 : 1;
    const int bl = (int)blockIdx.x - (side ? nb0 : 0);
    const HaloSegD g = a.segs[side][bl / HALO_BPS];
    const char *arr = (g.array == 0 ? a0 : g.array == 1 ? a1 : g.array == 2 ? a2 : a3) + g.arr_off;
    char *m = a.dst[side] + g.msg_off;
    const unsigned words = g.bytes >> 2;
    for (unsigned i = (unsigned)(bl % HALO_BPS) * 256 + threadIdx.x; i < words; i += HALO_BPS * 256)
        reinterpret_cast<uint32_t *>(m)[i] = reinterpret_cast<const uint32_t *>(arr)[i];
    __threadfence_system();               // this thread's stores have reached the neighbour's memory
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned nblk = (unsigned)a.nseg[side] * HALO_BPS;
        if (atomicAdd(&a.done[side], 1u) == nblk - 1u) {          // the last block of this message: everything of it has landed
            __hip_atomic_store(&a.done[side], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(a.flag[side], (unsigned long long)a.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct HaloPullArgs {
    const HaloSegD *segs[2];              // [from_side]: the right neighbour's first block / the left neighbour's last
    int nseg[2];
    const char *src[2];                   // this rank's landing buffers
    const unsigned long long *flag[2];
    unsigned tag;
    unsigned long long timeout_ticks;     // of the 100 MHz clock
    unsigned *err_host;                   // host-mapped: a wait ran out
};
__global__ __launch_bounds__(256) void halo_pull(const HaloPullArgs a, char *a0, char *a1, char *a2, char *a3) {
    __shared__ int ok_s;
    const int nb0 = a.nseg[0] * HALO_BPS;
    const int side = (int)blockIdx.x < nb0 ? 0 : 1;
    const int bl = (int)blockIdx.x - (side ? nb0 : 0);
    if (threadIdx.x == 0) {               // bounded wait for the neighbour's arrival word
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 0;
        for (unsigned spins = 0;; ++spins) {
            const unsigned long long v = __hip_atomic_load(a.flag[side], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)((unsigned)v - a.tag) >= 0) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
            if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) break;
        }
        if (!ok) __hip_atomic_store(a.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok_s = ok;
    }
    __syncthreads();
    if (!ok_s) return;
    __threadfence_system();
    const HaloSegD g = a.segs[side][bl / HALO_BPS];
    char *arr = (g.array == 0 ? a0 : g.array == 1 ? a1 : g.array == 2 ? a2 : a3) + g.arr_off;
    const char *m = a.src[side] + g.msg_off;
    const unsigned words = g.bytes >> 2;
    // (system-scope loads: written by another device / process, never served from this device's caches)
    for (unsigned i = (unsigned)(bl % HALO_BPS) * 256 + threadIdx.x; i < words; i += HALO_BPS * 256)
        reinterpret_cast<uint32_t *>(arr)[i] = __hip_atomic_load(reinterpret_cast<const uint32_t *>(m) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// streaming copy, 16 bytes per lane: the HBM ceiling this box reaches in practice (bench.py quotes it beside the 8 TB/s spec)
__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    // four independent 16-byte loads per thread in flight, then the four stores: one workgroup = 16 KB
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t i = base + (size_t)u * 256; v[u] = i < n ? src[i] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t i = base + (size_t)u * 256; if (i < n) dst[i] = v[u]; }
}

// ---- site-centric state <-> particle-indexed arrays (observation, hooks, upload of the tiles method)
// cells -> src (live particles only: an exit wrote its own record), occupancy per site
__global__ __launch_bounds__(256) void cells_to_slots(const uint32_t *__restrict__ cell, const uint32_t *__restrict__ slot_of,
                                                      uint32_t *src, uint32_t *occ_site, int L, int K, long long N, int Npad, int s_lo, int s_hi) {
    const int s = s_lo + blockIdx.x * blockDim.x + threadIdx.x, e = blockIdx.y;
    if (s >= s_hi) return;
    int n = 0;
    for (int k = 0; k < K; ++k) {
        const uint32_t c = cell[((size_t)e * L + s) * K + k];
        if (c == CELL_EMPTY) continue;
        ++n;
        src[(size_t)e * Npad + slot_of[(size_t)e * N + (c & CELL_ID)]] =
            (uint32_t)s | ((c & CELL_PLUS) ? SPIN_BIT : 0u) | ((c & CELL_BOUND) ? BOUND_BIT : 0u);
    }
    occ_site[(size_t)e * L + s] = (uint32_t)n;
}

// site-sharded handles: before the cells of the own sites are written back, every live record is marked "away"
__global__ __launch_bounds__(256) void mark_away(uint32_t *src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(src[i] & DEAD_BIT)) src[i] |= AWAY_BIT;
}

// src -> pre-decoded source words and per-tile info (what apply() maintains in the particle-indexed formulations)
__global__ __launch_bounds__(256) void derive_slots(const uint32_t *__restrict__ src, uint32_t *sp8, int4 *tinfo, int Npad, int ntiles) {
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    const uint32_t me = src[(size_t)e * Npad + slot];
    const bool live = !(me & (DEAD_BIT | AWAY_BIT));
    const int p = (int)(me & POS_MASK);
    sp8[(size_t)e * Npad + slot] = live ? (((uint32_t)p << 3) | ((me & SPIN_BIT) ? 1u : 0u)) : DEAD_P8;
    int lo = live ? p : 0x7fffffff, hi = live ? p : -1, cnt = live ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); cnt += __shfl_xor(cnt, off);
    }
    if ((threadIdx.x & 63) == 0) tinfo[(size_t)e * ntiles + slot / TILE] = make_int4(lo, hi, cnt < TILE ? 1 : 0, cnt);
}

// per-tile spin sum / live count of the cells (global-field mode), and their total into gsum
__global__ __launch_bounds__(256) void tile_parts(const uint32_t *__restrict__ cell, long long *gpart, long long *gsum, int L, int K, int own, int ntile) {
    const int tile = blockIdx.x, e = blockIdx.y;
    const int s0 = tile * own, s1 = min(s0 + own, L);
    long long sp = 0, n = 0;
    for (int i = threadIdx.x; i < (s1 - s0) * K; i += blockDim.x) {
        const uint32_t c = cell[((size_t)e * L + s0) * K + i];
        if (c != CELL_EMPTY) { sp += (c & CELL_PLUS) ? 1 : -1; n += 1; }
    }
    __shared__ long long acc[2];
    if (threadIdx.x == 0) acc[0] = acc[1] = 0;
    __syncthreads();
    const long long a0 = wave_sum(sp), a1 = wave_sum(n);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[0]), (unsigned long long)a0);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[1]), (unsigned long long)a1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        gpart[((size_t)e * ntile + tile) * 2] = acc[0]; gpart[((size_t)e * ntile + tile) * 2 + 1] = acc[1];
        atomicAdd(reinterpret_cast<unsigned long long *>(&gsum[2 * e]), (unsigned long long)acc[0]);
        atomicAdd(reinterpret_cast<unsigned long long *>(&gsum[2 * e + 1]), (unsigned long long)acc[1]);
    }
}

// ---- structure observables on the device (extract_structure_observables_from_out, PARTICLE_solver_BIOLOGY_local_structure.py:55-103)
// per-site sums: sum over sites of (count_plus + count_minus)^2, of the local magnetisation and of its square
__global__ __launch_bounds__(256) void structure_sites(const uint32_t *__restrict__ cnt_pm, const double *__restrict__ m_field, int L, double *out) {
    double c2 = 0.0, m1 = 0.0, m2 = 0.0;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < L; x += gridDim.x * blockDim.x) {
        const uint32_t c = cnt_pm[x];
        const double tot = (double)((c & 0xFFFFu) + (c >> 16)), m = m_field[x];
        c2 += tot * tot; m1 += m; m2 += m * m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { c2 += __shfl_xor(c2, off); m1 += __shfl_xor(m1, off); m2 += __shfl_xor(m2, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[1], c2); atomicAdd(&out[2], m1); atomicAdd(&out[3], m2); }
}

// Fourier sums of the site histogram, one workgroup per mode k: sum over live particles of exp(-2 pi i k pos / L)
// (= fft(counts)[k], the reference's np.fft.fft(total) up to the 1 / (n dx) normalisation, ref :527-533)
__global__ __launch_bounds__(256) void structure_dft(const uint32_t *__restrict__ src, int Npad, int L, double *out) {
    const int k = blockIdx.x;
    double re = 0.0, im = 0.0, n = 0.0;
    for (int i = threadIdx.x; i < Npad; i += blockDim.x) {
        const uint32_t w = src[i];
        if (w & (DEAD_BIT | AWAY_BIT)) continue;
        const long long r = ((long long)k * (long long)(w & POS_MASK)) % (long long)L;   // exact argument reduction
        double sn, cs;
        sincospi(-2.0 * ((double)r / (double)L), &sn, &cs);
        re += cs; im += sn; n += 1.0;
    }
    __shared__ double acc[3];
    if (threadIdx.x < 3) acc[threadIdx.x] = 0.0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { re += __shfl_xor(re, off); im += __shfl_xor(im, off); n += __shfl_xor(n, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[0], re); atomicAdd(&acc[1], im); atomicAdd(&acc[2], n); }
    __syncthreads();
    if (threadIdx.x == 0) { out[4 + 2 * k] = acc[0]; out[5 + 2 * k] = acc[1]; if (k == 0) out[0] = acc[2]; }
}

// coarse-grained site histograms: particles of either spin per bin of `bin_sites` consecutive sites (the PDE grid of the
// hydrodynamic-limit comparison, BASELINE config 5); exact integers
__global__ __launch_bounds__(256) void bin_counts(const uint32_t *__restrict__ src, int Npad, int bin_sites, unsigned long long *plus, unsigned long long *minus,
                                                  int site_lo, int site_hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Npad) return;
    const uint32_t w = src[i];
    if (w & (DEAD_BIT | AWAY_BIT)) return;
    if ((int)(w & POS_MASK) < site_lo || (int)(w & POS_MASK) >= site_hi) return;       // a site-sharded handle counts the particles on its own sites
    atomicAdd(((w & SPIN_BIT) ? plus : minus) + (w & POS_MASK) / (uint32_t)bin_sites, 1ull);
}

// ---------------------------------------------------------------------------------------------
std::string g_create_error;

}  // namespace

struct aps_handle {
    aps_params p{};
    Model model{};
    std::vector<double> beta;
    std::vector<double> table;     // tlen + 1 entries, last one 0
    int tlen = 0, q = 0;
    bool table_in_lds = true;
    size_t lds_bytes = 0, lds_bytes_field = 0;
    int E = 1, world = 1, rank = 0;
    int64_t N = 0, Npad = 0, SH = 0, ntiles = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t *d_src = nullptr, *d_orig = nullptr, *d_pcnt = nullptr, *d_plist = nullptr;
    uint8_t *d_prop = nullptr, *d_prop_own = nullptr, *d_anchor = nullptr;
    uint32_t *d_sp8 = nullptr;
    int4 *d_tinfo = nullptr;
    unsigned long long *d_stamps = nullptr;
    uint32_t *d_plan = nullptr, *d_plan_n = nullptr;
    double *d_accW = nullptr, *d_accS = nullptr;
    unsigned *d_occ = nullptr;
    int split = 1;
    bool plan_dirty = true;
    int plan_interval = 1, plan_age = 0;       // steps a plan stays valid (margin = 2 * interval sites)
    int num_cu = 256, wgs_per_cu = 1;
    double *d_table = nullptr, *d_beta = nullptr, *d_exit = nullptr, *d_S = nullptr, *d_W = nullptr, *d_mfield = nullptr;
    int *d_occ4 = nullptr;
    long long *d_gsum = nullptr;
    unsigned *d_nexit = nullptr;
    uint32_t *d_tmp_sp8 = nullptr; int4 *d_tmp_tinfo = nullptr; size_t tmp_cap = 0;
    int exit_cap = 0;
    int64_t step = 0;
    std::vector<int64_t> n_set;    // particles uploaded per ensemble
    std::vector<hipEvent_t> events;
    ncclComm_t comm = nullptr;     // set by aps_comm_init: aps_step then all-gathers the proposals itself
    // lattice formulation
    int method = APS_METHOD_PAIRS;
    double2 *d_ws = nullptr;                   // [E][L] {W, S}
    uint32_t *d_occ_site = nullptr;            // [E][L]
    uint32_t *d_dcnt = nullptr, *d_dep = nullptr;
    unsigned long long *d_stepw = nullptr;     // [2] device step words
    int bshift = 8, nb = 0, dcap = 0, fu_R = 1;
    bool fu_table_in_lds = true;
    bool field_dirty = true;
    hipStream_t cap_stream = nullptr;
    // captured graphs bake the buffer pointers in: one set per `flip` (which physical buffer set holds the even steps; the
    // resident loop swaps the sets after a call of an even number of steps)
    hipGraphExec_t gexec_f[2][2][6] = {};   // [flip][start parity][k]: runs of GRAPH_SIZES[k] steps
    hipGraphExec_t gexact_f[2][2][65] = {}; // [flip][start parity][n]: a run of exactly n <= 64 steps, captured the first time aps_step(n) is called
    bool graphs_built_f[2] = {false, false};
    int flip = 0;
    int64_t last_graph_steps = 0, last_single_steps = 0;      // how the last aps_step call was executed
    // tiles formulation (site-centric state, one kernel per step): everything double buffered by step parity
    double2 *d_wsb[2] = {nullptr, nullptr};
    uint32_t *d_cell[2] = {nullptr, nullptr}, *d_tdcnt[2] = {nullptr, nullptr}, *d_tdep[2] = {nullptr, nullptr};
    long long *d_gpart[2] = {nullptr, nullptr};
    uint32_t *d_slot_of = nullptr;
    Model *d_model = nullptr; TileRare *d_rare = nullptr;      // device copies read by the tile kernel
    // resident loop (tile_loop): many steps per launch while every tile of the grid is resident at once
    unsigned long long *d_xrec = nullptr;                      // [2][E][ntile][loop_rec] exchange records (8-byte granules)
    unsigned *d_abort = nullptr, *h_abort = nullptr, *h_abort_dev = nullptr;   // "a wait ran out": device word, host-mapped word and its device address
    int loop_rec = 0, loop_drec = 0, loop_seg = 0;
    int loop_state = -2;                                       // -2 not looked at yet, -1 gave up once (never again), 0 not eligible, 1 usable
    int loop_wanted = 1;                                       // aps_set_resident_loop
    uint32_t loop_tag = 0;                                     // tags handed out so far
    int64_t last_loop_steps = 0;
    unsigned loop_stall = 0u;                                  // APS_LOOP_TEST_STALL word currently in d_abort[1]
    uint32_t *loop_dbg = nullptr; int loop_dbg_n = 0;          // APS_LOOP_DEBUG builds
    bool loop_timed = false;                                   // the next loop launch carries start/stop events (aps_step_loop_timed)
    hipEvent_t loop_ev[2] = {nullptr, nullptr};
    std::string loop_why;                                      // why the loop is not used
    // fp32 mode: the tile kernel's field is int32 in units of 2^-q; d_wsb then serves as the double2 view the hooks read
    bool f32 = false, ws_view_stale = false;
    double *d_flip_tab = nullptr;              // aps_set_flip_table
    // the field update as an exact number-theoretic convolution (ntt_conv.hpp): 32-bit field, table beyond LDS, one rank, walls
    bool ntt_on = false, ntt_fused = false;
    NttPlan ntt{};
    uint32_t *d_ntt_sig = nullptr, *d_ntt_tab = nullptr;       // [E][2][M] residues; all the tables in one allocation
    int *d_ntt_csig = nullptr;                                 // [E][2][M] deposit coefficients of the step (index = site + Rt), cleared by the transform
    double prof_ntt_ms = 0.0; int64_t prof_ntt_n = 0;          // last profiling run: the convolution's launches
    int *d_table_i = nullptr;
    int2 *d_wsi[2] = {nullptr, nullptr};
    // site-range sharding of the tiles formulation: this rank steps tiles [ts_lo, ts_hi) = sites [own_lo, own_hi)
    int ts_lo = 0, ts_hi = 0, own_lo = 0, own_hi = 0, ts_reach = 0;
    // steps per halo exchange (ghost zone of (ts_kx - 1) * ts_reach tiles per side stepped redundantly), steps since the
    // last exchange, and which neighbour blocks of a due exchange have arrived (bit 0: the right one's, bit 1: the left one's)
    int ts_kx = 1, halo_age = 0;
    unsigned halo_got = 0;
    // packed halo messages: [0] this rank's first block (for the left neighbour) / the block received from the RIGHT neighbour
    // (its first block); [1] this rank's last block / the block received from the LEFT neighbour (its last block)
    char *d_halo_send[2] = {nullptr, nullptr}, *d_halo_recv[2] = {nullptr, nullptr};
    HaloSegD *d_halo_seg_send[2] = {nullptr, nullptr}, *d_halo_seg_recv[2] = {nullptr, nullptr};
    int halo_nseg_send[2] = {0, 0}, halo_nseg_recv[2] = {0, 0};
    size_t halo_bytes_send[2] = {0, 0}, halo_bytes_recv[2] = {0, 0};
    // peer-store transport of the halo (aps_ipc_export / aps_ipc_connect)
    char *ipc_land = nullptr;                  // this rank's landing allocation: arrival words, then [parity][from_side] buffers
    size_t ipc_land_bytes = 0, ipc_land_off[2][2] = {{0, 0}, {0, 0}};
    char *ipc_peer[2] = {nullptr, nullptr};    // [0] the left neighbour's landing allocation as mapped here, [1] the right one's
    void *ipc_opened[2] = {nullptr, nullptr};  // what hipIpcOpenMemHandle returned (closed in aps_destroy)
    size_t ipc_peer_off[2][2] = {{0, 0}, {0, 0}};   // [side][parity]: where this rank's block `side` lands in that neighbour
    unsigned *d_ipc_done = nullptr, *h_ipc_err = nullptr, *h_ipc_err_dev = nullptr;
    uint32_t ipc_seq = 0;                      // exchanges made so far
    bool ipc_on = false;
    int ts_RS = 2, ts_own = 124, ts_ntile = 0, ts_dcap = 0;
    bool ts_table_in_lds = true;
    bool slots_dirty = false;                  // the particle-indexed arrays lag behind the cells
    bool field_pending = false;                // the last step's deposits are not yet added to ws[cur]
    uint32_t *d_ref = nullptr, *d_cnt_pm = nullptr;   // observables: reference state per slot [E][Npad], per-site counts [L]
    uint8_t *d_block_table = nullptr, *d_ref_ok = nullptr; long long *d_scal = nullptr; int *d_lo_hi = nullptr;
    std::vector<char> ref_set;                         // per ensemble: reference marked (and slot order unchanged since)
    // per-kernel timing (aps_step_timed / aps_step_profile): an event before every launch, kind of that launch
    bool profiling = false;
    std::vector<int> prof_kind;
    size_t prof_n = 0;
    bool prof_dispatch = true;          // profiling runs: events attached to the kernel's own dispatch (else: events around it)
    hipEvent_t k_start = nullptr, k_stop = nullptr;
    std::string err;
};

namespace {

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return APS_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

int fail(aps_handle *h, int code, const std::string &msg) { h->err = msg; return code; }
void drop_graphs(aps_handle *h);

// RCCL is resolved at run time (no link dependency; reuses the copy the process already loaded, e.g. PyTorch's)
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(lib, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(lib, "ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString || !CommCount || !Send || !Recv || !GroupStart || !GroupEnd) {
            err = "librccl lacks a required symbol"; lib = nullptr; return false;
        }
        return true;
    }
} g_rccl;

void build_table(aps_handle *h) { weight_table(h->p.sigma_grid, h->p.L, h->p.K, h->p.periodic != 0, h->table, h->tlen, h->q, h->p.fp32 ? 29 : 51); }

template <typename T>
int dev_alloc(aps_handle *h, T **ptr, size_t count) {
    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(ptr), std::max<size_t>(count, 1) * sizeof(T)));
    HIP_TRY(h, hipMemsetAsync(*ptr, 0, std::max<size_t>(count, 1) * sizeof(T), h->stream));
    return APS_OK;
}

// per-tile info and pre-decoded source words of a slot array (host mirror of what apply() maintains)
void derive_sources(const std::vector<uint32_t> &src, std::vector<uint32_t> &sp8, std::vector<int4> &tinfo) {
    const size_t n = src.size(), nt = n / TILE;
    sp8.resize(n);
    tinfo.assign(nt, make_int4(0x7fffffff, -1, 1, 0));
    for (size_t s = 0; s < n; ++s) {
        const uint32_t w = src[s];
        const bool live = !(w & DEAD_BIT);
        sp8[s] = live ? (((w & POS_MASK) << 3) | ((w & SPIN_BIT) ? 1u : 0u)) : DEAD_P8;
        if (!live) continue;
        int4 &t = tinfo[s / TILE];
        const int p = (int)(w & POS_MASK);
        t.x = std::min(t.x, p); t.y = std::max(t.y, p); t.w += 1;
    }
    for (int4 &t : tinfo) t.z = t.w < TILE ? 1 : 0;
}

// host-side packing of one ensemble into slot order; fills the global sums
void pack_ensemble(const aps_handle *h, const int32_t *pos, const int8_t *sigma, const uint8_t *bound,
                   const uint8_t *alive, int64_t n, std::vector<uint32_t> &src, std::vector<uint32_t> &orig,
                   long long gsum[2]) {
    const int64_t Npad = h->Npad;
    std::vector<uint32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0u);
    if (h->p.sort_by_site)
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            const bool da = alive && !alive[a], db = alive && !alive[b];
            if (da != db) return db;                       // live particles first
            return pos[a] < pos[b];
        });
    src.assign((size_t)Npad, DEAD_BIT);
    orig.assign((size_t)Npad, 0xFFFFFFFFu);
    gsum[0] = gsum[1] = 0;
    for (int64_t s = 0; s < n; ++s) {
        const uint32_t i = order[(size_t)s];
        uint32_t w = (uint32_t)pos[i] & POS_MASK;
        if (sigma[i] > 0) w |= SPIN_BIT;
        if (bound && bound[i]) w |= BOUND_BIT;
        const bool dead = alive && !alive[i];
        if (dead) w |= DEAD_BIT;
        else { gsum[0] += sigma[i] > 0 ? 1 : -1; gsum[1] += 1; }
        src[(size_t)s] = w;
        orig[(size_t)s] = i;
    }
}

int upload_ensemble(aps_handle *h, int e, const std::vector<uint32_t> &src, const std::vector<uint32_t> &orig) {
    std::vector<uint32_t> sp8; std::vector<int4> tinfo;
    derive_sources(src, sp8, tinfo);
    HIP_TRY(h, hipMemcpyAsync(h->d_src + (size_t)e * h->Npad, src.data(), src.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_orig + (size_t)e * h->Npad, orig.data(), orig.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_sp8 + (size_t)e * h->Npad, sp8.data(), sp8.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_tinfo + (size_t)e * h->ntiles, tinfo.data(), tinfo.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->plan_dirty = true;
    return APS_OK;
}

PairArgs pair_args(aps_handle *h, bool hook, bool write_prop) {
    PairArgs a{};
    a.m = h->model;
    a.src = h->d_src; a.orig = h->d_orig;
    a.parity = (int)(h->step & 1);
    a.gsum = h->d_gsum + (size_t)a.parity * 2 * h->E; a.gsum_next = h->d_gsum + (size_t)(a.parity ^ 1) * 2 * h->E;
    a.plist = h->d_plist; a.fuse_claim = (h->world == 1 && write_prop) ? 1 : 0; a.stamps = h->d_stamps; a.plan_n = h->d_plan_n;
    a.accW = h->d_accW; a.accS = h->d_accS; a.occ = h->d_occ; a.split = h->split;
    a.beta = h->d_beta; a.anchor = h->d_anchor; a.prop = h->d_prop; a.pcnt = h->d_pcnt;
    a.S_out = hook ? h->d_S : nullptr; a.W_out = hook ? h->d_W : nullptr; a.occ4_out = hook ? h->d_occ4 : nullptr;
    a.tlen = h->tlen; a.Npad = (int)h->Npad; a.SH = (int)h->SH; a.E = h->E; a.ntiles = (int)h->ntiles;
    a.tile_lo = (int)(h->rank * h->SH / TILE); a.tile_cnt = (int)(h->SH / TILE);
    a.step_lo = (uint32_t)h->step; a.step_hi = (uint32_t)((uint64_t)h->step >> 32);
    a.write_prop = write_prop ? 1 : 0;
    return a;
}

int prof_mark(aps_handle *h, int kind);

// Every stepping kernel is launched through this: in a profiling run the start/stop events that prof_mark set up are
// attached to the dispatch itself (begin/end timestamps of the kernel, what rocprofv3 reports), otherwise a plain launch.
#define APS_K(h, kernel, grid, block, lds, ...) do { \
        if ((h)->profiling && (h)->prof_dispatch) hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), (h)->stream, (h)->k_start, (h)->k_stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, (h)->stream, __VA_ARGS__); } while (0)

int launch_plan(aps_handle *h, int first_tile, int tile_cnt) {
    PlanArgs pa{h->p.L, h->tlen, (int)h->ntiles, first_tile, tile_cnt, h->plan_interval > 1 ? 2 * h->plan_interval : 0,
                h->d_plan, h->d_plan_n};
    h->plan_age = 0;
    { int rc = prof_mark(h, 4 /* KIND_PLAN */); if (rc) return rc; }
    const dim3 grid((unsigned)((tile_cnt / RT + 3) / 4), (unsigned)h->E), block(256);
    if (h->p.periodic) APS_K(h, plan_tiles<1>, grid, block, 0, pa, h->d_tinfo);
    else APS_K(h, plan_tiles<0>, grid, block, 0, pa, h->d_tinfo);
    HIP_TRY(h, hipGetLastError());
    return APS_OK;
}

int prof_mark(aps_handle *h, int kind);

int launch_pair(aps_handle *h, const PairArgs &a, int first_tile, int tile_cnt) {
    PairArgs b = a;
    b.tile_lo = first_tile; b.tile_cnt = tile_cnt;
    const int shard_lo = (int)(h->rank * h->SH / TILE), shard_cnt = (int)(h->SH / TILE);
    if (h->plan_dirty || first_tile != shard_lo || tile_cnt != shard_cnt) {     // hook launches cover every tile
        int rc = launch_plan(h, first_tile, tile_cnt);
        if (rc) return rc;
        h->plan_dirty = (first_tile != shard_lo || tile_cnt != shard_cnt);
    }
    // Shares per target tile.  Items are dealt to the resident waves round-robin, so pick the split whose
    // item count fills whole rounds best (smallest idle fraction in the last round), with >= 2 rounds if possible.
    const unsigned slots = (unsigned)(h->num_cu * h->wgs_per_cu * WAVES);
    int split = 1;
    double best = 1e30;
    for (int sp = 1; sp <= MAX_SPLIT; ++sp) {
        const double items = (double)(tile_cnt / RT) * h->E * sp, rounds = std::ceil(items / slots);
        const double cost = rounds * slots / items * (1.0 + 0.02 * sp) * (rounds < 2 ? 1.5 : 1.0);   // waste x mild per-item overhead
        if (cost < best) { best = cost; split = sp; }
    }
    if (const char *env = std::getenv("APS_SPLIT")) split = std::max(1, std::min(MAX_SPLIT, std::atoi(env)));   // tuning knob
    b.split = h->split = split;
    const unsigned items = (unsigned)(tile_cnt / RT) * (unsigned)h->E * (unsigned)split;
    const dim3 grid(std::max(1u, std::min((items + WAVES - 1) / WAVES, (unsigned)(h->num_cu * h->wgs_per_cu)))), block(NTHREADS);
    const size_t lds = lds_total_bytes(h->tlen, h->table_in_lds);
    { int rc = prof_mark(h, 0 /* KIND_PAIR */); if (rc) return rc; }   // events bracket the dominant kernel alone
#define APS_LAUNCH(BC, TL) APS_K(h, (pair_accumulate<BC, TL>), grid, block, lds, b, h->d_sp8, h->d_tinfo, h->d_table, h->d_plan, h->d_plan_n)
    if (h->p.periodic) { if (h->table_in_lds) APS_LAUNCH(1, true); else APS_LAUNCH(1, false); }
    else { if (h->table_in_lds) APS_LAUNCH(0, true); else APS_LAUNCH(0, false); }
#undef APS_LAUNCH
    { int rc = prof_mark(h, 1 /* KIND_PROPOSE */); if (rc) return rc; }
    const dim3 pgrid((unsigned)((tile_cnt * TILE + 255) / 256), (unsigned)h->E);
    APS_K(h, propose, pgrid, dim3(256), 0, b);
    HIP_TRY(h, hipGetLastError());
    return APS_OK;
}

CommitArgs commit_args(aps_handle *h) {
    CommitArgs c{};
    c.m = h->model; c.src = h->d_src; c.orig = h->d_orig; c.prop = h->d_prop; c.pcnt = h->d_pcnt;
    c.sp8 = h->d_sp8; c.tinfo = h->d_tinfo;
    c.parity = (int)(h->step & 1);
    c.plist = h->d_plist; c.gsum = h->d_gsum + (size_t)(c.parity ^ 1) * 2 * h->E; c.exit_log = h->d_exit;
    c.n_exit = h->d_nexit; c.exit_cap = h->exit_cap; c.Npad = (int)h->Npad; c.SH = (int)h->SH; c.E = h->E;
    c.ntiles = (int)h->ntiles; c.step_as_double = (double)h->step;
    if (h->method == APS_METHOD_LATTICE) {
        c.stepw = h->d_stepw; c.occ_site = h->d_occ_site;
        if (h->model.field_mode) { c.dcnt = h->d_dcnt; c.dep = h->d_dep; }
        c.bshift = h->bshift; c.nb = h->nb; c.dcap = h->dcap;
    }
    return c;
}

int set_lds_limit(aps_handle *h) {
    h->table_in_lds = lds_total_bytes(h->tlen, true) <= 150 * 1024;
    const size_t need = lds_total_bytes(h->tlen, h->table_in_lds);
    if (need > 48 * 1024) {
        const int n = (int)need;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&pair_accumulate<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, n));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&pair_accumulate<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, n));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&field_sites<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, n));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&field_sites<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, n));
    }
    h->fu_table_in_lds = fu_lds_bytes(h->tlen, true, h->fu_R, h->bshift) <= 160 * 1024;
    if (fu_lds_bytes(h->tlen, h->fu_table_in_lds, h->fu_R, h->bshift) > 48 * 1024) {
        const int n = (int)fu_lds_bytes(h->tlen, h->fu_table_in_lds, h->fu_R, h->bshift);
#define APS_ATTR(BC, TL, RR) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&field_update<BC, TL, RR>), hipFuncAttributeMaxDynamicSharedMemorySize, n))
#define APS_ATTR4(BC, TL) APS_ATTR(BC, TL, 2); APS_ATTR(BC, TL, 3); APS_ATTR(BC, TL, 4); APS_ATTR(BC, TL, 5); APS_ATTR(BC, TL, 6); APS_ATTR(BC, TL, 7); APS_ATTR(BC, TL, 8)
        if (h->fu_table_in_lds) { APS_ATTR4(0, true); APS_ATTR4(1, true); } else { APS_ATTR4(0, false); APS_ATTR4(1, false); }
#undef APS_ATTR4
#undef APS_ATTR
    }
    hipDeviceProp_t prop;
    HIP_TRY(h, hipGetDeviceProperties(&prop, h->p.device));
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int by_lds = need ? (int)((160 * 1024) / need) : 8, by_waves = 16 / WAVES;   // 4 waves/SIMD saturate the LDS gather
    h->wgs_per_cu = std::max(1, std::min(by_lds, by_waves));
    {   // registers may allow fewer resident workgroups than LDS does: ask the runtime for this kernel
        int nb = 0;
        const void *fn = h->p.periodic ? (h->table_in_lds ? (const void *)&pair_accumulate<1, true> : (const void *)&pair_accumulate<1, false>)
                                       : (h->table_in_lds ? (const void *)&pair_accumulate<0, true> : (const void *)&pair_accumulate<0, false>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, NTHREADS, need) == hipSuccess && nb > 0)
            h->wgs_per_cu = std::min(h->wgs_per_cu, nb);
    }
    if (const char *env = std::getenv("APS_WGS_PER_CU")) h->wgs_per_cu = std::max(1, std::atoi(env));     // tuning knob
    return APS_OK;
}

int launch_field(aps_handle *h, int e, const uint32_t *sp8, const int4 *tinfo, int ntiles, double *m_out, double2 *ws_out = nullptr) {
    FieldArgs f{};
    f.m = h->model; f.gsum = h->d_gsum + (size_t)(h->step & 1) * 2 * h->E; f.m_out = m_out; f.ws_out = ws_out;
    f.tlen = h->tlen; f.ntiles = ntiles; f.e = e;
    const dim3 grid((unsigned)((h->p.L + TILE * WAVES - 1) / (TILE * WAVES))), block(NTHREADS);
    const size_t lds = lds_total_bytes(h->tlen, h->table_in_lds);
#define APS_LAUNCH(BC, TL) hipLaunchKernelGGL((field_sites<BC, TL>), grid, block, lds, h->stream, f, sp8, tinfo, h->d_table)
    if (h->p.periodic) { if (h->table_in_lds) APS_LAUNCH(1, true); else APS_LAUNCH(1, false); }
    else { if (h->table_in_lds) APS_LAUNCH(0, true); else APS_LAUNCH(0, false); }
#undef APS_LAUNCH
    HIP_TRY(h, hipGetLastError());
    return APS_OK;
}

enum { KIND_PAIR = 0, KIND_PROPOSE, KIND_CLAIM, KIND_APPLY, KIND_PLAN, KIND_PROPOSE_LATTICE, KIND_FIELD_UPDATE, KIND_TILE_STEP, KIND_NTT, KIND_END, KIND_N = KIND_END };

// profiling runs only.  Dispatch mode: hands the next APS_K launch a start/stop event pair of its own.  Bracket mode: an
// event in front of the launch that follows (KIND_END closes the last one of a step).
int prof_mark(aps_handle *h, int kind) {
    if (!h->profiling) return APS_OK;
    const size_t need = h->prof_dispatch ? 2 : 1;
    if (h->prof_dispatch && kind == KIND_END) return APS_OK;
    while (h->prof_n + need > h->events.size()) {
        hipEvent_t ev;
        HIP_TRY(h, hipEventCreate(&ev));
        h->events.push_back(ev);
    }
    if (h->prof_dispatch) {
        h->k_start = h->events[h->prof_n++]; h->k_stop = h->events[h->prof_n++];
    } else {
        HIP_TRY(h, hipEventRecord(h->events[h->prof_n++], h->stream));
    }
    h->prof_kind.push_back(kind);
    return APS_OK;
}

LatticeArgs lattice_args(aps_handle *h, bool hook, bool write_prop) {
    LatticeArgs a{};
    a.m = h->model; a.src = h->d_src; a.orig = h->d_orig; a.ws = h->d_ws; a.occ_site = h->d_occ_site;
    a.par = (int)(h->step & 1);
    a.gsum = h->d_gsum + (size_t)a.par * 2 * h->E; a.gsum_next = h->d_gsum + (size_t)(a.par ^ 1) * 2 * h->E;
    a.beta = h->d_beta; a.anchor = h->d_anchor; a.prop = h->d_prop; a.pcnt = h->d_pcnt; a.plist = h->d_plist;
    a.dcnt = h->model.field_mode ? h->d_dcnt : nullptr; a.stepw = h->d_stepw;
    a.S_out = hook ? h->d_S : nullptr; a.W_out = hook ? h->d_W : nullptr; a.occ4_out = hook ? h->d_occ4 : nullptr;
    a.fuse_claim = (h->world == 1 && write_prop) ? 1 : 0; a.write_prop = write_prop ? 1 : 0;
    a.Npad = (int)h->Npad; a.SH = (int)h->SH; a.E = h->E; a.nb = h->nb;
    a.tile_lo = (int)(h->rank * h->SH / TILE); a.tile_cnt = (int)(h->SH / TILE);
    return a;
}

int launch_field(aps_handle *h, int e, const uint32_t *sp8, const int4 *tinfo, int ntiles, double *m_out, double2 *ws_out);

// (re)build W, S on all sites from the particles (state upload; afterwards the field is kept incrementally)
int ensure_tiles(aps_handle *h);

// the device step words {even, odd} for the handle's step index (step n reads word n & 1 and writes n + 1 into the other)
int upload_stepw(aps_handle *h) {
    const unsigned long long sw[2] = {(unsigned long long)(h->step & 1 ? h->step - 1 : h->step), (unsigned long long)(h->step & 1 ? h->step : h->step + 1)};
    HIP_TRY(h, hipMemcpyAsync(h->d_stepw, sw, sizeof(sw), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return APS_OK;
}

int ensure_lattice(aps_handle *h) {
    if (h->method == APS_METHOD_TILES) return ensure_tiles(h);
    if (h->method != APS_METHOD_LATTICE) return APS_OK;
    if (h->field_dirty) {
        if (h->model.field_mode)
            for (int e = 0; e < h->E; ++e) {
                int rc = launch_field(h, e, h->d_sp8 + (size_t)e * h->Npad, h->d_tinfo + (size_t)e * h->ntiles, (int)h->ntiles,
                                      nullptr, h->d_ws + (size_t)e * h->p.L);
                if (rc) return rc;
            }
        h->field_dirty = false;
        // the step words are kept by the kernels themselves (step n writes n + 1 into the other word); set them once more
        // here so that a handle whose state was replaced mid-run starts from a known pair
        const unsigned long long s[2] = {(unsigned long long)(h->step & 1 ? h->step - 1 : h->step), (unsigned long long)(h->step & 1 ? h->step : h->step + 1)};
        HIP_TRY(h, hipMemcpyAsync(h->d_stepw, s, sizeof(s), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return APS_OK;
}

int launch_lattice_propose(aps_handle *h, const LatticeArgs &a, int first_tile, int tile_cnt) {
    LatticeArgs b = a;
    b.tile_lo = first_tile; b.tile_cnt = tile_cnt;
    int rc = prof_mark(h, KIND_PROPOSE_LATTICE);
    if (rc) return rc;
    const dim3 grid((unsigned)((tile_cnt * TILE + 255) / 256), (unsigned)h->E);
    APS_K(h, propose_lattice, grid, dim3(256), 0, b);
    HIP_TRY(h, hipGetLastError());
    return APS_OK;
}

int launch_field_update(aps_handle *h) {
    if (!h->model.field_mode) return APS_OK;
    int rc = prof_mark(h, KIND_FIELD_UPDATE);
    if (rc) return rc;
    FieldUpdArgs f{h->p.L, h->tlen, h->bshift, h->nb, h->dcap, h->d_ws, h->d_dcnt, h->d_dep, h->d_stamps};
    const int RS = h->fu_R;
    const dim3 grid((unsigned)((h->p.L + 64 * RS - 1) / (64 * RS)), (unsigned)h->E), block(FU_THREADS);
    const size_t lds = fu_lds_bytes(h->tlen, h->fu_table_in_lds, RS, h->bshift);
#define APS_FU(BC, TL, RR) APS_K(h, (field_update<BC, TL, RR>), grid, block, lds, f, h->d_table)
#define APS_FU_R(BC, TL) do { switch (RS) { case 8: APS_FU(BC, TL, 8); break; case 7: APS_FU(BC, TL, 7); break; case 6: APS_FU(BC, TL, 6); break; \
        case 5: APS_FU(BC, TL, 5); break; case 4: APS_FU(BC, TL, 4); break; case 3: APS_FU(BC, TL, 3); break; default: APS_FU(BC, TL, 2); } } while (0)
    if (h->p.periodic) { if (h->fu_table_in_lds) APS_FU_R(1, true); else APS_FU_R(1, false); }
    else { if (h->fu_table_in_lds) APS_FU_R(0, true); else APS_FU_R(0, false); }
#undef APS_FU_R
#undef APS_FU
    HIP_TRY(h, hipGetLastError());
    return APS_OK;
}


// ------------------------------------------------------------------------------- tiles formulation, host side
bool is_tiles(const aps_handle *h) { return h->method == APS_METHOD_TILES; }

template <bool F32>
const void *ts_kernel_f(bool periodic, bool tab, int RS, bool k1) {
#define TS_PICK(BC, TL, R) (k1 ? (const void *)&tile_step<BC, TL, R, true, F32> : (const void *)&tile_step<BC, TL, R, false, F32>)
#define TS_CASE(R) case R: return periodic ? (tab ? TS_PICK(1, true, R) : TS_PICK(1, false, R)) : (tab ? TS_PICK(0, true, R) : TS_PICK(0, false, R));
#ifdef APS_DEV_RS                 /* development builds: one frame size only (compiles in a fraction of the time) */
    switch (RS) { TS_CASE(APS_DEV_RS) default: return nullptr; }
#else
    switch (RS) { TS_CASE(1) TS_CASE(2) TS_CASE(3) TS_CASE(4) TS_CASE(5) TS_CASE(6) TS_CASE(7) TS_CASE(8) default: return nullptr; }
#endif
#undef TS_CASE
#undef TS_PICK
}
const void *ts_kernel(bool periodic, bool tab, int RS, bool k1, bool f32 = false) {
    return f32 ? ts_kernel_f<true>(periodic, tab, RS, k1) : ts_kernel_f<false>(periodic, tab, RS, k1);
}
constexpr int TS_RS_CHOICES[] = {1, 2, 3, 4, 5, 6, 7, 8};

int ts_wbytes(const aps_handle *h) { return h->f32 ? 4 : 8; }

// tile geometry (measured on MI355X, profiles/r02_*): while the whole grid is resident at once (<= 3 workgroups per CU)
// the frame that gives about 2.4 workgroups per CU is fastest (64 * 5 sites at L = 2e5); larger grids run in waves of
// workgroups and want the frame with the best work per instruction at three or four resident workgroups per CU: 256 sites
// in binary64, 384 (table in LDS) or 320 (table windows) sites with the 32-bit field
void ts_choose_geometry(aps_handle *h) {
    const int L = h->p.L;
    h->ts_RS = 1;
    double best = 1e300;
    for (int rs : TS_RS_CHOICES) {
        const int own = 64 * rs - 4;
        const double wgs = (double)(((int64_t)L + own - 1) / own) * h->E;
        const double miss = std::fabs(wgs - 2.4 * 256.0);
        if (miss < best) { best = miss; h->ts_RS = rs; }
    }
    if ((double)(((int64_t)L + 507) / 508) * h->E > 3.0 * 256.0)      // even the largest frame leaves more than 3 per CU
        h->ts_RS = !h->f32 ? 4 : 7;   // measured (r02 geometry sweeps): binary64 config 4 RS 3..8 = 67, 59.5, 65, 62, 73, 74 us, config 5 RS 4..6 = 550, 607, 544;
                                     // 32-bit field config 4 RS 4..8 = 56, 52, 50, 46.7, 47.4 us, config 5 = 428, 345, 347, 312, 371
    // One system whose tiles can all stay resident (the resident loop, tile_loop.hpp: every workgroup keeps its tile for the
    // whole call, so the busiest CU sets the pace): exactly two workgroups on every CU instead of three on some and two on
    // the others (config 2: 512 tiles of 391 sites, 12.8 us per step against 13.3 with 633 tiles of 316; one launch per
    // step is indifferent: 14.65 against 14.55)
    int own_even = 0;
    if (h->world == 1 && h->model.field_mode && 3 * h->p.K <= 32 && !(h->p.K == 1 && h->model.immobilize && h->model.k_exit > 0.0)) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->p.device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        const int64_t target = 2LL * cus / std::max(h->E, 1);
        if (target >= 1) {
            const int own = (int)((L + target - 1) / target), rs = (own + 4 + 63) / 64;
            if (own >= 124 && rs <= 8) { h->ts_RS = rs; own_even = own; }
        }
    }
    if (const char *env = std::getenv("APS_TS_R")) { const int r = std::atoi(env); if (ts_kernel(false, true, r, true)) { h->ts_RS = r; own_even = 0; } }
    h->ts_own = own_even ? own_even : 64 * h->ts_RS - 4;
    if (const char *env = std::getenv("APS_TS_OWN")) { const int o = std::atoi(env); if (o >= 32 * h->ts_RS && o <= 64 * h->ts_RS - 4) h->ts_own = o; }
    // a last tile of one or two sites (the resident loop wants three: its neighbour's halo must lie inside it): slightly smaller tiles
    while (h->ts_own > 32 * h->ts_RS && L > h->ts_own && (L % h->ts_own == 1 || L % h->ts_own == 2)) h->ts_own -= 1;
    h->ts_ntile = (L + h->ts_own - 1) / h->ts_own;
    h->ts_dcap = (int)std::max<int64_t>(2, std::min<int64_t>(2LL * h->p.K * h->ts_own, 2 * h->p.n_particles));
    h->ts_table_in_lds = ts_lds_layout(h->tlen, true, h->ts_RS, h->ts_own, h->p.K, ts_wbytes(h)).total <= 160 * 1024;
    // site-range sharding: contiguous, balanced tile ranges; `reach` = tiles beyond a tile whose deposits can reach its frame
    h->ts_lo = (int)((int64_t)h->rank * h->ts_ntile / h->world);
    h->ts_hi = (int)((int64_t)(h->rank + 1) * h->ts_ntile / h->world);
    h->own_lo = h->ts_lo * h->ts_own;
    h->own_hi = std::min(L, h->ts_hi * h->ts_own);
    h->ts_reach = h->model.field_mode ? (h->tlen + 2 + h->ts_own - 1) / h->ts_own : 0;
    // steps per halo exchange: the ghost tiles a rank steps on top of its own cost next to nothing while the launch stays
    // within three workgroups per CU (one wave of workgroups: the step is latency bound there, 11 us at 256 tiles, 14.5 us
    // at 633), and at most a quarter more tiles otherwise
    h->ts_kx = 1;
    if (h->world > 1) {
        const int reach = std::max(h->ts_reach, 1);
        int own_min = h->ts_ntile, own_max = 0;
        for (int r = 0; r < h->world; ++r) {
            const int n = (int)((int64_t)(r + 1) * h->ts_ntile / h->world) - (int)((int64_t)r * h->ts_ntile / h->world);
            own_min = std::min(own_min, n); own_max = std::max(own_max, n);
        }
        auto fits = [&](int k) { return k * reach <= own_min && own_max + 2 * (k - 1) * reach <= h->ts_ntile; };
        const int64_t budget = std::max<int64_t>(768 / std::max(h->E, 1), (int64_t)own_max + own_max / 4);
        int k = 1;
        while (k < 8 && fits(k + 1) && own_max + 2 * k * reach <= budget) ++k;
        if (h->p.halo_interval > 0) k = h->p.halo_interval;
        else if (const char *env = std::getenv("APS_HALO_INTERVAL")) k = std::max(1, std::atoi(env));
        h->ts_kx = k;                                  // aps_create refuses an interval that does not fit
    }
}

TileArgs tile_args(aps_handle *h, bool field_only) {
    TileArgs a{};
    const int par = (int)(h->step & 1), out = field_only ? par : par ^ 1;
    a.L = h->p.L; a.K = h->p.K; a.tlen = h->tlen; a.own = h->ts_own; a.ntile = h->ts_ntile; a.dcap = h->ts_dcap; a.par = par;
    a.dense = h->ntt_on ? h->d_ntt_csig : nullptr; a.dense_rt = h->ntt.Rt; a.dense_m = h->ntt.m; a.periodic = h->p.periodic;
    a.tile_lo = h->ts_lo; a.field_only = field_only ? 1 : 0; a.field_mode = h->model.field_mode; a.ens_base = h->model.ens_base; a.E = h->E;
    a.seed_lo = h->model.seed_lo; a.seed_hi = h->model.seed_hi;
    a.model = h->d_model; a.rare = h->d_rare;
    a.ws_in = h->f32 ? (const void *)h->d_wsi[par] : (const void *)h->d_wsb[par];
    a.ws_out = h->f32 ? (void *)h->d_wsi[out] : (void *)h->d_wsb[out];
    a.cell_in = h->d_cell[par]; a.cell_out = h->d_cell[par ^ 1];
    a.dcnt_in = h->d_tdcnt[par]; a.dep_in = h->d_tdep[par]; a.dcnt_out = h->d_tdcnt[par ^ 1]; a.dep_out = h->d_tdep[par ^ 1];
    a.gpart_in = h->d_gpart[par]; a.gpart_out = h->d_gpart[par ^ 1];
    a.stepw = h->d_stepw; a.beta = h->d_beta; a.anchor = h->d_anchor;
    return a;
}

int launch_tile_step(aps_handle *h, bool field_only = false) {
    int rc = field_only ? APS_OK : prof_mark(h, KIND_TILE_STEP);
    if (rc) return rc;
    TileArgs a = tile_args(h, field_only);
    if (h->ntt_on) {                                            // the convolution keeps {W, S} complete: the step without any sweep (tile_dense.hpp)
        if (field_only) return fail(h, APS_ERR_STATE, "tiles, convolution: no deposits are ever pending");
        a.tile_lo = 0;
        const dim3 grid((unsigned)td_tiles(h->p.L), (unsigned)h->E), block(FU_THREADS);
        const void *fn = h->f32 ? (h->p.K == 1 ? reinterpret_cast<const void *>(&tile_dense<true, true>) : reinterpret_cast<const void *>(&tile_dense<false, true>))
                                : (h->p.K == 1 ? reinterpret_cast<const void *>(&tile_dense<true, false>) : reinterpret_cast<const void *>(&tile_dense<false, false>));
        void *args[] = {(void *)&a};
        if (h->profiling && h->prof_dispatch) HIP_TRY(h, hipExtLaunchKernel(fn, grid, block, args, td_lds_bytes(h->p.K), h->stream, h->k_start, h->k_stop, 0));
        else HIP_TRY(h, hipLaunchKernel(fn, grid, block, args, td_lds_bytes(h->p.K), h->stream));
        h->slots_dirty = true; h->field_pending = true; h->ws_view_stale = true;
        return APS_OK;
    }
    int t0 = h->ts_lo, t1 = h->ts_hi;
    if (h->world > 1) {
        // the ghost tiles whose inputs are still complete `halo_age` steps after the exchange; a flush also covers the
        // next tile on either side: it holds the two halo sites the outermost frames read
        if (!field_only && h->halo_age >= h->ts_kx) return fail(h, APS_ERR_STATE, "tiles, sharded: the halo exchange of the last step is missing");
        const int g = (h->ts_kx - 1 - std::min(h->halo_age, h->ts_kx - 1)) * h->ts_reach + (field_only ? 1 : 0);
        t0 = h->p.periodic ? t0 - g : std::max(0, t0 - g);
        t1 = h->p.periodic ? t1 + g : std::min(h->ts_ntile, t1 + g);
    }
    const void *fn = ts_kernel(h->p.periodic != 0, h->ts_table_in_lds, h->ts_RS, h->p.K == 1, h->f32);
    const size_t lds = ts_lds_layout(h->tlen, h->ts_table_in_lds, h->ts_RS, h->ts_own, h->p.K, ts_wbytes(h)).total;
    const void *table_ptr = h->f32 ? (const void *)h->d_table_i : (const void *)h->d_table;
    void *args[] = {(void *)&a, (void *)&table_ptr};
    // a range that wraps around the torus (flush of a sharded periodic handle) is launched in pieces
    struct Piece { int lo, hi; } pieces[3];
    int np = 0;
    if (t0 < 0) pieces[np++] = {t0 + h->ts_ntile, h->ts_ntile};
    pieces[np++] = {std::max(t0, 0), std::min(t1, h->ts_ntile)};
    if (t1 > h->ts_ntile) pieces[np++] = {0, t1 - h->ts_ntile};
    for (int i = 0; i < np; ++i) {
        if (pieces[i].hi <= pieces[i].lo) continue;
        a.tile_lo = pieces[i].lo;
        const dim3 grid((unsigned)(pieces[i].hi - pieces[i].lo), (unsigned)h->E), block(FU_THREADS);
        if (!field_only && h->profiling && h->prof_dispatch)
            HIP_TRY(h, hipExtLaunchKernel(fn, grid, block, args, lds, h->stream, h->k_start, h->k_stop, 0));
        else
            HIP_TRY(h, hipLaunchKernel(fn, grid, block, args, lds, h->stream));
    }
    if (!field_only) { h->slots_dirty = true; h->field_pending = true; }
    h->ws_view_stale = true;
    return APS_OK;
}

// ---- the step's deposits -> W, S of every site by ONE exact convolution (ntt_conv.hpp): five launches behind the tile kernel
int launch_ntt_conv(aps_handle *h) {
    const int out = (int)((h->step & 1) ^ 1);                   // the buffer the tile kernel of this step wrote
    const NttPlan &pl = h->ntt;
    const unsigned tiles = (unsigned)(((size_t)1 << pl.m) / NTT_TILE);
    const dim3 grid_e(tiles, 2u, (unsigned)h->E), grid_p(tiles, 2u, (unsigned)(h->E * pl.np)), block(NTT_THREADS);   // y: the two signals; z: ensembles (x primes)
    void *ws = h->f32 ? (void *)h->d_wsi[out] : (void *)h->d_wsb[out];
    int rc;
    const bool timed = h->profiling && h->prof_dispatch;
    // the first and the last sweep take every prime inside the workgroup (grid over the ensembles), the launches in between one prime per workgroup
#define NTT_STRIDED(AXIS, INV, NP_, GRID, A_, CSIG, WS, FLAG) do { if ((rc = prof_mark(h, KIND_NTT))) return rc; \
        ntt_launch_strided<AXIS, INV, NP_>(A_, GRID, block, h->stream, h->k_start, h->k_stop, timed, pl, h->d_ntt_sig, CSIG, WS, FLAG); } while (0)
#define NTT_ENDS(AXIS, INV, A_, CSIG, WS) do { if (pl.np == 2) NTT_STRIDED(2, INV, 2, grid_e, A_, CSIG, WS, 1); else NTT_STRIDED(AXIS, INV, 1, grid_e, A_, CSIG, WS, 1); } while (0)
    if (h->ntt_fused) {                                         // i2 sweep, the whole middle in one launch, i2 sweep back
        NTT_ENDS(2, false, pl.a2, h->d_ntt_csig, nullptr);
        if ((rc = prof_mark(h, KIND_NTT))) return rc;
        APS_K(h, ntt_mid, dim3(1u << pl.a2, 2u, (unsigned)(h->E * pl.np)), dim3(NTT_MID_THREADS), NTT_MID_LDS, pl, h->d_ntt_sig);
        NTT_ENDS(2, true, pl.a2, (int *)nullptr, ws);
        HIP_TRY(h, hipGetLastError());
        h->field_pending = false;
        return APS_OK;
    }
    if (pl.a2 > 0) {
        NTT_ENDS(2, false, pl.a2, h->d_ntt_csig, nullptr);
        NTT_STRIDED(1, false, 1, grid_p, pl.a1, (int *)nullptr, nullptr, 0);
    } else NTT_ENDS(1, false, pl.a1, h->d_ntt_csig, nullptr);
    if ((rc = prof_mark(h, KIND_NTT))) return rc;
    APS_K(h, (ntt_contig<false>), grid_p, block, 0, pl, h->d_ntt_sig);
    if (pl.a2 > 0) {
        NTT_STRIDED(1, true, 1, grid_p, pl.a1, (int *)nullptr, nullptr, 0);
        NTT_ENDS(2, true, pl.a2, (int *)nullptr, ws);
    } else NTT_ENDS(1, true, pl.a1, (int *)nullptr, ws);
#undef NTT_ENDS
#undef NTT_STRIDED
    HIP_TRY(h, hipGetLastError());
    h->field_pending = false;                                   // nothing is left in deposit lists: ws[out] is the field of the new cells
    return APS_OK;
}

// eligibility and tables of the convolution; called from aps_create once the geometry and the (integer) table exist
int ntt_setup(aps_handle *h) {
    h->ntt_on = false;
    const char *env = std::getenv("APS_NTT");
    if (env && env[0] == '0') return APS_OK;
    const bool forced = env && env[0] == '1';
    if (!is_tiles(h) || !h->model.field_mode || h->world != 1) return APS_OK;
    if (h->ts_table_in_lds && !forced) return APS_OK;           // the in-LDS sweep (and the resident loop) is faster for short tables
    const int Rt = h->tlen - 1, L = h->p.L;
    // one image per deposit at most: between walls the table must be short of half the box; on a torus (ring-wide table, Rt <= L / 2: a
    // deposit within Rt of either end of [0, L) is entered a second time one period on) the box must be far longer than a frame of tile_dense
    if (h->p.periodic ? !(2 * Rt <= L && L >= 4 * TD_SITES) : !(2 * Rt + 64 * h->ts_RS + h->ts_own + 4 < L)) return APS_OK;
    int m = 14;
    while (((int64_t)1 << m) < (int64_t)L + 2 * Rt) ++m;
    const int np = h->f32 ? 1 : 2;                              // the binary64 field: two primes, put together by the last sweep
    if (m > 21 || (np == 2 && m < 15) || td_lds_bytes(h->p.K) > 160 * 1024) return APS_OK;
    if (td_lds_bytes(h->p.K) > 48 * 1024) {                     // frames of tile_dense with many cells per site
        const void *fn = h->f32 ? reinterpret_cast<const void *>(&tile_dense<false, true>) : reinterpret_cast<const void *>(&tile_dense<false, false>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)td_lds_bytes(h->p.K)) != hipSuccess) { (void)hipGetLastError(); return APS_OK; }
    }
    // exactness: |dW|, |dS| <= 2 K sum_d w(d) in grid units must stay below half the modulus (32-bit field: P0; binary64: P0 P1 = 2^61.7)
    double wsum = 0.0;
    for (int t = 0; t < h->tlen; ++t) wsum += std::ldexp(h->table[(size_t)t], h->q) * (t ? 2.0 : 1.0);
    if (2.0 * h->p.K * wsum >= 0.5 * (np == 1 ? (double)NTT_PRIMES[0] : (double)NTT_PRIMES[0] * (double)NTT_PRIMES[1])) return APS_OK;
    NttPlan &pl = h->ntt;
    pl.m = m; ntt_split(m, pl.a0, pl.a1, pl.a2); pl.L = L; pl.Rt = Rt; pl.E = h->E; pl.np = np;
    pl.crt_inv = ntt_powmod(NTT_PRIMES[0] % NTT_PRIMES[1], NTT_PRIMES[1] - 2ull, NTT_PRIMES[1]);
    pl.unit = std::ldexp(1.0, -h->q);
    const size_t M = (size_t)1 << m;
    NttTables T[2];
    for (int k = 0; k < np; ++k) ntt_build_tables(m, NTT_PRIMES[k], NTT_ROOTS[k], T[k]);
    const size_t o_wr = 0, o_t1 = o_wr + T[0].wr.size(), o_hi = o_t1 + T[0].t1.size(), o_lo = o_hi + T[0].t2hi.size(), o_what = o_lo + T[0].t2lo.size(),
                 o_whatp = o_what + M, per_prime = o_whatp + M;
    int rc;
    if ((rc = dev_alloc(h, &h->d_ntt_tab, per_prime * np)) || (rc = dev_alloc(h, &h->d_ntt_sig, (size_t)np * h->E * 2 * M)) ||
        (rc = dev_alloc(h, &h->d_ntt_csig, (size_t)h->E * 2 * M))) return rc;
    for (int k = 0; k < np; ++k) {
        uint32_t *tb = h->d_ntt_tab + (size_t)k * per_prime;
        HIP_TRY(h, hipMemcpyAsync(tb + o_wr, T[k].wr.data(), T[k].wr.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(tb + o_t1, T[k].t1.data(), T[k].t1.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(tb + o_hi, T[k].t2hi.data(), T[k].t2hi.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(tb + o_lo, T[k].t2lo.data(), T[k].t2lo.size() * 4, hipMemcpyHostToDevice, h->stream));
        NttPrime &pp = pl.pr[k];
        pp.P = NTT_PRIMES[k]; pp.md.P = (double)NTT_PRIMES[k]; pp.md.Pinv = 1.0 / (double)NTT_PRIMES[k];
        pp.wr = tb + o_wr; pp.t1 = tb + o_t1; pp.t2hi = tb + o_hi; pp.t2lo = tb + o_lo; pp.what = tb + o_what; pp.whatp = tb + o_whatp;
    }
    if (np == 1) pl.pr[1] = pl.pr[0];
    // the three middle launches as one (ntt_mid): 128 x 128 slabs, at least two of them; APS_NTT_FUSED=0 keeps the five launches
    const char *fenv = std::getenv("APS_NTT_FUSED");
    h->ntt_fused = pl.a0 == 7 && pl.a1 == 7 && pl.a2 >= 1 && !(fenv && fenv[0] == '0');
    if (h->ntt_fused && hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_mid), hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_MID_LDS) != hipSuccess) {
        (void)hipGetLastError(); h->ntt_fused = false;
    }
    // spectrum of the table mod every prime: w(|d|) (an integer in grid units) at index d mod M, forward sweeps only, times 1 / M.
    // One signal per prime in the layout of a plan with one ensemble ([prime][1][W | S][M]: the S halves stay unused).
    {
        NttPlan ps = pl;
        ps.E = 1;
        std::vector<uint32_t> wext((size_t)np * 2 * M, 0u);
        for (int k = 0; k < np; ++k)
            for (int t = 0; t < h->tlen; ++t) {
                const uint32_t v = (uint32_t)((unsigned long long)std::ldexp(h->table[(size_t)t], h->q) % NTT_PRIMES[k]);
                wext[(size_t)k * 2 * M + (size_t)t] = v;
                // (torus of even length: the tap at distance L / 2 counts once -- a site sees a deposit half a ring away either directly
                //  or through its image; the kernel keeps +L / 2 and drops -L / 2)
                if (t && !(h->p.periodic && 2 * t == L)) wext[(size_t)k * 2 * M + M - (size_t)t] = v;
            }
        HIP_TRY(h, hipMemcpyAsync(h->d_ntt_sig, wext.data(), wext.size() * 4, hipMemcpyHostToDevice, h->stream));
        const dim3 grid((unsigned)(M / NTT_TILE), 1u, (unsigned)np), block(NTT_THREADS);
        if (pl.a2 > 0) ntt_launch_strided<2, false, 1>(pl.a2, grid, block, h->stream, nullptr, nullptr, false, ps, h->d_ntt_sig, nullptr, nullptr, 0);
        ntt_launch_strided<1, false, 1>(pl.a1, grid, block, h->stream, nullptr, nullptr, false, ps, h->d_ntt_sig, nullptr, nullptr, 0);
        hipLaunchKernelGGL((ntt_contig<true>), grid, block, 0, h->stream, ps, h->d_ntt_sig);
        HIP_TRY(h, hipGetLastError());
        std::vector<uint32_t> spec(M), specp;
        for (int k = 0; k < np; ++k) {
            HIP_TRY(h, hipMemcpyAsync(spec.data(), h->d_ntt_sig + (size_t)k * 2 * M, M * 4, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            const uint32_t P = NTT_PRIMES[k], minv = ntt_powmod((uint32_t)(M % P), P - 2ull, P);
            for (size_t i = 0; i < M; ++i) spec[i] = ntt_mulmod_u64(spec[i], minv, P);
            HIP_TRY(h, hipMemcpy(h->d_ntt_tab + (size_t)k * per_prime + o_what, spec.data(), M * 4, hipMemcpyHostToDevice));
            if (h->ntt_fused) {                                 // the spectrum in the slots' order: slot s holds frequency brev(s)
                specp.resize(M);
                auto brev7 = [](size_t v) { size_t r = 0; for (int b = 0; b < 7; ++b) r |= ((v >> b) & 1) << (6 - b); return r; };
                for (size_t k2 = 0; k2 < (M >> 14); ++k2)
                    for (size_t s0 = 0; s0 < 128; ++s0)
                        for (size_t s1 = 0; s1 < 128; ++s1) specp[(k2 << 14) + s0 * 128 + s1] = spec[(k2 << 14) + brev7(s1) * 128 + brev7(s0)];
                HIP_TRY(h, hipMemcpy(h->d_ntt_tab + (size_t)k * per_prime + o_whatp, specp.data(), M * 4, hipMemcpyHostToDevice));
            }
        }
    }
    HIP_TRY(h, hipMemsetAsync(h->d_ntt_sig, 0, (size_t)np * h->E * 2 * M * 4, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // {W, S} need no second buffer: the tile kernel only reads them and the last sweep of the convolution updates them in place
    if (h->d_wsi[1] && h->d_wsi[1] != h->d_wsi[0]) { (void)hipFree(h->d_wsi[1]); h->d_wsi[1] = h->d_wsi[0]; }
    if (!h->f32 && h->d_wsb[1] && h->d_wsb[1] != h->d_wsb[0]) { (void)hipFree(h->d_wsb[1]); h->d_wsb[1] = h->d_wsb[0]; }
    h->ntt_on = true;
    return APS_OK;
}

// ---- resident loop (tile_loop.hpp): when every tile of the grid is resident at once, aps_step runs its steps inside ONE
// launch.  Eligible: one rank, table in LDS, a local field, no exits, the whole grid within the kernel's residency on this
// device.  A call that gives up (a wait ran out: the workgroups were not all resident after all) leaves the inputs intact,
// is repeated with one launch per step, and the loop is not tried again on that handle.
template <bool F32>
const void *tl_kernel_f(bool periodic, int RS, bool k1) {
#define TL_PICK(BC, R) (k1 ? (const void *)&tile_loop<BC, R, true, F32> : (const void *)&tile_loop<BC, R, false, F32>)
#define TL_CASE(R) case R: return periodic ? TL_PICK(1, R) : TL_PICK(0, R);
    // binary64 frames that do not fit 256 VGPRs (they would spill to scratch) are not built: 8 x 64 sites, and 7 x 64 with K > 1 -- such handles
    // step with one launch per step ("no kernel for this frame")
    if (!F32 && (RS == 8 || (RS == 7 && !k1))) return nullptr;
#ifdef APS_DEV_RS
    switch (RS) { TL_CASE(APS_DEV_RS) default: return nullptr; }
#else
    if constexpr (F32) { if (RS == 8) return periodic ? TL_PICK(1, 8) : TL_PICK(0, 8); }
    switch (RS) {
        TL_CASE(1) TL_CASE(2) TL_CASE(3) TL_CASE(4) TL_CASE(5) TL_CASE(6)
        case 7: return k1 ? (periodic ? (const void *)&tile_loop<1, 7, true, F32> : (const void *)&tile_loop<0, 7, true, F32>)
                          : (F32 ? (periodic ? (const void *)&tile_loop<1, 7, false, true> : (const void *)&tile_loop<0, 7, false, true>) : nullptr);
        default: return nullptr;
    }
#endif
#undef TL_CASE
#undef TL_PICK
}
const void *tl_kernel(const aps_handle *h) {
    return h->f32 ? tl_kernel_f<true>(h->p.periodic != 0, h->ts_RS, h->p.K == 1) : tl_kernel_f<false>(h->p.periodic != 0, h->ts_RS, h->p.K == 1);
}

int loop_prepare(aps_handle *h) {
    if (h->loop_state != -2) return APS_OK;
    h->loop_state = 0;
    auto no = [&](const char *why) { h->loop_why = why; return APS_OK; };
    if (!is_tiles(h)) return no("not the tiles formulation");
    if (h->world != 1) return no("sharded handle");
    if (h->ntt_on) return no("field updated by the exact convolution");
    if (!h->ts_table_in_lds) return no("weight table beyond LDS");
    if (!h->model.field_mode) return no("global mean field");
    // K = 1: nothing ever binds (an anchor binds into free capacity next to the particle itself), so nothing leaves either -- unless
    // the caller's state already holds bound particles; tile_loop's K = 1 path carries no exit code
    if (h->p.K == 1 && h->model.immobilize && h->model.k_exit > 0.0) return no("particles can leave the system (one cell per site)");
    if (3 * h->p.K > 32) return no("site capacity above 10");
    if (h->p.L - (int64_t)(h->ts_ntile - 1) * h->ts_own < 3) return no("last tile shorter than three sites");
#ifdef APS_STAMPS
    return no("diagnostic build");
#endif
    const void *fn = tl_kernel(h);
    if (!fn) return no("no kernel for this frame");
    // The API can promise one block more than the hardware admits (MI355X_MICROARCH.md, residency).  Measured here: with
    // 54 040 B of LDS per block the API said 3 per CU and only 2 were ever resident -- LDS is handed out in 128 granules of
    // 1 280 B (160 KB / 128), so a block takes ceil(bytes / 1280) of them; and at most 8 blocks of 256 threads per CU.
    // Deposit segments: the longest (of 256, 192, 128, 88 entries) that still lets the whole grid be resident.
    const int64_t need_per_cu = ((int64_t)h->ts_ntile * h->E + h->num_cu - 1) / h->num_cu;
    h->loop_seg = 0;
    size_t lds_total = 0;
    for (int seg : {256, 192, 128, TL_SEG_MIN}) {
        const TlLds lay = tl_lds_layout(h->tlen, h->ts_RS, h->ts_own, h->p.K, ts_wbytes(h), seg);
        if (lay.total > 160 * 1024) continue;
        if (128 / (int)((lay.total + 1279) / 1280) >= need_per_cu) { h->loop_seg = seg; lds_total = lay.total; break; }
    }
    if (!h->loop_seg) return no("more tiles than the device keeps resident at once (LDS)");
    if (lds_total > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total) != hipSuccess) { (void)hipGetLastError(); return no("hipFuncSetAttribute failed"); }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, FU_THREADS, lds_total) != hipSuccess) { (void)hipGetLastError(); return no("occupancy query failed"); }
    if (need_per_cu > std::min(per_cu, 8)) return no("more tiles than the device keeps resident at once");
    h->loop_drec = (h->ts_dcap / 16 + 1) * 16;
    h->loop_rec = (h->loop_drec + 6 * h->p.K + 15) / 16 * 16;
    int rc;
    if ((rc = dev_alloc(h, &h->d_xrec, (size_t)2 * h->E * h->ts_ntile * h->loop_rec)) || (rc = dev_alloc(h, &h->d_abort, 4))) return rc;
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_abort), 64, hipHostMallocMapped));
    h->h_abort[0] = 0u;
    HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&h->h_abort_dev), h->h_abort, 0));
    h->loop_tag = 0;
    h->loop_state = 1;
    return APS_OK;
}

// n steps from the current state in one launch; the caller synchronises and looks at h_abort
int launch_tile_loop(aps_handle *h, int64_t n) {
    LoopArgs la{};
    la.a = tile_args(h, false);
    la.a.tile_lo = 0;
    la.step0 = (unsigned long long)h->step;
    la.nsteps = (int)n;
    if ((uint64_t)h->loop_tag + (uint64_t)n + 2 > 0xFFFFFFF0ull) {            // tags must never repeat: start over on clean records
        HIP_TRY(h, hipMemsetAsync(h->d_xrec, 0, (size_t)2 * h->E * h->ts_ntile * h->loop_rec * sizeof(unsigned long long), h->stream));
        h->loop_tag = 0;
    }
    la.tag0 = h->loop_tag;
    h->loop_tag += (uint32_t)n + 1u;
    la.rec = h->loop_rec; la.drec = h->loop_drec; la.xrec = h->d_xrec; la.seg = h->loop_seg;
    la.abort_dev = h->d_abort; la.abort_host = h->h_abort_dev;
    la.timeout_ticks = 5000000ull;                                             // 50 ms of the 100 MHz clock per wait (a hand-off takes microseconds)
    if (const char *env = std::getenv("APS_LOOP_TIMEOUT_MS")) la.timeout_ticks = (unsigned long long)std::max(1, std::atoi(env)) * 100000ull;
#ifdef APS_LOOP_DEBUG
    static uint32_t *dbg = nullptr;
    if (!dbg) (void)hipMalloc(reinterpret_cast<void **>(&dbg), (size_t)64 * 3 * h->p.L * 4);
    (void)hipMemsetAsync(dbg, 0xEE, (size_t)64 * 3 * h->p.L * 4, h->stream);
    la.dbg = n <= 64 ? dbg : nullptr;
    h->loop_dbg = dbg; h->loop_dbg_n = (int)n;
#endif
    if (std::getenv("APS_LOOP_TEST_ABORT")) HIP_TRY(h, hipMemsetAsync(h->d_abort, 1, 4, h->stream));   // tests: the call gives up at once
    {   // tests: "<tile>:<iteration>" -- that tile (of ensemble 0) leaves at the top of that iteration without writing its record and
        // without raising the give-up word, like a workgroup that never became resident: its neighbours' waits run out mid-loop
        unsigned stall = 0u;
        if (const char *env = std::getenv("APS_LOOP_TEST_STALL")) {
            int st_tile = -1, st_it = 0;
            if (std::sscanf(env, "%d:%d", &st_tile, &st_it) == 2 && st_tile >= 0 && st_tile < h->ts_ntile && st_it >= 0 && st_it < 0xFFFF)
                stall = ((unsigned)(st_tile + 1) << 16) | (unsigned)st_it;
        }
        if (stall != h->loop_stall) {
            HIP_TRY(h, hipMemcpyAsync(h->d_abort + 1, &stall, 4, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));       // (the source is a stack variable)
            h->loop_stall = stall;
        }
    }
    const void *fn = tl_kernel(h);
    const size_t lds = tl_lds_layout(h->tlen, h->ts_RS, h->ts_own, h->p.K, ts_wbytes(h), h->loop_seg).total;
    const void *table_ptr = h->f32 ? (const void *)h->d_table_i : (const void *)h->d_table;
    void *args[] = {(void *)&la, (void *)&table_ptr};
    const dim3 grid((unsigned)h->ts_ntile, (unsigned)h->E), block(FU_THREADS);
    if (h->loop_timed) {
        for (hipEvent_t &ev : h->loop_ev) if (!ev) HIP_TRY(h, hipEventCreate(&ev));
        HIP_TRY(h, hipExtLaunchKernel(fn, grid, block, args, lds, h->stream, h->loop_ev[0], h->loop_ev[1], 0));
    } else
        HIP_TRY(h, hipLaunchKernel(fn, grid, block, args, lds, h->stream));
    return APS_OK;
}

// ---- halo of a site-sharded tiles handle: what the two neighbour ranks need of this rank's freshly written buffers.
// Everything is addressed by GLOBAL index (every rank allocates the whole lattice), so a segment is received at the very
// offsets it was sent from.  side 0: this rank's FIRST sites / tiles (for the left neighbour), side 1: its LAST ones.
struct HaloSeg { size_t off, bytes; int array; };   // array: 0 cells, 1 ws, 2 dcnt, 3 dep (byte offsets into the [buf] arrays)

void halo_segments(const aps_handle *h, int owner_lo_tile, int owner_hi_tile, int side, std::vector<HaloSeg> &out) {
    const int L = h->p.L, K = h->p.K, own = h->ts_own;
    const int s_lo = owner_lo_tile * own, s_hi = std::min(L, owner_hi_tile * own);
    const int nt = std::min(h->ts_kx * h->ts_reach, owner_hi_tile - owner_lo_tile);
    // whole tiles of cells and {W, S}: the ghost zone the receiver steps itself between two exchanges
    const int64_t gs = (int64_t)(h->ts_kx - 1) * h->ts_reach * own;
    for (int e = 0; e < h->E; ++e) {
        const int64_t g1 = std::min<int64_t>(s_hi, (int64_t)owner_hi_tile * own - gs);      // first site of the last ghost tiles
        const int c0 = side == 0 ? s_lo : (int)std::max<int64_t>(s_lo, g1 - 3);
        const int c1 = side == 0 ? (int)std::min<int64_t>(s_hi, s_lo + gs + 3) : s_hi;
        const int w0 = side == 0 ? s_lo : (int)std::max<int64_t>(s_lo, g1 - 2);
        const int w1 = side == 0 ? (int)std::min<int64_t>(s_hi, s_lo + gs + 2) : s_hi;
        const int t0 = side == 0 ? owner_lo_tile : owner_hi_tile - nt, t1 = t0 + nt;
        out.push_back({((size_t)e * L + c0) * K * 4, (size_t)(c1 - c0) * K * 4, 0});
        if (h->model.field_mode) {
            out.push_back({((size_t)e * L + w0) * 2 * ts_wbytes(h), (size_t)(w1 - w0) * 2 * ts_wbytes(h), 1});
            out.push_back({((size_t)e * h->ts_ntile + t0) * 4, (size_t)(t1 - t0) * 4, 2});
            out.push_back({((size_t)e * h->ts_ntile + t0) * h->ts_dcap * 4, (size_t)(t1 - t0) * h->ts_dcap * 4, 3});
        }
    }
}

char *halo_array(aps_handle *h, int array, int buf) {
    switch (array) {
        case 0: return reinterpret_cast<char *>(h->d_cell[buf]);
        case 1: return h->f32 ? reinterpret_cast<char *>(h->d_wsi[buf]) : reinterpret_cast<char *>(h->d_wsb[buf]);
        case 2: return reinterpret_cast<char *>(h->d_tdcnt[buf]);
        default: return reinterpret_cast<char *>(h->d_tdep[buf]);
    }
}

void rank_tiles(const aps_handle *h, int r, int &lo, int &hi) {
    lo = (int)((int64_t)r * h->ts_ntile / h->world);
    hi = (int)((int64_t)(r + 1) * h->ts_ntile / h->world);
}

// the neighbours of this rank (-1: none, a reflecting wall)
void halo_peers(const aps_handle *h, int &left, int &right) {
    left = h->rank - 1; right = h->rank + 1;
    if (h->p.periodic) { left = (left + h->world) % h->world; right %= h->world; }
    else { if (right >= h->world) right = -1; }
}

// segment tables of the four blocks this rank packs / unpacks, on the device (built once: they do not depend on the step)
int halo_setup(aps_handle *h) {
    int left, right, lo, hi;
    halo_peers(h, left, right);
    for (int side = 0; side < 2; ++side) {
        for (int recv = 0; recv < 2; ++recv) {
            std::vector<HaloSeg> segs;
            if (!recv) halo_segments(h, h->ts_lo, h->ts_hi, side, segs);
            else {
                const int peer = side == 0 ? right : left;       // recv[0]: the right neighbour's first block, recv[1]: the left one's last
                if (peer < 0) continue;
                rank_tiles(h, peer, lo, hi);
                halo_segments(h, lo, hi, side, segs);
            }
            std::vector<HaloSegD> tab;
            size_t off = 0;
            for (const HaloSeg &g : segs) {
                if (!g.bytes) continue;
                tab.push_back({(unsigned long long)g.off, (unsigned long long)off, (unsigned)g.bytes, g.array});
                off += (g.bytes + 15) / 16 * 16;
            }
            HaloSegD **dseg = recv ? &h->d_halo_seg_recv[side] : &h->d_halo_seg_send[side];
            char **dmsg = recv ? &h->d_halo_recv[side] : &h->d_halo_send[side];
            (recv ? h->halo_nseg_recv[side] : h->halo_nseg_send[side]) = (int)tab.size();
            (recv ? h->halo_bytes_recv[side] : h->halo_bytes_send[side]) = off;
            int rc;
            if ((rc = dev_alloc(h, dseg, tab.size())) || (rc = dev_alloc(h, dmsg, off))) return rc;
            HIP_TRY(h, hipMemcpyAsync(*dseg, tab.data(), tab.size() * sizeof(HaloSegD), hipMemcpyHostToDevice, h->stream));
        }
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return APS_OK;
}

// pack this rank's block `side` of the freshly written buffers into its message / unpack a received message
int halo_launch(aps_handle *h, int side, bool unpack) {
    const int buf = (int)((h->step & 1) ^ 1);
    const int nseg = unpack ? h->halo_nseg_recv[side] : h->halo_nseg_send[side];
    if (!nseg) return APS_OK;
    hipLaunchKernelGGL(halo_move, dim3((unsigned)(nseg * HALO_BPS)), dim3(256), 0, h->stream,
                       unpack ? h->d_halo_seg_recv[side] : h->d_halo_seg_send[side], halo_array(h, 0, buf), halo_array(h, 1, buf),
                       halo_array(h, 2, buf), halo_array(h, 3, buf), unpack ? h->d_halo_recv[side] : h->d_halo_send[side], unpack ? 1 : 0);
    HIP_TRY(h, hipGetLastError());
    return APS_OK;
}

// after the tile kernel of a step: one packed message to each neighbour rank, one from each (ncclSend / ncclRecv in one group)
// the neighbour blocks a due exchange must bring in (bit 0: the right neighbour's first block, bit 1: the left one's last)
unsigned halo_expected(const aps_handle *h) {
    int left, right;
    halo_peers(h, left, right);
    return (right >= 0 && h->halo_nseg_recv[0] ? 1u : 0u) | (left >= 0 && h->halo_nseg_recv[1] ? 2u : 0u);
}

bool halo_due(const aps_handle *h) { return h->world > 1 && h->halo_age + 1 == h->ts_kx; }

int halo_exchange_rccl(aps_handle *h) {
    int left, right, rc;
    halo_peers(h, left, right);
    if (left >= 0 && (rc = halo_launch(h, 0, false))) return rc;
    if (right >= 0 && (rc = halo_launch(h, 1, false))) return rc;
    ncclResult_t nr = g_rccl.GroupStart();
    // order per pair of ranks (matters when left and right are the same rank): sends first-block then last-block,
    // receives the peer's first-block (it is my right neighbour's) then its last-block
    if (nr == ncclSuccess && left >= 0 && h->halo_bytes_send[0]) nr = g_rccl.Send(h->d_halo_send[0], h->halo_bytes_send[0], ncclUint8, left, h->comm, h->stream);
    if (nr == ncclSuccess && right >= 0 && h->halo_bytes_send[1]) nr = g_rccl.Send(h->d_halo_send[1], h->halo_bytes_send[1], ncclUint8, right, h->comm, h->stream);
    if (nr == ncclSuccess && right >= 0 && h->halo_bytes_recv[0]) nr = g_rccl.Recv(h->d_halo_recv[0], h->halo_bytes_recv[0], ncclUint8, right, h->comm, h->stream);
    if (nr == ncclSuccess && left >= 0 && h->halo_bytes_recv[1]) nr = g_rccl.Recv(h->d_halo_recv[1], h->halo_bytes_recv[1], ncclUint8, left, h->comm, h->stream);
    const ncclResult_t ge = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ge;
    if (nr != ncclSuccess) return fail(h, APS_ERR_HIP, std::string("halo exchange (ncclSend/ncclRecv): ") + g_rccl.GetErrorString(nr));
    if (right >= 0 && (rc = halo_launch(h, 0, true))) return rc;
    if (left >= 0 && (rc = halo_launch(h, 1, true))) return rc;
    h->halo_got = halo_expected(h);
    return APS_OK;
}

// the same exchange by peer stores: one push launch (both blocks), one pull launch (both blocks), nothing else
int halo_exchange_ipc(aps_handle *h) {
    int left, right;
    halo_peers(h, left, right);
    const int buf = (int)((h->step & 1) ^ 1), par = (int)(h->ipc_seq & 1u);
    const unsigned tag = h->ipc_seq + 1u;
    HaloPushArgs pu{};
    HaloPullArgs pl{};
    const int peer_of_side[2] = {left, right};
    for (int side = 0; side < 2; ++side) {
        if (peer_of_side[side] >= 0 && h->halo_nseg_send[side]) {
            pu.segs[side] = h->d_halo_seg_send[side]; pu.nseg[side] = h->halo_nseg_send[side];
            pu.dst[side] = h->ipc_peer[side] + h->ipc_peer_off[side][par];
            pu.flag[side] = reinterpret_cast<unsigned long long *>(h->ipc_peer[side] + (size_t)(par * 2 + side) * 64);
        }
        const int from = side == 0 ? right : left;               // recv[0]: the right neighbour's first block, recv[1]: the left one's last
        if (from >= 0 && h->halo_nseg_recv[side]) {
            pl.segs[side] = h->d_halo_seg_recv[side]; pl.nseg[side] = h->halo_nseg_recv[side];
            pl.src[side] = h->ipc_land + h->ipc_land_off[par][side];
            pl.flag[side] = reinterpret_cast<const unsigned long long *>(h->ipc_land + (size_t)(par * 2 + side) * 64);
        }
    }
    pu.done = h->d_ipc_done; pu.tag = tag;
    pl.tag = tag; pl.err_host = h->h_ipc_err_dev;
    pl.timeout_ticks = 20ull * 100000000ull;                     // 20 s: the neighbour may be far behind (another process)
    if (const char *env = std::getenv("APS_HALO_TIMEOUT_MS")) pl.timeout_ticks = (unsigned long long)std::max(1, std::atoi(env)) * 100000ull;
    char *a0 = halo_array(h, 0, buf), *a1 = halo_array(h, 1, buf), *a2 = halo_array(h, 2, buf), *a3 = halo_array(h, 3, buf);
    const unsigned nbu = (unsigned)(pu.nseg[0] + pu.nseg[1]) * HALO_BPS, nbl = (unsigned)(pl.nseg[0] + pl.nseg[1]) * HALO_BPS;
    if (nbu) hipLaunchKernelGGL(halo_push, dim3(nbu), dim3(256), 0, h->stream, pu, a0, a1, a2, a3);
    if (nbl) hipLaunchKernelGGL(halo_pull, dim3(nbl), dim3(256), 0, h->stream, pl, a0, a1, a2, a3);
    HIP_TRY(h, hipGetLastError());
    h->ipc_seq += 1u;
    h->halo_got = halo_expected(h);
    return APS_OK;
}

// add the deposits of the last step to W, S now (the next step would do it first thing): afterwards ws[cur] is the field
// of the current cells.  Exact arithmetic: when the deposits are added changes no bit of any later result.
int flush_field(aps_handle *h) {
    if (!is_tiles(h) || !h->field_pending) return APS_OK;
    const int cur = (int)(h->step & 1);
    if (h->model.field_mode) {
        int rc = launch_tile_step(h, true);
        if (rc) return rc;
    }
    HIP_TRY(h, hipMemsetAsync(h->d_tdcnt[cur], 0, (size_t)h->E * h->ts_ntile * 4, h->stream));
    h->field_pending = false;
    return APS_OK;
}

// bring the particle-indexed arrays (src, sp8, tinfo, occ_site, gsum) up to date with the cells
int sync_slots(aps_handle *h) {
    if (!is_tiles(h)) return APS_OK;
    int rc = flush_field(h);
    if (rc) return rc;
    const int cur = (int)(h->step & 1);
    h->d_ws = h->d_wsb[cur];
    const int L = h->p.L;
    if (h->f32 && h->model.field_mode && h->ws_view_stale) {      // the hooks read the binary64 view of the integer field
        const size_t n = (size_t)h->E * L;
        hipLaunchKernelGGL(ws_to_double, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_wsi[cur], h->d_wsb[cur], n, std::ldexp(1.0, -h->q));
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->ws_view_stale = false;
    }
    if (!h->slots_dirty) return APS_OK;
    const int s_lo = h->world > 1 ? h->own_lo : 0, s_hi = h->world > 1 ? h->own_hi : L;
    if (h->world > 1)
        hipLaunchKernelGGL(mark_away, dim3((unsigned)(((size_t)h->E * h->Npad + 255) / 256)), dim3(256), 0, h->stream, h->d_src, (size_t)h->E * h->Npad);
    if (s_hi > s_lo)
        hipLaunchKernelGGL(cells_to_slots, dim3((unsigned)((s_hi - s_lo + 255) / 256), (unsigned)h->E), dim3(256), 0, h->stream,
                           h->d_cell[cur], h->d_slot_of, h->d_src, h->d_occ_site, L, h->p.K, (long long)h->N, (int)h->Npad, s_lo, s_hi);
    hipLaunchKernelGGL(derive_slots, dim3((unsigned)(h->Npad / 256), (unsigned)h->E), dim3(256), 0, h->stream,
                       h->d_src, h->d_sp8, h->d_tinfo, (int)h->Npad, (int)h->ntiles);
    long long *gs = h->d_gsum + (size_t)cur * 2 * h->E;
    HIP_TRY(h, hipMemsetAsync(gs, 0, (size_t)2 * h->E * sizeof(long long), h->stream));
    hipLaunchKernelGGL(tile_parts, dim3((unsigned)h->ts_ntile, (unsigned)h->E), dim3(256), 0, h->stream,
                       h->d_cell[cur], h->d_gpart[cur], gs, L, h->p.K, h->ts_own, h->ts_ntile);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->slots_dirty = false;
    h->plan_dirty = true;
    return APS_OK;
}

// (re)build everything the tile kernel reads from the uploaded particles: W, S on all sites, no pending deposits,
// the per-tile parts of the global sums, the step words
int ensure_tiles(aps_handle *h) {
    if (!h->field_dirty) return APS_OK;
    const int cur = (int)(h->step & 1), L = h->p.L;
    if (h->model.field_mode)
        for (int e = 0; e < h->E; ++e) {
            int rc = launch_field(h, e, h->d_sp8 + (size_t)e * h->Npad, h->d_tinfo + (size_t)e * h->ntiles, (int)h->ntiles,
                                  nullptr, h->d_wsb[cur] + (size_t)e * L);
            if (rc) return rc;
        }
    if (h->f32 && h->model.field_mode) {
        const size_t n = (size_t)h->E * L;
        hipLaunchKernelGGL(ws_to_int, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_wsb[cur], h->d_wsi[cur], n, std::ldexp(1.0, h->q));
        h->ws_view_stale = false;
    }
    HIP_TRY(h, hipMemsetAsync(h->d_tdcnt[cur], 0, (size_t)h->E * h->ts_ntile * 4, h->stream));
    long long *gs = h->d_gsum + (size_t)cur * 2 * h->E;
    HIP_TRY(h, hipMemsetAsync(gs, 0, (size_t)2 * h->E * sizeof(long long), h->stream));
    hipLaunchKernelGGL(tile_parts, dim3((unsigned)h->ts_ntile, (unsigned)h->E), dim3(256), 0, h->stream,
                       h->d_cell[cur], h->d_gpart[cur], gs, L, h->p.K, h->ts_own, h->ts_ntile);
    HIP_TRY(h, hipGetLastError());
    const unsigned long long sw[2] = {(unsigned long long)(h->step & 1 ? h->step - 1 : h->step), (unsigned long long)(h->step & 1 ? h->step : h->step + 1)};
    HIP_TRY(h, hipMemcpyAsync(h->d_stepw, sw, sizeof(sw), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->d_ws = h->d_wsb[cur];
    h->field_dirty = false;
    h->field_pending = false;
    h->halo_age = 0;                                         // built from the whole uploaded state: every tile is complete
    h->halo_got = 0;
    return APS_OK;
}

// cells + id -> slot map of one ensemble, from the caller's arrays and the slot order pack_ensemble chose
int upload_cells(aps_handle *h, int e, const int32_t *pos, const int8_t *sigma, const uint8_t *bound, const uint8_t *alive, int64_t n,
                 const std::vector<uint32_t> &orig) {
    const int L = h->p.L, K = h->p.K, cur = (int)(h->step & 1);
    std::vector<uint32_t> cells((size_t)L * K, CELL_EMPTY), slot_of((size_t)std::max<int64_t>(h->N, 1), 0u);
    std::vector<int> fill((size_t)L, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (alive && !alive[i]) continue;
        uint32_t c = (uint32_t)i;
        if (sigma[i] > 0) c |= CELL_PLUS;
        if (bound && bound[i]) c |= CELL_BOUND;
        cells[(size_t)pos[i] * K + (size_t)fill[(size_t)pos[i]]++] = c;
    }
    for (int64_t sl = 0; sl < h->Npad; ++sl) if (orig[(size_t)sl] != 0xFFFFFFFFu) slot_of[orig[(size_t)sl]] = (uint32_t)sl;
    HIP_TRY(h, hipMemcpyAsync(h->d_cell[cur] + (size_t)e * L * K, cells.data(), cells.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_slot_of + (size_t)e * h->N, slot_of.data(), (size_t)h->N * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return APS_OK;
}

int do_propose(aps_handle *h) {
    if (is_tiles(h)) {
        int rc = launch_tile_step(h);
        if (!rc && h->ntt_on) rc = launch_ntt_conv(h);
        return rc;
    }
    if (h->method == APS_METHOD_LATTICE) {
        const LatticeArgs a = lattice_args(h, false, true);
        return launch_lattice_propose(h, a, a.tile_lo, a.tile_cnt);
    }
    const PairArgs a = pair_args(h, false, true);
    return launch_pair(h, a, a.tile_lo, a.tile_cnt);
}

int do_commit(aps_handle *h) {
    if (is_tiles(h)) {                                       // the tile kernel already wrote the new state
        if (h->world > 1) {
            if (halo_due(h)) {
                if (h->halo_got != halo_expected(h)) return fail(h, APS_ERR_STATE, "aps_commit: the halo exchange is due before this commit (aps_halo_info)");
                h->halo_age = 0;
            } else h->halo_age += 1;
            h->halo_got = 0;
        }
        h->step += 1;
        return prof_mark(h, KIND_END);
    }
    const CommitArgs c = commit_args(h);
    const dim3 grid((unsigned)(h->Npad / 256), (unsigned)h->E), block(256);
    int rc;
    if (h->world != 1) {                                     // one GPU: the propose kernel registered the hops
        if ((rc = prof_mark(h, KIND_CLAIM))) return rc;
        APS_K(h, claim, grid, block, 0, c);
    }
    if ((rc = prof_mark(h, KIND_APPLY))) return rc;
    APS_K(h, apply, grid, block, 0, c);
    HIP_TRY(h, hipGetLastError());
    h->step += 1;
    if (h->method == APS_METHOD_LATTICE) {
        h->plan_dirty = true;                                // the all-pairs hook replans when it is used
        if ((rc = launch_field_update(h))) return rc;
    } else if (++h->plan_age >= h->plan_interval) {          // source-tile lists for the following step(s)
        if ((rc = launch_plan(h, (int)(h->rank * h->SH / TILE), (int)(h->SH / TILE)))) return rc;
        h->plan_dirty = false;
    }
    return prof_mark(h, KIND_END);
}

bool all_set(const aps_handle *h) {
    for (int64_t n : h->n_set) if (n < 0) return false;
    return true;
}

}  // namespace

// ================================================================================== C ABI
namespace { int download_hook(aps_handle *h, int e, double *S, double *W, int32_t *occ4); }

extern "C" {

int aps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *aps_last_error(const aps_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int aps_create(const aps_params *p, aps_handle **out) {
    if (!p || !out) { g_create_error = "aps_create: null argument"; return APS_ERR_ARG; }
    *out = nullptr;
    auto bad = [&](const char *m) { g_create_error = std::string("aps_create: ") + m; return APS_ERR_ARG; };
    if (p->L < 2 || p->L > (1 << 25)) return bad("L must be in [2, 2^25]");
    if (p->K < 1 || p->K > 32) return bad("site capacity K must be in [1, 32]");
    if (p->n_ensembles < 1) return bad("n_ensembles must be >= 1");
    if (p->n_particles < 0 || p->n_particles > (int64_t)p->K * p->L) return bad("n_particles must be in [0, K*L]");
    if (!(p->dt > 0.0)) return bad("dt must be > 0");
    if (!p->beta) return bad("beta pointer is null");
    if (p->world < 1 || p->rank < 0 || p->rank >= p->world) return bad("bad rank/world");
    if (p->method != APS_METHOD_AUTO && p->method != APS_METHOD_PAIRS && p->method != APS_METHOD_LATTICE && p->method != APS_METHOD_TILES)
        return bad("method must be APS_METHOD_AUTO, _PAIRS, _LATTICE or _TILES");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "aps_create: no HIP device"; return APS_ERR_NODEVICE; }
    if (p->device < 0 || p->device >= ndev) return bad("device ordinal out of range");

    aps_handle *h = new aps_handle();
    h->p = *p;
    h->beta.assign(p->beta, p->beta + p->n_ensembles);
    h->p.beta = nullptr;
    h->E = p->n_ensembles; h->world = p->world; h->rank = p->rank; h->N = p->n_particles;
    // shard length: whole 256-slot groups so that apply()'s blocks and the 64-slot tiles never straddle ranks
    const int64_t per_rank = (h->N + h->world - 1) / h->world;
    h->SH = std::max<int64_t>(256, (per_rank + 255) / 256 * 256);
    h->Npad = h->SH * h->world;
    h->ntiles = h->Npad / TILE;
    h->n_set.assign((size_t)h->E, -1);
    Model &M = h->model;
    M.L = p->L; M.K = p->K; M.periodic = p->periodic ? 1 : 0; M.field_mode = p->sigma_grid > 0.0 ? 1 : 0;
    M.minus_anchor = p->minus_anchor ? 1 : 0; M.immobilize = p->immobilize ? 1 : 0;
    M.suppress_flip = p->suppress_flip ? 1 : 0; M.crowding = p->crowding ? 1 : 0;
    M.rate_diffusion = p->rate_diffusion; M.rate_active = p->rate_active; M.k_on = p->k_on; M.k_off = p->k_off;
    M.k_exit = p->k_exit; M.dt = p->dt; M.seed_lo = (uint32_t)p->seed; M.seed_hi = (uint32_t)(p->seed >> 32);
    M.ens_base = p->ensemble_base;
    build_table(h);
    // formulation: lattice (incremental field on the L sites) unless asked otherwise or its deposit lists would be huge
    {
        const int B = 1 << h->bshift;
        h->nb = (p->L + B - 1) / B;
        h->dcap = (int)std::min<int64_t>(2LL * p->K * B, std::max<int64_t>(2 * p->n_particles, 2));
        const double dep_bytes = (double)h->E * h->nb * h->dcap * 4.0;
        // tiles: site-centric state, one kernel per step (single GPU handles; ids must fit the 30-bit cell field)
        // tiles: site-centric state, one kernel per step; ids must fit the 30-bit cell field.  Sharded (world > 1): by site
        // ranges with a halo exchange -- needs the local field (the global mean would be an all-reduce) and one ensemble
        const bool tiles_ok = p->n_particles < (int64_t)CELL_ID && dep_bytes <= 16e9 && (p->world == 1 || M.field_mode);
        if (p->method == APS_METHOD_TILES && !tiles_ok) { delete h; return bad("method tiles needs fewer than 2^30 - 1 particles and, sharded, a local field (sigma_grid > 0)"); }
        h->method = p->method == APS_METHOD_AUTO ? ((tiles_ok && p->world == 1) ? APS_METHOD_TILES : (dep_bytes <= 16e9 ? APS_METHOD_LATTICE : APS_METHOD_PAIRS)) : p->method;
        // far beyond the caches (state of the binary64 field above 512 MB: N >~ 6e6 particles at half filling) the three streaming
        // kernels of the lattice formulation move their bytes faster than one tile kernel with its per-tile prologue
        // (measured at N = 1.6e7, L = 3.2e7: 0.68 ms against 0.85 ms per step; with the 32-bit field the tile kernel wins: 0.63 ms)
        if (p->method == APS_METHOD_AUTO && h->method == APS_METHOD_TILES && !p->fp32 && dep_bytes <= 16e9 &&
            (32.0 + 8.0 * p->K) * (double)p->L * (double)h->E > 512e6)
            h->method = APS_METHOD_LATTICE;
        if (const char *env = std::getenv("APS_METHOD")) {    // test / tuning knob for method = auto
            if (p->method == APS_METHOD_AUTO && !std::strcmp(env, "pairs")) h->method = APS_METHOD_PAIRS;
            if (p->method == APS_METHOD_AUTO && !std::strcmp(env, "lattice")) h->method = APS_METHOD_LATTICE;
            if (p->method == APS_METHOD_AUTO && !std::strcmp(env, "tiles") && tiles_ok) h->method = APS_METHOD_TILES;
        }
        h->f32 = p->fp32 != 0 && h->method == APS_METHOD_TILES;
        if (p->fp32 && M.field_mode && h->q < 4) { delete h; return bad("fp32: the sums of this lattice do not fit a 32-bit field (q < 4)"); }
        ts_choose_geometry(h);
        // sites per lane of field_update: the largest tile that still gives about two workgroups per CU
        // sites per lane of field_update (tile = 64 * RS sites per workgroup): about 2.4 workgroups per CU was the
        // fastest grid on MI355X (measured, RS = 5 at L = 2e5); large lattices take the largest tile (fewest table copies)
        h->fu_R = 2;
        double best = 1e300;
        for (int rs : {2, 3, 4, 5, 6, 7, 8}) {
            const double wgs = (double)(((int64_t)p->L + 64 * rs - 1) / (64 * rs)) * h->E;
            const double miss = std::fabs(wgs - 2.4 * 256.0);
            if (miss < best) { best = miss; h->fu_R = rs; }
        }
        // grids of many workgroups per CU (the streaming regime): 7 x 64 sites per workgroup (measured at N = 1.6e7, L = 3.2e7, whole step:
        // R = 2 .. 8 -> 786, 664, 641, 697, 656, 630, 681 us)
        if ((double)(((int64_t)p->L + 64 * 8 - 1) / (64 * 8)) * h->E > 8.0 * 256.0 && h->tlen <= 4096) h->fu_R = 7;
        if (const char *env = std::getenv("APS_FU_R")) { const int r = std::atoi(env); if (r >= 2 && r <= 8) h->fu_R = r; }
    }
    // A particle moves at most one site per step, so tile bounds drift by <= 1 per step: when nobody can die
    // (has-dead flags are then static) or wrap around, a plan with a 16-site margin serves 8 steps.
    h->plan_interval = (!p->periodic && !(p->k_exit > 0.0)) ? 8 : 1;
    if (const char *env = std::getenv("APS_PLAN_INTERVAL")) h->plan_interval = std::max(1, std::atoi(env));   // tuning knob

    auto die = [&](int code) { g_create_error = h->err; aps_destroy(h); return code; };
    if (hipSetDevice(p->device) != hipSuccess) { h->err = "hipSetDevice failed"; return die(APS_ERR_HIP); }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return die(APS_ERR_HIP); }
    h->own_stream = true;
    int rc = set_lds_limit(h);
    if (rc) return die(rc);
    const size_t EN = (size_t)h->E * (size_t)h->Npad, EL = (size_t)h->E * (size_t)p->L;
    h->exit_cap = (int)std::max<int64_t>(h->N, 1);
    if ((rc = dev_alloc(h, &h->d_src, EN)) || (rc = dev_alloc(h, &h->d_orig, EN)) || (rc = dev_alloc(h, &h->d_prop_own, EN)) ||
        (rc = dev_alloc(h, &h->d_sp8, EN)) ||
        (rc = dev_alloc(h, &h->d_tinfo, (size_t)h->E * h->ntiles)) ||
        (rc = dev_alloc(h, &h->d_stamps, (size_t)8 * 4096)) ||
        (rc = dev_alloc(h, &h->d_plan, (size_t)h->E * h->ntiles * PLAN_CAP)) || (rc = dev_alloc(h, &h->d_plan_n, (size_t)h->E * h->ntiles)) ||
        (rc = dev_alloc(h, &h->d_accW, EN * MAX_SPLIT)) || (rc = dev_alloc(h, &h->d_accS, EN * MAX_SPLIT)) ||
        (rc = dev_alloc(h, &h->d_occ, EN * MAX_SPLIT)) || (rc = dev_alloc(h, &h->d_pcnt, 2 * EL)) ||
        (rc = dev_alloc(h, &h->d_plist, EL * 2 * p->K)) || (rc = dev_alloc(h, &h->d_table, h->table.size() + ((size_t)1 << h->bshift) + 8192))   /* zeros behind the table: windows and LDS staging run past its end */ ||
        (rc = dev_alloc(h, &h->d_beta, (size_t)h->E)) || (rc = dev_alloc(h, &h->d_gsum, (size_t)4 * h->E)) ||
        (rc = dev_alloc(h, &h->d_exit, (size_t)h->E * h->exit_cap * 3)) || (rc = dev_alloc(h, &h->d_nexit, (size_t)h->E)) ||
        (rc = dev_alloc(h, &h->d_mfield, (size_t)p->L)))
        return die(rc);
    h->d_prop = h->d_prop_own;
    if (h->method == APS_METHOD_LATTICE) {
        if ((rc = dev_alloc(h, &h->d_ws, EL)) || (rc = dev_alloc(h, &h->d_occ_site, EL)) ||
            (rc = dev_alloc(h, &h->d_dcnt, (size_t)h->E * h->nb)) || (rc = dev_alloc(h, &h->d_dep, (size_t)h->E * h->nb * h->dcap)) ||
            (rc = dev_alloc(h, &h->d_stepw, 2)))
            return die(rc);
    }
    if (h->method == APS_METHOD_TILES && h->world > 1 && (h->ts_kx < 1 || h->ts_kx * std::max(h->ts_reach, 1) > h->ts_ntile / h->world ||
                                                         (h->ts_ntile + h->world - 1) / h->world + 2 * (h->ts_kx - 1) * h->ts_reach > h->ts_ntile)) {
        h->err = "tiles, sharded: halo_interval " + std::to_string(h->ts_kx) + " x reach " + std::to_string(h->ts_reach) +
                 " tiles does not fit a rank's tile range (" + std::to_string(h->ts_ntile / h->world) + " tiles)";
        return die(APS_ERR_ARG);
    }
    if (h->method == APS_METHOD_TILES && h->world > 1 && 2 * h->ts_reach + 1 > h->ts_ntile / h->world) {
        h->err = "tiles, sharded: the table's reach (" + std::to_string(h->ts_reach) + " tiles) must stay below half a rank's tile range (" +
                 std::to_string(h->ts_ntile / h->world) + " tiles): fewer ranks or a larger lattice";
        return die(APS_ERR_ARG);
    }
    if (h->method == APS_METHOD_TILES) {
        const size_t ET = (size_t)h->E * h->ts_ntile;
        if ((rc = dev_alloc(h, &h->d_occ_site, EL)) || (rc = dev_alloc(h, &h->d_stepw, 2)) ||
            (rc = dev_alloc(h, &h->d_slot_of, (size_t)h->E * std::max<int64_t>(h->N, 1))))
            return die(rc);
        for (int b = 0; b < 2; ++b) {
            if ((rc = dev_alloc(h, &h->d_wsb[b], EL)) || (rc = dev_alloc(h, &h->d_cell[b], EL * p->K)) ||
                (rc = dev_alloc(h, &h->d_tdcnt[b], ET)) || (rc = dev_alloc(h, &h->d_tdep[b], ET * h->ts_dcap)) ||
                (rc = dev_alloc(h, &h->d_gpart[b], ET * 2)))
                return die(rc);
            if (hipMemsetAsync(h->d_cell[b], 0xFF, EL * p->K * 4, h->stream) != hipSuccess) { h->err = "cell init failed"; return die(APS_ERR_HIP); }
        }
        if ((rc = dev_alloc(h, &h->d_model, 1)) || (rc = dev_alloc(h, &h->d_rare, 1))) return die(rc);
        {
            const TileRare rare{h->d_exit, h->d_nexit, h->d_src, h->d_slot_of, h->d_stamps, (long long)h->N, h->exit_cap, (int)h->Npad, h->ts_lo, h->ts_hi};
            if (hipMemcpyAsync(h->d_model, &h->model, sizeof(Model), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipMemcpyAsync(h->d_rare, &rare, sizeof(TileRare), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "tile argument upload failed"; return die(APS_ERR_HIP); }
        }
        if (h->f32) {
            std::vector<int> ti(h->table.size());
            for (size_t i = 0; i < ti.size(); ++i) ti[i] = (int)std::ldexp(h->table[i], h->q);      // exact: multiples of 2^-q below 2^29
            if ((rc = dev_alloc(h, &h->d_table_i, h->table.size() + ((size_t)1 << h->bshift) + 8192)) ||
                (rc = dev_alloc(h, &h->d_wsi[0], EL)) || (rc = dev_alloc(h, &h->d_wsi[1], EL))) return die(rc);
            if (hipMemcpyAsync(h->d_table_i, ti.data(), ti.size() * sizeof(int), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "integer table upload failed"; return die(APS_ERR_HIP); }
        }
        if ((rc = ntt_setup(h))) return die(rc);
        if (h->world > 1 && (rc = halo_setup(h))) return die(rc);
        const size_t need = ts_lds_layout(h->tlen, h->ts_table_in_lds, h->ts_RS, h->ts_own, p->K, ts_wbytes(h)).total;
        if (need > 160 * 1024) { h->err = "tiles: site capacity too large for the tile kernel's LDS staging"; return die(APS_ERR_ARG); }
        if (need > 48 * 1024 &&
            hipFuncSetAttribute(ts_kernel(p->periodic != 0, h->ts_table_in_lds, h->ts_RS, p->K == 1, h->f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) != hipSuccess) {
            h->err = "hipFuncSetAttribute(tile_step) failed"; return die(APS_ERR_HIP);
        }
    }
    if (p->anchor_mask) {
        if ((rc = dev_alloc(h, &h->d_anchor, (size_t)p->L))) return die(rc);
        if (hipMemcpyAsync(h->d_anchor, p->anchor_mask, (size_t)p->L, hipMemcpyHostToDevice, h->stream) != hipSuccess) { h->err = "anchor upload failed"; return die(APS_ERR_HIP); }
    }
    h->p.anchor_mask = nullptr;
    if (hipMemcpyAsync(h->d_table, h->table.data(), h->table.size() * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(h->d_beta, h->beta.data(), h->beta.size() * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "table upload failed"; return die(APS_ERR_HIP); }
    *out = h;
    return APS_OK;
}

void aps_destroy(aps_handle *h) {
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    for (hipEvent_t ev : h->events) (void)hipEventDestroy(ev);
    drop_graphs(h);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->method == APS_METHOD_TILES) h->d_ws = nullptr;   // an alias of d_wsb[cur] there
    for (int b = 0; b < 2; ++b)
        for (void *q : {(void *)(b == 1 && h->d_wsb[1] == h->d_wsb[0] ? nullptr : h->d_wsb[b]), (void *)h->d_cell[b], (void *)h->d_tdcnt[b], (void *)h->d_tdep[b], (void *)h->d_gpart[b]}) if (q) (void)hipFree(q);
    if (h->h_abort) (void)hipHostFree(h->h_abort);
    if (h->d_flip_tab) (void)hipFree(h->d_flip_tab);
    for (void *q : {(void *)h->d_ntt_sig, (void *)h->d_ntt_tab, (void *)h->d_ntt_csig}) if (q) (void)hipFree(q);
    for (void *q : h->ipc_opened) if (q) (void)hipIpcCloseMemHandle(q);
    if (h->ipc_land) (void)hipFree(h->ipc_land);
    if (h->d_ipc_done) (void)hipFree(h->d_ipc_done);
    if (h->h_ipc_err) (void)hipHostFree(h->h_ipc_err);
    for (hipEvent_t ev : h->loop_ev) if (ev) (void)hipEventDestroy(ev);
    for (void *q : {(void *)h->d_xrec, (void *)h->d_abort}) if (q) (void)hipFree(q);
    for (void *q : {(void *)h->d_slot_of, (void *)h->d_model, (void *)h->d_rare, (void *)h->d_table_i, (void *)h->d_wsi[0], (void *)(h->d_wsi[1] == h->d_wsi[0] ? nullptr : h->d_wsi[1]),
                    (void *)h->d_halo_send[0], (void *)h->d_halo_send[1], (void *)h->d_halo_recv[0], (void *)h->d_halo_recv[1], (void *)h->d_halo_seg_send[0],
                    (void *)h->d_halo_seg_send[1], (void *)h->d_halo_seg_recv[0], (void *)h->d_halo_seg_recv[1]}) if (q) (void)hipFree(q);
    for (void *q : {(void *)h->d_ref, (void *)h->d_cnt_pm, (void *)h->d_block_table, (void *)h->d_scal, (void *)h->d_lo_hi, (void *)h->d_ref_ok, (void *)h->d_ws, (void *)h->d_occ_site, (void *)h->d_dcnt, (void *)h->d_dep, (void *)h->d_stepw}) if (q) (void)hipFree(q);
    void *ptrs[] = {h->d_src, h->d_orig, h->d_pcnt, h->d_plist, h->d_prop_own, h->d_anchor, h->d_sp8, h->d_tinfo,
                    h->d_stamps, h->d_plan, h->d_plan_n, h->d_accW, h->d_accS, h->d_occ, h->d_table, h->d_beta, h->d_exit, h->d_S, h->d_W, h->d_mfield, h->d_occ4, h->d_gsum,
                    h->d_nexit, h->d_tmp_sp8, h->d_tmp_tinfo};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int aps_set_stream(aps_handle *h, void *hip_stream) {
    if (!h) return APS_ERR_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) { HIP_TRY(h, hipStreamDestroy(h->stream)); h->own_stream = false; }
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);   // taken literally: NULL is the legacy default stream
    return APS_OK;
}

int aps_set_state(aps_handle *h, int32_t e, const int32_t *pos, const int8_t *sigma, const uint8_t *bound,
                  const uint8_t *alive, int64_t n) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !pos || !sigma || n < 0 || n > h->N) return fail(h, APS_ERR_ARG, "aps_set_state: bad ensemble, pointer or n");
    std::vector<int> occ((size_t)h->p.L, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (pos[i] < 0 || pos[i] >= h->p.L) return fail(h, APS_ERR_ARG, "aps_set_state: position outside [0, L)");
        if (sigma[i] != 1 && sigma[i] != -1) return fail(h, APS_ERR_ARG, "aps_set_state: sigma must be +1 or -1");
        if (!(alive && !alive[i]) && ++occ[(size_t)pos[i]] > h->p.K) return fail(h, APS_ERR_ARG, "aps_set_state: site capacity exceeded");
    }
    std::vector<uint32_t> src, orig; long long gsum[2];
    int rc = sync_slots(h);                                  // the other ensembles' particle-indexed arrays must be current
    if (rc) return rc;
    pack_ensemble(h, pos, sigma, bound, alive, n, src, orig, gsum);
    if ((rc = upload_ensemble(h, e, src, orig))) return rc;
    if (is_tiles(h) && (rc = upload_cells(h, e, pos, sigma, bound, alive, n, orig))) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_gsum + (size_t)(h->step & 1) * 2 * h->E + 2 * e, gsum, sizeof(gsum), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_nexit + e, 0, sizeof(unsigned), h->stream));
    if (h->d_occ_site) {
        static_assert(sizeof(int) == sizeof(uint32_t), "occupancy upload");
        HIP_TRY(h, hipMemcpyAsync(h->d_occ_site + (size_t)e * h->p.L, occ.data(), (size_t)h->p.L * 4, hipMemcpyHostToDevice, h->stream));
        h->field_dirty = true;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n_set[(size_t)e] = n;
    if (!h->ref_set.empty()) h->ref_set[(size_t)e] = 0;
    return APS_OK;
}

int aps_get_state(aps_handle *h, int32_t e, int32_t *pos, int8_t *sigma, uint8_t *bound, uint8_t *alive, int64_t n) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E) return fail(h, APS_ERR_ARG, "aps_get_state: bad ensemble");
    if (h->n_set[(size_t)e] < 0) return fail(h, APS_ERR_STATE, "aps_get_state: no state uploaded for this ensemble");
    if (n != h->n_set[(size_t)e]) return fail(h, APS_ERR_ARG, "aps_get_state: n differs from the uploaded particle count");
    { int rc_ = sync_slots(h); if (rc_) return rc_; }
    std::vector<uint32_t> src((size_t)h->Npad), orig((size_t)h->Npad);
    HIP_TRY(h, hipMemcpyAsync(src.data(), h->d_src + (size_t)e * h->Npad, src.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(orig.data(), h->d_orig + (size_t)e * h->Npad, orig.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int64_t s = 0; s < h->Npad; ++s) {
        const uint32_t i = orig[(size_t)s];
        if (i == 0xFFFFFFFFu) continue;
        if (i >= (uint64_t)n) return fail(h, APS_ERR_STATE, "aps_get_state: corrupt slot table");
        const uint32_t w = src[(size_t)s];
        if (pos) pos[i] = (int32_t)(w & POS_MASK);
        if (sigma) sigma[i] = (w & SPIN_BIT) ? 1 : -1;
        if (bound) bound[i] = (w & BOUND_BIT) ? 1 : 0;
        if (alive) alive[i] = (w & DEAD_BIT) ? 0 : ((w & AWAY_BIT) ? 2 : 1);
    }
    return APS_OK;
}

int aps_pair_accumulate(aps_handle *h, int32_t e, double *S, double *W, int32_t *occ4, int64_t n) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !S || !W || !occ4) return fail(h, APS_ERR_ARG, "aps_pair_accumulate: bad argument");
    if (!all_set(h)) return fail(h, APS_ERR_STATE, "aps_pair_accumulate: upload a state for every ensemble first");
    if (n != h->n_set[(size_t)e]) return fail(h, APS_ERR_ARG, "aps_pair_accumulate: n differs from the uploaded particle count");
    { int rc_ = sync_slots(h); if (rc_) return rc_; }
    const size_t EN = (size_t)h->E * (size_t)h->Npad;
    int rc;
    if (!h->d_S && ((rc = dev_alloc(h, &h->d_S, EN)) || (rc = dev_alloc(h, &h->d_W, EN)) || (rc = dev_alloc(h, &h->d_occ4, EN * 4)))) return rc;
    const PairArgs a = pair_args(h, true, false);
    if ((rc = launch_pair(h, a, 0, (int)h->ntiles))) return rc;            // the hook covers every tile, not only this rank's
    return download_hook(h, e, S, W, occ4);
}

}  // extern "C"

namespace {
// per-slot hook outputs of ensemble e -> caller's arrays in original particle order
int download_hook(aps_handle *h, int e, double *S, double *W, int32_t *occ4) {
    std::vector<double> s((size_t)h->Npad), w((size_t)h->Npad); std::vector<int> o((size_t)h->Npad * 4); std::vector<uint32_t> orig((size_t)h->Npad);
    HIP_TRY(h, hipMemcpyAsync(s.data(), h->d_S + (size_t)e * h->Npad, s.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(w.data(), h->d_W + (size_t)e * h->Npad, w.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(o.data(), h->d_occ4 + (size_t)e * h->Npad * 4, o.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(orig.data(), h->d_orig + (size_t)e * h->Npad, orig.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int64_t sl = 0; sl < h->Npad; ++sl) {
        const uint32_t i = orig[(size_t)sl];
        if (i == 0xFFFFFFFFu) continue;
        S[i] = s[(size_t)sl]; W[i] = w[(size_t)sl];
        for (int k = 0; k < 4; ++k) occ4[4 * (size_t)i + k] = o[4 * (size_t)sl + k];
    }
    return APS_OK;
}
}  // namespace

extern "C" {

int aps_propose(aps_handle *h) {
    if (!h) return APS_ERR_ARG;
    if (!all_set(h)) return fail(h, APS_ERR_STATE, "aps_propose: upload a state for every ensemble first");
    int rc = ensure_lattice(h);
    if (rc) return rc;
    return do_propose(h);
}

int aps_commit(aps_handle *h) {
    if (!h) return APS_ERR_ARG;
    if (!all_set(h)) return fail(h, APS_ERR_STATE, "aps_commit: upload a state for every ensemble first");
    return do_commit(h);
}

}  // extern "C"

namespace {

constexpr int NGRAPH = 6;
constexpr int GRAPH_SIZES[NGRAPH] = {32, 16, 8, 4, 2, 1};   // steps per captured graph

// Lattice / tile steps are a few microseconds of GPU time each, less than the host needs to launch their kernels
// one by one: runs of 32, 16, 8, 4, 2 and 1 steps are captured once -- for either parity of the first step, which the kernels'
// buffer arguments depend on; the step index itself lives in device memory (stepw) -- and any step count is replayed as
// a sum of those.
int capture_run(aps_handle *h, int par, int nsteps, hipGraphExec_t *out) {
    if (!h->cap_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    const bool dirty = h->slots_dirty, pending = h->field_pending, stale = h->ws_view_stale;
    const hipStream_t user_stream = h->stream;
    const int64_t step0 = h->step;
    HIP_TRY(h, hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    h->stream = h->cap_stream;
    h->step = par;                                           // only the parity is baked in
    int rc = APS_OK;
    for (int k = 0; k < nsteps && !rc; ++k) { rc = do_propose(h); if (!rc) rc = do_commit(h); }
    h->stream = user_stream;
    h->step = step0;
    h->slots_dirty = dirty; h->field_pending = pending; h->ws_view_stale = stale;      // capturing launched nothing
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(h->cap_stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ce != hipSuccess) return fail(h, APS_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    const hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) { *out = nullptr; return fail(h, APS_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
    return APS_OK;
}

int build_graphs(aps_handle *h) {
    if (h->graphs_built_f[h->flip]) return APS_OK;
    for (int par = 0; par < 2; ++par)
        for (int g = 0; g < NGRAPH; ++g) {
            int rc = capture_run(h, par, GRAPH_SIZES[g], &h->gexec_f[h->flip][par][g]);
            if (rc) return rc;
        }
    h->graphs_built_f[h->flip] = true;
    return APS_OK;
}

void drop_graphs(aps_handle *h) {
    for (int par = 0; par < 2; ++par)
        for (int f = 0; f < 2; ++f) for (int g = 0; g < NGRAPH; ++g) if (h->gexec_f[f][par][g]) { (void)hipGraphExecDestroy(h->gexec_f[f][par][g]); h->gexec_f[f][par][g] = nullptr; }
    for (int par = 0; par < 2; ++par)
        for (int f = 0; f < 2; ++f) for (int n = 0; n <= 64; ++n) if (h->gexact_f[f][par][n]) { (void)hipGraphExecDestroy(h->gexact_f[f][par][n]); h->gexact_f[f][par][n] = nullptr; }
    h->graphs_built_f[0] = h->graphs_built_f[1] = false;
}

int one_step(aps_handle *h) {
    int rc;
    if ((rc = do_propose(h))) return rc;
    if (h->ipc_on && is_tiles(h)) {                          // site-range shards: boundary state to the neighbours, by peer stores
        if (halo_due(h) && (rc = halo_exchange_ipc(h))) return rc;
    } else if (h->comm && is_tiles(h)) {                     // the same through ncclSend / ncclRecv
        if (halo_due(h) && (rc = halo_exchange_rccl(h))) return rc;
    } else if (h->comm) {                                    // one in-place all-gather of 1 byte per particle
        const size_t block = (size_t)h->E * (size_t)h->SH;
        const ncclResult_t nr = g_rccl.AllGather(h->d_prop + block * (size_t)h->rank, h->d_prop, block, ncclUint8, h->comm, h->stream);
        if (nr != ncclSuccess) return fail(h, APS_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(nr));
    }
    return do_commit(h);
}

int run_profiled(aps_handle *h, int64_t nsteps, double ms[KIND_N], int64_t counts[KIND_N], double *work) {
    for (int k = 0; k < KIND_N; ++k) { ms[k] = 0.0; counts[k] = 0; }
    if (work) *work = 0.0;
    int rc = ensure_lattice(h);
    if (rc) return rc;
    std::vector<uint32_t> pn((size_t)h->E * std::max<int64_t>(std::max<int64_t>(h->ntiles / RT, h->nb), h->ts_ntile));
    for (int64_t s = 0; s < nsteps; ++s) {
        h->profiling = true; h->prof_n = 0; h->prof_kind.clear();
        h->prof_dispatch = std::getenv("APS_PROF_BRACKET") == nullptr;
        rc = one_step(h);
        h->profiling = false;
        if (rc) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->prof_dispatch) {
            for (size_t i = 0; i < h->prof_kind.size(); ++i) {
                float t = 0.f;
                HIP_TRY(h, hipEventElapsedTime(&t, h->events[2 * i], h->events[2 * i + 1]));
                ms[h->prof_kind[i]] += t; counts[h->prof_kind[i]] += 1;
            }
        } else for (size_t i = 0; i + 1 < h->prof_n; ++i) {
            const int kind = h->prof_kind[i];
            if (kind == KIND_END) continue;
            float t = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&t, h->events[i], h->events[i + 1]));
            ms[kind] += t; counts[kind] += 1;
        }
        if (work && (s == 0 || s == nsteps - 1)) {           // the work per step changes slowly: sample first and last step
            const double wgt = nsteps == 1 ? 1.0 : 0.5 * (double)nsteps;
            double t = 0.0;
            if (is_tiles(h)) {                               // deposits of the step just taken (read by the next step)
                if (h->model.field_mode) {
                    HIP_TRY(h, hipMemcpy(pn.data(), h->d_tdcnt[h->step & 1], (size_t)h->E * h->ts_ntile * 4, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < (size_t)h->E * h->ts_ntile; ++i) t += (double)pn[i];
                }
            } else if (h->method == APS_METHOD_LATTICE) {    // deposits of the step just taken
                if (h->model.field_mode) {
                    HIP_TRY(h, hipMemcpy(pn.data(), h->d_dcnt, (size_t)h->E * h->nb * 4, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < (size_t)h->E * h->nb; ++i) t += (double)pn[i];
                }
            } else {                                         // (target tile, source tile) blocks = RT x sum of the list lengths
                HIP_TRY(h, hipMemcpy(pn.data(), h->d_plan_n, (size_t)h->E * (h->ntiles / RT) * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < (size_t)h->E * (h->ntiles / RT); ++i) t += (double)pn[i] * RT * TILE * TILE;
            }
            *work += t * wgt;
        }
    }
    return APS_OK;
}

}  // namespace

extern "C" {

int aps_step(aps_handle *h, int64_t nsteps) {
    if (!h) return APS_ERR_ARG;
    if (nsteps < 0) return fail(h, APS_ERR_ARG, "aps_step: nsteps < 0");
    if (h->world != 1 && !h->comm && !h->ipc_on)
        return fail(h, APS_ERR_STATE, "aps_step: sharded handle without transport; call aps_ipc_export / aps_ipc_connect or aps_comm_init, or use aps_propose / exchange / aps_commit");
    if (!all_set(h)) return fail(h, APS_ERR_STATE, "aps_step: upload a state for every ensemble first");
    int rc = ensure_lattice(h);
    if (rc) return rc;
    int64_t s = 0;
    static const bool no_graph = std::getenv("APS_NO_GRAPH") != nullptr;
    h->last_graph_steps = h->last_single_steps = h->last_loop_steps = 0;
    // Below that a call is not worth the loop's set-up: its launch costs ~22 us more than a graph replay (table and state
    // staged once, the synchronisation that reads the give-up word) and gains ~2 us per step on a full device (config 2:
    // break-even near 10 steps); a grid of at most one workgroup per CU gains more per step (reference-shaped runs with 7
    // steps between observations: 43 ms with the loop against 47 ms)
    int64_t loop_min = (int64_t)h->ts_ntile * h->E > h->num_cu ? 10 : 3;
    if (const char *env = std::getenv("APS_LOOP_MIN")) loop_min = std::max(1, std::atoi(env));
    if (is_tiles(h) && h->world == 1 && !h->comm && h->loop_wanted && nsteps >= loop_min && h->loop_state != 0 && h->loop_state != -1) {
        const char *env = std::getenv("APS_TILE_LOOP");
        if (!(env && env[0] == '0')) {
            if ((rc = loop_prepare(h))) return rc;
            if (h->loop_state == 1) {
                // the loop writes its final state into the buffer set of the OTHER parity whatever the step count, so its inputs
                // stay intact for the fallback; after an even number of steps that set holds the state of the SAME parity:
                // the two sets trade places (captured graphs are kept per `flip`)
                const int64_t n = std::min<int64_t>(nsteps - s, (int64_t)1 << 30);
                // particles that can leave: a call that is given up has already logged exits of steps that will be repeated --
                // the counts of before put the log back (the repeated steps write the same rows and the same dead marks again)
                const bool exits = h->model.k_exit > 0.0;
                std::vector<unsigned> n_exit_before;
                if (exits) {
                    n_exit_before.resize((size_t)h->E);
                    HIP_TRY(h, hipMemcpyAsync(n_exit_before.data(), h->d_nexit, (size_t)h->E * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
                }
                if ((rc = launch_tile_loop(h, n))) return rc;
                HIP_TRY(h, hipStreamSynchronize(h->stream));
#ifdef APS_LOOP_DEBUG
                if (const char *path = std::getenv("APS_LOOP_DUMP")) {
                    std::vector<uint32_t> buf((size_t)h->loop_dbg_n * 3 * h->p.L);
                    (void)hipMemcpy(buf.data(), h->loop_dbg, buf.size() * 4, hipMemcpyDeviceToHost);
                    if (FILE *f = std::fopen(path, "wb")) { std::fwrite(buf.data(), 4, buf.size(), f); std::fclose(f); }
                }
#endif
                if (h->h_abort[0]) {                             // the workgroups were not all resident: repeat the ordinary way, never again
                    h->h_abort[0] = 0u;
                    h->loop_state = -1;
                    h->loop_why = "a wait ran out (the grid was not resident at once); steps repeated with one launch per step";
                    // a tile that finished all n iterations before the call was given up has written the step word of the
                    // final parity -- for an even n that is the word the first repeated step reads: set the pair again
                    if ((rc = upload_stepw(h))) return rc;
                    if (exits) HIP_TRY(h, hipMemcpy(h->d_nexit, n_exit_before.data(), (size_t)h->E * sizeof(unsigned), hipMemcpyHostToDevice));
                } else {
                    h->step += n; s += n; h->last_loop_steps = n;
                    h->slots_dirty = true; h->field_pending = true; h->ws_view_stale = true;
                    if (!(n & 1)) {
                        std::swap(h->d_cell[0], h->d_cell[1]); std::swap(h->d_wsb[0], h->d_wsb[1]); std::swap(h->d_wsi[0], h->d_wsi[1]);
                        std::swap(h->d_tdcnt[0], h->d_tdcnt[1]); std::swap(h->d_tdep[0], h->d_tdep[1]); std::swap(h->d_gpart[0], h->d_gpart[1]);
                        h->d_ws = h->d_wsb[h->step & 1];
                        h->flip ^= 1;
                    }
                }
            }
        }
    }
    if ((h->method == APS_METHOD_LATTICE || is_tiles(h)) && h->world == 1 && !no_graph && nsteps > s) {
        if ((rc = build_graphs(h))) return rc;                             // once per handle, on the first stepping call
        if (s == 0 && nsteps <= 64 && nsteps != 32 && nsteps != 16 && nsteps != 8 && nsteps != 4 && nsteps != 2 && nsteps != 1) {
            // a short call that is not one of the stock sizes: one graph of exactly that many steps (captured on first use),
            // one launch instead of several
            hipGraphExec_t &ge = h->gexact_f[h->flip][h->step & 1][nsteps];
            if (!ge && (rc = capture_run(h, (int)(h->step & 1), (int)nsteps, &ge))) return rc;
            HIP_TRY(h, hipGraphLaunch(ge, h->stream));
            h->step += nsteps; h->last_graph_steps += nsteps; s = nsteps;
            if (is_tiles(h)) { h->slots_dirty = true; h->field_pending = !h->ntt_on; h->ws_view_stale = true; }
        }
        for (int g = 0; g < NGRAPH; ++g)
            for (; nsteps - s >= GRAPH_SIZES[g]; s += GRAPH_SIZES[g]) {
                HIP_TRY(h, hipGraphLaunch(h->gexec_f[h->flip][h->step & 1][g], h->stream));
                h->step += GRAPH_SIZES[g];
                h->last_graph_steps += GRAPH_SIZES[g];
                if (is_tiles(h)) { h->slots_dirty = true; h->field_pending = !h->ntt_on; h->ws_view_stale = true; }
            }
    }
    for (; s < nsteps; ++s, ++h->last_single_steps)
        if ((rc = one_step(h))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->h_ipc_err && h->h_ipc_err[0]) {
        h->h_ipc_err[0] = 0u;
        return fail(h, APS_ERR_STATE, "aps_step: a neighbour rank's halo did not arrive in time (peer-store transport); the state of this handle is incomplete");
    }
    return APS_OK;
}

int aps_set_resident_loop(aps_handle *h, int32_t on) {
    if (!h) return APS_ERR_ARG;
    h->loop_wanted = on ? 1 : 0;
    return APS_OK;
}

int aps_step_loop_timed(aps_handle *h, int64_t nsteps, double *kernel_ms, int64_t *loop_steps) {
    if (!h || !kernel_ms) return APS_ERR_ARG;
    h->loop_timed = true;
    const int rc = aps_step(h, nsteps);
    h->loop_timed = false;
    if (rc) return rc;
    *kernel_ms = 0.0;
    if (loop_steps) *loop_steps = h->last_loop_steps;
    if (h->last_loop_steps > 0) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->loop_ev[0], h->loop_ev[1]));
        *kernel_ms = ms;
    }
    return APS_OK;
}

int aps_loop_info(aps_handle *h, int64_t *loop_steps, int32_t *state, char *why, int32_t why_len) {
    if (!h) return APS_ERR_ARG;
    if (loop_steps) *loop_steps = h->last_loop_steps;
    if (state) *state = h->loop_state;
    if (why && why_len > 0) { std::strncpy(why, h->loop_why.c_str(), (size_t)why_len - 1); why[why_len - 1] = 0; }
    return APS_OK;
}

int aps_step_info(aps_handle *h, int64_t *graph_steps, int64_t *single_steps) {
    if (!h) return APS_ERR_ARG;
    if (graph_steps) *graph_steps = h->last_graph_steps;
    if (single_steps) *single_steps = h->last_single_steps;
    return APS_OK;
}

int aps_copy_bandwidth(aps_handle *h, int64_t nbytes, int32_t reps, double *gbytes_per_s) {
    if (!h || !gbytes_per_s || nbytes < (1 << 20) || reps < 1) return APS_ERR_ARG;
    const size_t n16 = (size_t)nbytes / 16;
    uint4 *a = nullptr, *b = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&a), n16 * 16));
    if (hipMalloc(reinterpret_cast<void **>(&b), n16 * 16) != hipSuccess) { (void)hipFree(a); return fail(h, APS_ERR_HIP, "aps_copy_bandwidth: out of device memory"); }
    (void)hipMemsetAsync(a, 1, n16 * 16, h->stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned blocks = (unsigned)((n16 + 1023) / 1024);
    hipLaunchKernelGGL(copy16, dim3(blocks), dim3(256), 0, h->stream, a, b, n16);        // warm-up
    (void)hipEventRecord(e0, h->stream);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy16, dim3(blocks), dim3(256), 0, h->stream, a, b, n16);
    (void)hipEventRecord(e1, h->stream);
    const hipError_t se = hipStreamSynchronize(h->stream);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
    if (se != hipSuccess || !(ms > 0.f)) return fail(h, APS_ERR_HIP, "aps_copy_bandwidth: copy kernel failed");
    *gbytes_per_s = 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9;                // bytes read + bytes written
    return APS_OK;
}

int aps_step_timed(aps_handle *h, int64_t nsteps, double *kernel_ms, int64_t *launches, double *work) {
    if (!h) return APS_ERR_ARG;
    if (nsteps < 0 || !kernel_ms) return fail(h, APS_ERR_ARG, "aps_step_timed: bad argument");
    if (h->world != 1) return fail(h, APS_ERR_STATE, "aps_step_timed: sharded handle");
    if (!all_set(h)) return fail(h, APS_ERR_STATE, "aps_step_timed: upload a state for every ensemble first");
    double ms[KIND_N]; int64_t cnt[KIND_N];
    int rc = run_profiled(h, nsteps, ms, cnt, work);
    if (rc) return rc;
    const int kind = is_tiles(h) ? KIND_TILE_STEP : (h->method == APS_METHOD_LATTICE ? KIND_FIELD_UPDATE : KIND_PAIR);
    *kernel_ms = ms[kind];
    if (launches) *launches = cnt[kind];
    return APS_OK;
}

int aps_step_profile(aps_handle *h, int64_t nsteps, double *ms8, int64_t *launches8) {
    if (!h) return APS_ERR_ARG;
    if (nsteps < 0 || !ms8) return fail(h, APS_ERR_ARG, "aps_step_profile: bad argument");
    if (h->world != 1) return fail(h, APS_ERR_STATE, "aps_step_profile: sharded handle");
    if (!all_set(h)) return fail(h, APS_ERR_STATE, "aps_step_profile: upload a state for every ensemble first");
    double ms[KIND_N]; int64_t cnt[KIND_N];
    int rc = run_profiled(h, nsteps, ms, cnt, nullptr);
    if (rc) return rc;
    for (int k = 0; k < 8; ++k) { ms8[k] = ms[k]; if (launches8) launches8[k] = cnt[k]; }
    h->prof_ntt_ms = ms[KIND_NTT]; h->prof_ntt_n = cnt[KIND_NTT];
    return APS_OK;
}

int aps_ntt_info(aps_handle *h, int32_t *on, int32_t *log2_m, double *prof_ms, int64_t *prof_launches) {
    if (!h) return APS_ERR_ARG;
    if (on) *on = h->ntt_on ? 1 : 0;
    if (log2_m) *log2_m = h->ntt_on ? h->ntt.m : 0;
    if (prof_ms) *prof_ms = h->prof_ntt_ms;
    if (prof_launches) *prof_launches = h->prof_ntt_n;
    return APS_OK;
}
int aps_ntt_launches(aps_handle *h) {
    if (!h || !h->ntt_on) return 0;
    return h->ntt_fused ? 3 : h->ntt.a2 > 0 ? 5 : 3;
}

int aps_mark_reference(aps_handle *h, int32_t e) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E) return fail(h, APS_ERR_ARG, "aps_mark_reference: bad ensemble");
    if (h->n_set[(size_t)e] < 0) return fail(h, APS_ERR_STATE, "aps_mark_reference: no state uploaded for this ensemble");
    { int rc_ = sync_slots(h); if (rc_) return rc_; }
    int rc;
    if (!h->d_ref && (rc = dev_alloc(h, &h->d_ref, (size_t)h->E * h->Npad))) return rc;
    if (h->ref_set.empty()) h->ref_set.assign((size_t)h->E, 0);
    HIP_TRY(h, hipMemcpyAsync(h->d_ref + (size_t)e * h->Npad, h->d_src + (size_t)e * h->Npad, (size_t)h->Npad * 4, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->ref_set[(size_t)e] = 1;
    return APS_OK;
}

}  // extern "C"

namespace {
// scalar sums of ensembles [e0, e0 + n): one counting launch, one summing launch, one download
int observe_scalars_impl(aps_handle *h, int e0, int n, int32_t x_wall, const int32_t *lo_hi, int32_t lo, int32_t hi,
                         const uint8_t *block_table, int64_t *out11) {
    const int K = h->p.K, L = h->p.L;
    int rc;
    if ((rc = sync_slots(h))) return rc;
    if (!h->d_cnt_pm && (rc = dev_alloc(h, &h->d_cnt_pm, (size_t)L * h->E))) return rc;
    if (!h->d_scal && ((rc = dev_alloc(h, &h->d_scal, (size_t)16 * h->E)) ||
                       (rc = dev_alloc(h, &h->d_block_table, (size_t)(K + 1) * (K + 1))) || (rc = dev_alloc(h, &h->d_lo_hi, (size_t)2 * h->E)) ||
                       (rc = dev_alloc(h, &h->d_ref_ok, (size_t)h->E)))) return rc;
    std::vector<uint8_t> table((size_t)(K + 1) * (K + 1), 0);
    if (block_table) table.assign(block_table, block_table + table.size());
    else for (int cp = 0; cp <= K; ++cp) for (int cm = 0; cm <= K; ++cm) table[(size_t)cp * (K + 1) + cm] = cp + cm >= 1;
    std::vector<long long> init((size_t)16 * n, 0);
    for (int k = 0; k < n; ++k) init[(size_t)16 * k + SC_MAXPOS] = -1;
    std::vector<uint8_t> ok((size_t)n, 0);
    for (int k = 0; k < n; ++k) ok[(size_t)k] = (!h->ref_set.empty() && h->ref_set[(size_t)(e0 + k)]) ? 1 : 0;
    HIP_TRY(h, hipMemcpyAsync(h->d_block_table, table.data(), table.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_scal, init.data(), init.size() * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_ref_ok, ok.data(), ok.size(), hipMemcpyHostToDevice, h->stream));
    if (lo_hi) HIP_TRY(h, hipMemcpyAsync(h->d_lo_hi, lo_hi, (size_t)2 * n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_cnt_pm, 0, (size_t)L * n * 4, h->stream));
    const uint32_t *src = h->d_src + (size_t)e0 * h->Npad;
    hipLaunchKernelGGL(count_sites, dim3((unsigned)(h->Npad / 256), (unsigned)n), dim3(256), 0, h->stream, src, h->d_cnt_pm, (int)h->Npad, L);
    ScalarArgs a{};
    a.src = src; a.block_table = h->d_block_table; a.cnt_pm = h->d_cnt_pm; a.out = h->d_scal;
    a.ref = h->d_ref ? h->d_ref + (size_t)e0 * h->Npad : nullptr; a.ref_ok = h->d_ref_ok;
    a.lo_hi = lo_hi ? h->d_lo_hi : nullptr;
    a.Npad = (int)h->Npad; a.L = L; a.K = K; a.x_wall = x_wall; a.lo = lo; a.hi = hi;
    hipLaunchKernelGGL(observe_scalars, dim3((unsigned)std::min<int64_t>(h->Npad / 256, 512), (unsigned)n), dim3(256), 0, h->stream, a);
    HIP_TRY(h, hipGetLastError());
    std::vector<long long> res((size_t)16 * n);
    HIP_TRY(h, hipMemcpyAsync(res.data(), h->d_scal, res.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < n; ++k) for (int q = 0; q < SC_COUNT; ++q) out11[(size_t)k * SC_COUNT + q] = res[(size_t)16 * k + q];
    return APS_OK;
}
}  // namespace

extern "C" {

int aps_observe_scalars(aps_handle *h, int32_t e, int32_t x_wall, int32_t range_lo, int32_t range_hi, const uint8_t *block_table,
                        int64_t *out11) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !out11) return fail(h, APS_ERR_ARG, "aps_observe_scalars: bad argument");
    if (h->n_set[(size_t)e] < 0) return fail(h, APS_ERR_STATE, "aps_observe_scalars: no state uploaded for this ensemble");
    return observe_scalars_impl(h, e, 1, x_wall, nullptr, range_lo, range_hi, block_table, out11);
}

int aps_observe_scalars_all(aps_handle *h, int32_t x_wall, const int32_t *range_lo_hi, const uint8_t *block_table, int64_t *out11) {
    if (!h) return APS_ERR_ARG;
    if (!out11) return fail(h, APS_ERR_ARG, "aps_observe_scalars_all: bad argument");
    if (!all_set(h)) return fail(h, APS_ERR_STATE, "aps_observe_scalars_all: upload a state for every ensemble first");
    return observe_scalars_impl(h, 0, h->E, x_wall, range_lo_hi, 0, -1, block_table, out11);
}

int aps_observe_structure(aps_handle *h, int32_t e, int32_t k_max, double *out) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !out || k_max < 1 || k_max > h->p.L) return fail(h, APS_ERR_ARG, "aps_observe_structure: bad argument");
    if (h->n_set[(size_t)e] < 0) return fail(h, APS_ERR_STATE, "aps_observe_structure: no state uploaded for this ensemble");
    int rc = sync_slots(h);
    if (rc) return rc;
    const int L = h->p.L;
    const size_t nout = 4 + 2 * (size_t)k_max;
    if (!h->d_cnt_pm && (rc = dev_alloc(h, &h->d_cnt_pm, (size_t)L * h->E))) return rc;
    double *d_out = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&d_out), nout * sizeof(double)));
    auto done = [&](int code) { (void)hipFree(d_out); return code; };
    if (hipMemsetAsync(d_out, 0, nout * sizeof(double), h->stream) != hipSuccess ||
        hipMemsetAsync(h->d_cnt_pm, 0, (size_t)L * 4, h->stream) != hipSuccess) return done(fail(h, APS_ERR_HIP, "aps_observe_structure: memset failed"));
    const uint32_t *src = h->d_src + (size_t)e * h->Npad;
    hipLaunchKernelGGL(count_sites, dim3((unsigned)(h->Npad / 256), 1u), dim3(256), 0, h->stream, src, h->d_cnt_pm, (int)h->Npad, L);
    if ((rc = launch_field(h, e, h->d_sp8 + (size_t)e * h->Npad, h->d_tinfo + (size_t)e * h->ntiles, (int)h->ntiles, h->d_mfield))) return done(rc);
    hipLaunchKernelGGL(structure_sites, dim3((unsigned)std::min(1024, (L + 255) / 256)), dim3(256), 0, h->stream, h->d_cnt_pm, h->d_mfield, L, d_out);
    hipLaunchKernelGGL(structure_dft, dim3((unsigned)k_max), dim3(256), 0, h->stream, src, (int)h->Npad, L, d_out);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, d_out, nout * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return done(fail(h, APS_ERR_HIP, "aps_observe_structure: kernel or copy failed"));
    return done(APS_OK);
}

int aps_observe_bins(aps_handle *h, int32_t e, int32_t nbins, int64_t *plus, int64_t *minus) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !plus || !minus || nbins < 1 || nbins > h->p.L) return fail(h, APS_ERR_ARG, "aps_observe_bins: bad argument");
    if (h->n_set[(size_t)e] < 0) return fail(h, APS_ERR_STATE, "aps_observe_bins: no state uploaded for this ensemble");
    int rc = sync_slots(h);
    if (rc) return rc;
    const int bin_sites = (h->p.L + nbins - 1) / nbins;
    unsigned long long *d = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&d), (size_t)2 * nbins * 8));
    auto done = [&](int code) { (void)hipFree(d); return code; };
    if (hipMemsetAsync(d, 0, (size_t)2 * nbins * 8, h->stream) != hipSuccess) return done(fail(h, APS_ERR_HIP, "aps_observe_bins: memset failed"));
    hipLaunchKernelGGL(bin_counts, dim3((unsigned)(h->Npad / 256)), dim3(256), 0, h->stream, h->d_src + (size_t)e * h->Npad, (int)h->Npad, bin_sites, d, d + nbins,
                       is_tiles(h) && h->world > 1 ? h->own_lo : 0, is_tiles(h) && h->world > 1 ? h->own_hi : h->p.L);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(plus, d, (size_t)nbins * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipMemcpyAsync(minus, d + nbins, (size_t)nbins * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return done(fail(h, APS_ERR_HIP, "aps_observe_bins: kernel or copy failed"));
    return done(APS_OK);
}

int aps_method(aps_handle *h) { return h ? h->method : APS_ERR_ARG; }

int aps_event_overhead(aps_handle *h, int32_t reps, double *ms_per_pair) {
    if (!h || !ms_per_pair || reps < 1) return APS_ERR_ARG;
    while ((int64_t)h->events.size() < 2) { hipEvent_t ev; HIP_TRY(h, hipEventCreate(&ev)); h->events.push_back(ev); }
    double total = 0.0;
    for (int r = 0; r < reps; ++r) {                        // two events recorded back to back on an idle stream
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipEventRecord(h->events[0], h->stream));
        HIP_TRY(h, hipEventRecord(h->events[1], h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        float t = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&t, h->events[0], h->events[1]));
        total += t;
    }
    *ms_per_pair = total / reps;
    return APS_OK;
}

int aps_lattice_accumulate(aps_handle *h, int32_t e, double *S, double *W, int32_t *occ4, int64_t n) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !S || !W || !occ4) return fail(h, APS_ERR_ARG, "aps_lattice_accumulate: bad argument");
    if (h->method != APS_METHOD_LATTICE && !is_tiles(h)) return fail(h, APS_ERR_STATE, "aps_lattice_accumulate: handle uses the all-pairs formulation");
    if (!all_set(h)) return fail(h, APS_ERR_STATE, "aps_lattice_accumulate: upload a state for every ensemble first");
    if (n != h->n_set[(size_t)e]) return fail(h, APS_ERR_ARG, "aps_lattice_accumulate: n differs from the uploaded particle count");
    const size_t EN = (size_t)h->E * (size_t)h->Npad;
    int rc;
    if (!h->d_S && ((rc = dev_alloc(h, &h->d_S, EN)) || (rc = dev_alloc(h, &h->d_W, EN)) || (rc = dev_alloc(h, &h->d_occ4, EN * 4)))) return rc;
    if ((rc = ensure_lattice(h)) || (rc = sync_slots(h))) return rc;
    const LatticeArgs a = lattice_args(h, true, false);
    if ((rc = launch_lattice_propose(h, a, 0, (int)h->ntiles))) return rc;
    return download_hook(h, e, S, W, occ4);
}

int aps_get_lattice(aps_handle *h, int32_t e, double *W, double *S, int32_t *occ) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E) return fail(h, APS_ERR_ARG, "aps_get_lattice: bad ensemble");
    if (h->method != APS_METHOD_LATTICE && !is_tiles(h)) return fail(h, APS_ERR_STATE, "aps_get_lattice: handle uses the all-pairs formulation");
    if (h->n_set[(size_t)e] < 0) return fail(h, APS_ERR_STATE, "aps_get_lattice: no state uploaded for this ensemble");
    int rc = ensure_lattice(h);
    if (rc || (rc = sync_slots(h))) return rc;
    const size_t L = (size_t)h->p.L;
    if (W || S) {
        std::vector<double2> ws(L, make_double2(0.0, 0.0));
        if (h->model.field_mode) HIP_TRY(h, hipMemcpy(ws.data(), h->d_ws + (size_t)e * L, L * sizeof(double2), hipMemcpyDeviceToHost));
        for (size_t x = 0; x < L; ++x) { if (W) W[x] = ws[x].x; if (S) S[x] = ws[x].y; }
    }
    if (occ) HIP_TRY(h, hipMemcpy(occ, h->d_occ_site + (size_t)e * L, L * 4, hipMemcpyDeviceToHost));
    return APS_OK;
}

int aps_rates_from_field(aps_handle *h, int32_t e, const int32_t *pos, const int8_t *sigma, const uint8_t *bound, int64_t n,
                         const double *m_field, const int64_t *counts_p, const int64_t *counts_m, double *out9n) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !pos || !sigma || !bound || !m_field || !counts_p || !counts_m || !out9n || n < 0)
        return fail(h, APS_ERR_ARG, "aps_rates_from_field: bad argument");
    if (n == 0) return APS_OK;
    const int L = h->p.L;
    for (int64_t i = 0; i < n; ++i)
        if (pos[i] < 0 || pos[i] >= L) return fail(h, APS_ERR_ARG, "aps_rates_from_field: position outside [0, L)");
    std::vector<int32_t> occ((size_t)L);
    for (int x = 0; x < L; ++x) occ[(size_t)x] = (int32_t)(counts_p[x] + counts_m[x]);
    void *d_pos = nullptr, *d_sig = nullptr, *d_bnd = nullptr, *d_m = nullptr, *d_occ = nullptr, *d_out = nullptr;
    auto cleanup = [&]() { for (void *q : {d_pos, d_sig, d_bnd, d_m, d_occ, d_out}) if (q) (void)hipFree(q); };
#define TRY_(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); h->err = std::string(#expr) + ": " + hipGetErrorString(e_); return APS_ERR_HIP; } } while (0)
    TRY_(hipMalloc(&d_pos, (size_t)n * 4)); TRY_(hipMalloc(&d_sig, (size_t)n)); TRY_(hipMalloc(&d_bnd, (size_t)n));
    TRY_(hipMalloc(&d_m, (size_t)L * 8)); TRY_(hipMalloc(&d_occ, (size_t)L * 4)); TRY_(hipMalloc(&d_out, (size_t)n * 9 * 8));
    TRY_(hipMemcpyAsync(d_pos, pos, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    TRY_(hipMemcpyAsync(d_sig, sigma, (size_t)n, hipMemcpyHostToDevice, h->stream));
    TRY_(hipMemcpyAsync(d_bnd, bound, (size_t)n, hipMemcpyHostToDevice, h->stream));
    TRY_(hipMemcpyAsync(d_m, m_field, (size_t)L * 8, hipMemcpyHostToDevice, h->stream));
    TRY_(hipMemcpyAsync(d_occ, occ.data(), (size_t)L * 4, hipMemcpyHostToDevice, h->stream));
    RatesArgs a{};
    a.m = h->model; a.beta = h->beta[(size_t)e]; a.pos = (const int32_t *)d_pos; a.sigma = (const int8_t *)d_sig;
    a.bound = (const uint8_t *)d_bnd; a.m_field = (const double *)d_m; a.occ = (const int32_t *)d_occ; a.anchor = h->d_anchor;
    a.out = (double *)d_out; a.n = n;
    hipLaunchKernelGGL(rates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a);
    TRY_(hipGetLastError());
    TRY_(hipMemcpyAsync(out9n, d_out, (size_t)n * 9 * 8, hipMemcpyDeviceToHost, h->stream));
    TRY_(hipStreamSynchronize(h->stream));
#undef TRY_
    cleanup();
    return APS_OK;
}

int aps_comm_unique_id(uint8_t *out128) {
    if (!out128) return APS_ERR_ARG;
    if (!g_rccl.load()) { g_create_error = g_rccl.err; return APS_ERR_HIP; }
    ncclUniqueId id;
    const ncclResult_t nr = g_rccl.GetUniqueId(&id);
    if (nr != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(nr); return APS_ERR_HIP; }
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, &id, 128);
    return APS_OK;
}

int aps_comm_init(aps_handle *h, const uint8_t *id128) {
    if (!h || !id128) return APS_ERR_ARG;
    if (h->comm) return fail(h, APS_ERR_STATE, "aps_comm_init: communicator already initialised");
    if (!g_rccl.load()) return fail(h, APS_ERR_HIP, g_rccl.err);
    HIP_TRY(h, hipSetDevice(h->p.device));
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    const ncclResult_t nr = g_rccl.CommInitRank(&h->comm, h->world, id, h->rank);
    if (nr != ncclSuccess) { h->comm = nullptr; return fail(h, APS_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(nr)); }
    return APS_OK;
}

int aps_comm_ranks(aps_handle *h, int32_t *nranks) {
    if (!h || !nranks) return APS_ERR_ARG;
    if (!h->comm) return fail(h, APS_ERR_STATE, "aps_comm_ranks: no communicator (aps_comm_init first)");
    int n = 0;
    const ncclResult_t nr = g_rccl.CommCount(h->comm, &n);
    if (nr != ncclSuccess) return fail(h, APS_ERR_HIP, std::string("ncclCommCount: ") + g_rccl.GetErrorString(nr));
    *nranks = n;
    return APS_OK;
}

int aps_comm_selftest(aps_handle *h, int64_t nbytes) {
    if (!h || nbytes < 1 || nbytes > (1 << 26)) return APS_ERR_ARG;
    if (!h->comm) return fail(h, APS_ERR_STATE, "aps_comm_selftest: no communicator (aps_comm_init first)");
    // the halo's transport calls (ncclGroupStart, ncclSend, ncclRecv, ncclGroupEnd on the handle's stream), rank -> itself
    uint8_t *snd = nullptr, *rcv = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&snd), (size_t)nbytes));
    if (hipMalloc(reinterpret_cast<void **>(&rcv), (size_t)nbytes) != hipSuccess) { (void)hipFree(snd); return fail(h, APS_ERR_HIP, "aps_comm_selftest: out of device memory"); }
    std::vector<uint8_t> pat((size_t)nbytes), got((size_t)nbytes, 0);
    for (size_t i = 0; i < pat.size(); ++i) pat[i] = (uint8_t)(i * 131u + 7u + (unsigned)h->rank);
    hipError_t he = hipMemcpyAsync(snd, pat.data(), pat.size(), hipMemcpyHostToDevice, h->stream);
    if (he == hipSuccess) he = hipMemsetAsync(rcv, 0, (size_t)nbytes, h->stream);
    ncclResult_t nr = g_rccl.GroupStart();
    if (nr == ncclSuccess) nr = g_rccl.Send(snd, (size_t)nbytes, ncclUint8, h->rank, h->comm, h->stream);
    if (nr == ncclSuccess) nr = g_rccl.Recv(rcv, (size_t)nbytes, ncclUint8, h->rank, h->comm, h->stream);
    const ncclResult_t ge = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ge;
    if (he == hipSuccess) he = hipMemcpyAsync(got.data(), rcv, got.size(), hipMemcpyDeviceToHost, h->stream);
    const hipError_t se = hipStreamSynchronize(h->stream);
    (void)hipFree(snd); (void)hipFree(rcv);
    if (nr != ncclSuccess) return fail(h, APS_ERR_HIP, std::string("aps_comm_selftest (ncclSend/ncclRecv): ") + g_rccl.GetErrorString(nr));
    if (he != hipSuccess || se != hipSuccess) return fail(h, APS_ERR_HIP, "aps_comm_selftest: HIP error");
    if (got != pat) return fail(h, APS_ERR_HIP, "aps_comm_selftest: received bytes differ from the bytes sent");
    return APS_OK;
}

// ---- peer-store transport of the halo: export this rank's landing buffers, connect to the neighbours'
namespace {
struct IpcBlob {                              // what aps_ipc_export hands out (APS_IPC_BLOB_BYTES = 256)
    uint32_t magic; int32_t pid, device, rank;
    uint64_t base, total, recv_bytes[2], land_off[2][2];   // land_off[parity][from_side]
    hipIpcMemHandle_t handle;
};
static_assert(sizeof(IpcBlob) <= 256, "IpcBlob must fit the blob");
constexpr uint32_t IPC_MAGIC = 0x41505331u;   // "APS1"
}

int aps_ipc_export(aps_handle *h, uint8_t *blob256) {
    if (!h || !blob256) return APS_ERR_ARG;
    if (!is_tiles(h) || h->world < 2) return fail(h, APS_ERR_STATE, "aps_ipc_export: not a site-sharded tiles handle");
    HIP_TRY(h, hipSetDevice(h->p.device));
    if (!h->ipc_land) {
        size_t off = 256;                                        // arrival words first: [parity][from_side], 64 bytes apart
        for (int par = 0; par < 2; ++par)
            for (int side = 0; side < 2; ++side) { h->ipc_land_off[par][side] = off; off += (h->halo_bytes_recv[side] + 255) / 256 * 256; }
        h->ipc_land_bytes = off;
        // fine-grained device memory: written by another device while this one polls it, never cached on the way
        void *ptr = nullptr;
        hipError_t e = hipExtMallocWithFlags(&ptr, off, hipDeviceMallocFinegrained);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&ptr, off); }
        if (e != hipSuccess) return fail(h, APS_ERR_HIP, std::string("aps_ipc_export: landing buffer: ") + hipGetErrorString(e));
        h->ipc_land = static_cast<char *>(ptr);
        HIP_TRY(h, hipMemset(h->ipc_land, 0, off));
        int rc;
        if ((rc = dev_alloc(h, &h->d_ipc_done, 2))) return rc;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_ipc_err), 64, hipHostMallocMapped));
        h->h_ipc_err[0] = 0u;
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void **>(&h->h_ipc_err_dev), h->h_ipc_err, 0));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    IpcBlob b{};
    b.magic = IPC_MAGIC; b.pid = (int32_t)getpid(); b.device = h->p.device; b.rank = h->rank;
    b.base = (uint64_t)(uintptr_t)h->ipc_land; b.total = h->ipc_land_bytes;
    for (int side = 0; side < 2; ++side) b.recv_bytes[side] = h->halo_bytes_recv[side];
    for (int par = 0; par < 2; ++par) for (int side = 0; side < 2; ++side) b.land_off[par][side] = h->ipc_land_off[par][side];
    const hipError_t e = hipIpcGetMemHandle(&b.handle, h->ipc_land);
    if (e != hipSuccess) return fail(h, APS_ERR_HIP, std::string("aps_ipc_export: hipIpcGetMemHandle: ") + hipGetErrorString(e));
    std::memset(blob256, 0, 256);
    std::memcpy(blob256, &b, sizeof(b));
    return APS_OK;
}

int aps_ipc_connect(aps_handle *h, const uint8_t *left_blob, const uint8_t *right_blob) {
    if (!h) return APS_ERR_ARG;
    if (!is_tiles(h) || h->world < 2) return fail(h, APS_ERR_STATE, "aps_ipc_connect: not a site-sharded tiles handle");
    if (!h->ipc_land) return fail(h, APS_ERR_STATE, "aps_ipc_connect: call aps_ipc_export first");
    if (h->ipc_on) return fail(h, APS_ERR_STATE, "aps_ipc_connect: already connected");
    int left, right;
    halo_peers(h, left, right);
    const int want[2] = {left, right};
    const uint8_t *blobs[2] = {left_blob, right_blob};
    HIP_TRY(h, hipSetDevice(h->p.device));
    IpcBlob b[2];
    for (int side = 0; side < 2; ++side) {
        if (want[side] < 0) continue;
        if (!blobs[side]) return fail(h, APS_ERR_ARG, "aps_ipc_connect: a neighbour's blob is missing");
        std::memcpy(&b[side], blobs[side], sizeof(IpcBlob));
        if (b[side].magic != IPC_MAGIC || b[side].rank != want[side]) return fail(h, APS_ERR_ARG, "aps_ipc_connect: not the blob of that neighbour rank");
        // this rank's first block (side 0) lands where the LEFT neighbour keeps "the right neighbour's first block" (its recv[0]);
        // the last block (side 1) where the RIGHT neighbour keeps "the left neighbour's last block" (its recv[1])
        if (b[side].recv_bytes[side] != h->halo_bytes_send[side]) return fail(h, APS_ERR_ARG, "aps_ipc_connect: the neighbour expects a block of another size (different lattice, tiling or halo interval)");
    }
    for (int side = 0; side < 2; ++side) {
        if (want[side] < 0) continue;
        if (side == 1 && want[0] == want[1] && h->ipc_peer[0]) { h->ipc_peer[1] = h->ipc_peer[0]; }          // two ranks on a torus: one neighbour
        else if (b[side].pid == (int32_t)getpid()) h->ipc_peer[side] = reinterpret_cast<char *>((uintptr_t)b[side].base);   // another handle of this process
        else {
            int can = 0;
            if (b[side].device != h->p.device && hipDeviceCanAccessPeer(&can, h->p.device, b[side].device) == hipSuccess && can)
                if (hipDeviceEnablePeerAccess(b[side].device, 0) != hipSuccess) (void)hipGetLastError();      // (already enabled is fine)
            void *ptr = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&ptr, b[side].handle, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { (void)hipGetLastError(); return fail(h, APS_ERR_HIP, std::string("aps_ipc_connect: hipIpcOpenMemHandle: ") + hipGetErrorString(e)); }
            h->ipc_opened[side] = ptr;
            h->ipc_peer[side] = static_cast<char *>(ptr);
        }
        for (int par = 0; par < 2; ++par) h->ipc_peer_off[side][par] = (size_t)b[side].land_off[par][side];
    }
    h->ipc_on = true;
    return APS_OK;
}

int aps_set_flip_table(aps_handle *h, const double *table, int32_t n) {
    if (!h) return APS_ERR_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->graphs_built_f[0] || h->graphs_built_f[1]) drop_graphs(h);      // captured kernels hold the rate parameters by value
    if (h->d_flip_tab) { (void)hipFree(h->d_flip_tab); h->d_flip_tab = nullptr; }
    h->model.flip_tab = nullptr; h->model.flip_n = 0;
    if (table) {
        if (n < 1 || n > (1 << 24)) return fail(h, APS_ERR_ARG, "aps_set_flip_table: n must be in [1, 2^24]");
        for (int64_t i = 0; i < 2 * ((int64_t)n + 1); ++i)
            if (!(table[i] >= 0.0) || !std::isfinite(table[i])) return fail(h, APS_ERR_ARG, "aps_set_flip_table: rates must be finite and >= 0");
        HIP_TRY(h, hipMalloc(reinterpret_cast<void **>(&h->d_flip_tab), (size_t)2 * ((size_t)n + 1) * sizeof(double)));
        HIP_TRY(h, hipMemcpy(h->d_flip_tab, table, (size_t)2 * ((size_t)n + 1) * sizeof(double), hipMemcpyHostToDevice));
        h->model.flip_tab = h->d_flip_tab; h->model.flip_n = n;
    }
    if (h->d_model) HIP_TRY(h, hipMemcpy(h->d_model, &h->model, sizeof(Model), hipMemcpyHostToDevice));
    return APS_OK;
}

int aps_tiles_info(aps_handle *h, int32_t *frame_sites, int32_t *owned_sites, int32_t *n_tiles, int32_t *table_in_lds) {
    if (!h) return APS_ERR_ARG;
    if (!is_tiles(h)) return fail(h, APS_ERR_STATE, "aps_tiles_info: not a tiles handle");
    if (frame_sites) *frame_sites = 64 * h->ts_RS;
    if (owned_sites) *owned_sites = h->ts_own;
    if (n_tiles) *n_tiles = h->ts_ntile;
    if (table_in_lds) *table_in_lds = h->ts_table_in_lds ? 1 : 0;
    return APS_OK;
}

int aps_exchange_kind(aps_handle *h) { return !h ? APS_ERR_ARG : (h->ipc_on ? 2 : (h->comm ? 1 : 0)); }

int aps_owned_sites(aps_handle *h, int32_t *lo, int32_t *hi) {
    if (!h) return APS_ERR_ARG;
    const bool sharded_sites = is_tiles(h) && h->world > 1;
    if (lo) *lo = sharded_sites ? h->own_lo : 0;
    if (hi) *hi = sharded_sites ? h->own_hi : h->p.L;
    return APS_OK;
}

int aps_halo_copy(aps_handle *dst, aps_handle *src) {
    if (!dst || !src) return APS_ERR_ARG;
    aps_handle *h = dst;
    if (!is_tiles(dst) || !is_tiles(src) || dst->world != src->world || dst->world < 2 || dst->p.L != src->p.L || dst->E != src->E ||
        dst->ts_own != src->ts_own || dst->ts_dcap != src->ts_dcap || dst->p.K != src->p.K || dst->p.device != src->p.device)
        return fail(h, APS_ERR_ARG, "aps_halo_copy: both handles must be site-sharded tiles handles of the same shape on one device");
    int left, right;
    halo_peers(dst, left, right);
    if (src->rank != left && src->rank != right) return fail(h, APS_ERR_ARG, "aps_halo_copy: src is not a neighbour rank of dst");
    // both handles have launched this step's kernel and not yet committed: the fresh buffers are [(step & 1) ^ 1] on either side
    if (dst->step != src->step) return fail(h, APS_ERR_STATE, "aps_halo_copy: the two handles are at different steps");
    if (dst->ts_kx != src->ts_kx || dst->halo_age != src->halo_age) return fail(h, APS_ERR_STATE, "aps_halo_copy: the two handles differ in halo interval or age");
    if (!halo_due(dst)) return fail(h, APS_ERR_STATE, "aps_halo_copy: no halo exchange is due at this step (aps_halo_info)");
    int rc;
    for (int side = 0; side < 2; ++side) {                       // side 0: src's first block (src is dst's right neighbour), 1: its last
        if (src->rank != (side == 0 ? right : left)) continue;
        if (src->halo_bytes_send[side] != dst->halo_bytes_recv[side]) return fail(h, APS_ERR_STATE, "aps_halo_copy: block sizes differ");
        if ((rc = halo_launch(src, side, false))) { h->err = src->err; return rc; }
        HIP_TRY(h, hipStreamSynchronize(src->stream));
        HIP_TRY(h, hipMemcpyAsync(dst->d_halo_recv[side], src->d_halo_send[side], src->halo_bytes_send[side], hipMemcpyDeviceToDevice, dst->stream));
        if ((rc = halo_launch(dst, side, true))) return rc;
        dst->halo_got |= 1u << side;
    }
    HIP_TRY(h, hipStreamSynchronize(dst->stream));
    return APS_OK;
}

// transport-agnostic halo: this rank's first (side 0) or last (side 1) block as one packed message in a host buffer, and
// the reverse for a message received from a neighbour (from_side 0: the RIGHT neighbour's first block, 1: the LEFT one's last)
int aps_halo_pack(aps_handle *h, int32_t side, uint8_t *host, int64_t cap, int64_t *nbytes) {
    if (!h || !nbytes || side < 0 || side > 1) return APS_ERR_ARG;
    if (!is_tiles(h) || h->world < 2) return fail(h, APS_ERR_STATE, "aps_halo_pack: not a site-sharded tiles handle");
    *nbytes = (int64_t)h->halo_bytes_send[side];
    if (!host) return APS_OK;
    if (!halo_due(h)) return fail(h, APS_ERR_STATE, "aps_halo_pack: no halo exchange is due at this step (aps_halo_info)");
    if (cap < *nbytes) return fail(h, APS_ERR_ARG, "aps_halo_pack: buffer too small");
    int rc = halo_launch(h, side, false);
    if (rc) return rc;
    if (*nbytes) HIP_TRY(h, hipMemcpyAsync(host, h->d_halo_send[side], (size_t)*nbytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return APS_OK;
}

int aps_halo_unpack(aps_handle *h, int32_t from_side, const uint8_t *host, int64_t nbytes) {
    if (!h || !host || from_side < 0 || from_side > 1) return APS_ERR_ARG;
    if (!is_tiles(h) || h->world < 2) return fail(h, APS_ERR_STATE, "aps_halo_unpack: not a site-sharded tiles handle");
    if (!h->halo_nseg_recv[from_side]) return fail(h, APS_ERR_ARG, "aps_halo_unpack: no neighbour on that side (reflecting wall)");
    if ((int64_t)h->halo_bytes_recv[from_side] != nbytes) return fail(h, APS_ERR_ARG, "aps_halo_unpack: byte count does not match the neighbour's block");
    if (!halo_due(h)) return fail(h, APS_ERR_STATE, "aps_halo_unpack: no halo exchange is due at this step (aps_halo_info)");
    HIP_TRY(h, hipMemcpyAsync(h->d_halo_recv[from_side], host, (size_t)nbytes, hipMemcpyHostToDevice, h->stream));
    int rc = halo_launch(h, from_side, true);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->halo_got |= 1u << from_side;
    return APS_OK;
}

int aps_halo_sizes(aps_handle *h, int64_t send_bytes[2], int64_t recv_bytes[2]) {
    if (!h) return APS_ERR_ARG;
    if (!is_tiles(h) || h->world < 2) return fail(h, APS_ERR_STATE, "aps_halo_sizes: not a site-sharded tiles handle");
    int left, right;
    halo_peers(h, left, right);
    if (send_bytes) { send_bytes[0] = left >= 0 ? (int64_t)h->halo_bytes_send[0] : 0; send_bytes[1] = right >= 0 ? (int64_t)h->halo_bytes_send[1] : 0; }
    if (recv_bytes) { recv_bytes[0] = right >= 0 ? (int64_t)h->halo_bytes_recv[0] : 0; recv_bytes[1] = left >= 0 ? (int64_t)h->halo_bytes_recv[1] : 0; }
    return APS_OK;
}

int aps_halo_info(aps_handle *h, int32_t *interval, int32_t *age, int32_t *due) {
    if (!h) return APS_ERR_ARG;
    const bool sharded_sites = is_tiles(h) && h->world > 1;
    if (interval) *interval = sharded_sites ? h->ts_kx : 0;
    if (age) *age = sharded_sites ? h->halo_age : 0;
    if (due) *due = sharded_sites && halo_due(h) ? 1 : 0;
    return APS_OK;
}

int aps_exchange_buffer(aps_handle *h, void **dev_ptr, int64_t *total_bytes, int64_t *my_offset, int64_t *my_bytes) {
    if (!h) return APS_ERR_ARG;
    const int64_t block = (int64_t)h->E * h->SH;
    if (dev_ptr) *dev_ptr = h->d_prop;
    if (total_bytes) *total_bytes = block * h->world;
    if (my_offset) *my_offset = block * h->rank;
    if (my_bytes) *my_bytes = block;
    return APS_OK;
}

int aps_bind_exchange_buffer(aps_handle *h, void *dev_ptr, int64_t nbytes) {
    if (!h) return APS_ERR_ARG;
    if (h->graphs_built_f[0] || h->graphs_built_f[1]) { (void)hipStreamSynchronize(h->stream); drop_graphs(h); }   // captured kernels hold the old pointer
    if (!dev_ptr) { h->d_prop = h->d_prop_own; return APS_OK; }
    if (nbytes < (int64_t)h->E * h->SH * h->world) return fail(h, APS_ERR_ARG, "aps_bind_exchange_buffer: buffer too small");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->d_prop = static_cast<uint8_t *>(dev_ptr);
    return APS_OK;
}

// diagnostic builds only (-DAPS_STAMPS): per-workgroup phase cycle totals of the last pair_propose launch
int aps_debug_stamps(aps_handle *h, unsigned long long *out, int64_t nwords) {
    if (!h || !out) return APS_ERR_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->d_stamps, (size_t)std::min<int64_t>(nwords, 8 * 4096) * 8, hipMemcpyDeviceToHost));
    return APS_OK;
}

// diagnostic: per-target-tile planned list lengths of ensemble 0
int aps_debug_plan_n(aps_handle *h, uint32_t *out, int64_t n) {
    if (!h || !out) return APS_ERR_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->d_plan_n, (size_t)std::min<int64_t>(n, h->ntiles / RT) * 4, hipMemcpyDeviceToHost));
    return APS_OK;
}

int aps_time(aps_handle *h, double *t, int64_t *step_index) {
    if (!h) return APS_ERR_ARG;
    if (t) *t = (double)h->step * h->p.dt;
    if (step_index) *step_index = h->step;
    return APS_OK;
}

int aps_get_table(aps_handle *h, double *out, int32_t cap, int32_t *tlen, int32_t *q) {
    if (!h) return APS_ERR_ARG;
    if (tlen) *tlen = h->tlen;
    if (q) *q = h->q;
    if (out) {
        if (cap < h->tlen) return fail(h, APS_ERR_ARG, "aps_get_table: buffer too small");
        std::memcpy(out, h->table.data(), (size_t)h->tlen * sizeof(double));
    }
    return APS_OK;
}

int aps_get_exits(aps_handle *h, int32_t e, double *rows3, int64_t cap_rows, int64_t *n_rows) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !n_rows) return fail(h, APS_ERR_ARG, "aps_get_exits: bad argument");
    unsigned n = 0;
    HIP_TRY(h, hipMemcpyAsync(&n, h->d_nexit + e, sizeof(n), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    n = std::min<unsigned>(n, (unsigned)h->exit_cap);
    *n_rows = n;
    if (!rows3 || !n) return APS_OK;
    if (cap_rows < (int64_t)n) return fail(h, APS_ERR_ARG, "aps_get_exits: buffer too small");
    std::vector<double> raw((size_t)n * 3);
    HIP_TRY(h, hipMemcpy(raw.data(), h->d_exit + (size_t)e * h->exit_cap * 3, raw.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned> idx(n);
    std::iota(idx.begin(), idx.end(), 0u);
    std::sort(idx.begin(), idx.end(), [&](unsigned a, unsigned b) {
        if (raw[3 * a] != raw[3 * b]) return raw[3 * a] < raw[3 * b];
        return raw[3 * a + 2] < raw[3 * b + 2];
    });
    for (unsigned k = 0; k < n; ++k) {
        rows3[3 * k] = raw[3 * idx[k]] * h->p.dt;           // logged step index -> time of the step's start
        rows3[3 * k + 1] = raw[3 * idx[k] + 1];
        rows3[3 * k + 2] = raw[3 * idx[k] + 2];
    }
    return APS_OK;
}

int aps_resort(aps_handle *h) {
    if (!h) return APS_ERR_ARG;
    if (!h->p.sort_by_site || is_tiles(h)) return APS_OK;   // the tile kernel does not care about the slot order
    for (int e = 0; e < h->E; ++e) {
        const int64_t n = h->n_set[(size_t)e];
        if (n < 0) continue;
        std::vector<int32_t> pos((size_t)n); std::vector<int8_t> sg((size_t)n); std::vector<uint8_t> bd((size_t)n), al((size_t)n);
        int rc = aps_get_state(h, e, pos.data(), sg.data(), bd.data(), al.data(), n);
        if (rc) return rc;
        std::vector<uint32_t> ref_orig((size_t)h->Npad);      // slot -> particle map before the re-sort
        HIP_TRY(h, hipMemcpy(ref_orig.data(), h->d_orig + (size_t)e * h->Npad, ref_orig.size() * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> src, orig; long long gsum[2];
        pack_ensemble(h, pos.data(), sg.data(), bd.data(), al.data(), n, src, orig, gsum);
        if ((rc = upload_ensemble(h, e, src, orig))) return rc;
        if (!h->ref_set.empty() && h->ref_set[(size_t)e]) {   // the reference follows the particles into the new slot order
            std::vector<uint32_t> ref((size_t)h->Npad), oldorig((size_t)h->Npad), byorig((size_t)n, DEAD_BIT), neu((size_t)h->Npad, DEAD_BIT);
            HIP_TRY(h, hipMemcpy(ref.data(), h->d_ref + (size_t)e * h->Npad, ref.size() * 4, hipMemcpyDeviceToHost));
            for (int64_t sl = 0; sl < h->Npad; ++sl) if (ref_orig[(size_t)sl] != 0xFFFFFFFFu) byorig[ref_orig[(size_t)sl]] = ref[(size_t)sl];
            for (int64_t sl = 0; sl < h->Npad; ++sl) if (orig[(size_t)sl] != 0xFFFFFFFFu) neu[(size_t)sl] = byorig[orig[(size_t)sl]];
            HIP_TRY(h, hipMemcpy(h->d_ref + (size_t)e * h->Npad, neu.data(), neu.size() * 4, hipMemcpyHostToDevice));
        }
    }
    return APS_OK;
}

int aps_observe(aps_handle *h, int32_t e, int64_t *counts_p, int64_t *counts_m, double *m_field) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E) return fail(h, APS_ERR_ARG, "aps_observe: bad ensemble");
    if (h->n_set[(size_t)e] < 0) return fail(h, APS_ERR_STATE, "aps_observe: no state uploaded for this ensemble");
    { int rc_ = sync_slots(h); if (rc_) return rc_; }
    const int L = h->p.L;
    if (counts_p || counts_m) {
        std::vector<uint32_t> src((size_t)h->Npad);
        HIP_TRY(h, hipMemcpyAsync(src.data(), h->d_src + (size_t)e * h->Npad, src.size() * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (counts_p) std::fill(counts_p, counts_p + L, 0);
        if (counts_m) std::fill(counts_m, counts_m + L, 0);
        for (uint32_t w : src) {
            if (w & DEAD_BIT) continue;
            if (w & SPIN_BIT) { if (counts_p) counts_p[w & POS_MASK]++; }
            else if (counts_m) counts_m[w & POS_MASK]++;
        }
    }
    if (m_field) {
        int rc = launch_field(h, e, h->d_sp8 + (size_t)e * h->Npad, h->d_tinfo + (size_t)e * h->ntiles, (int)h->ntiles, h->d_mfield);
        if (rc) return rc;
        HIP_TRY(h, hipMemcpyAsync(m_field, h->d_mfield, (size_t)L * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return APS_OK;
}

int aps_field_from_counts(aps_handle *h, int32_t e, const int64_t *counts_p, const int64_t *counts_m, double *m_field) {
    if (!h) return APS_ERR_ARG;
    if (e < 0 || e >= h->E || !counts_p || !counts_m || !m_field) return fail(h, APS_ERR_ARG, "aps_field_from_counts: bad argument");
    const int L = h->p.L;
    std::vector<uint32_t> src;
    long long gs[2] = {0, 0};
    for (int x = 0; x < L; ++x) {
        if (counts_p[x] < 0 || counts_m[x] < 0) return fail(h, APS_ERR_ARG, "aps_field_from_counts: negative count");
        for (int64_t k = 0; k < counts_p[x]; ++k) src.push_back((uint32_t)x | SPIN_BIT);
        for (int64_t k = 0; k < counts_m[x]; ++k) src.push_back((uint32_t)x);
        gs[0] += counts_p[x] - counts_m[x]; gs[1] += counts_p[x] + counts_m[x];
    }
    if (gs[1] > (long long)4 * h->p.K * L) return fail(h, APS_ERR_ARG, "aps_field_from_counts: more than 4*K*L particles would break the exact-sum bound");
    const size_t nt = (src.size() + TILE - 1) / TILE + 1;
    src.resize(nt * TILE, DEAD_BIT);
    std::vector<uint32_t> sp8; std::vector<int4> tinfo;
    derive_sources(src, sp8, tinfo);
    if (nt > h->tmp_cap) {
        if (h->d_tmp_sp8) {
            (void)hipFree(h->d_tmp_sp8); (void)hipFree(h->d_tmp_tinfo);
            h->d_tmp_sp8 = nullptr; h->d_tmp_tinfo = nullptr;
        }
        int rc;
        if ((rc = dev_alloc(h, &h->d_tmp_sp8, nt * TILE)) || (rc = dev_alloc(h, &h->d_tmp_tinfo, nt))) return rc;
        h->tmp_cap = nt;
    }
    long long saved[2];
    long long *gcur = h->d_gsum + (size_t)(h->step & 1) * 2 * h->E + 2 * e;
    HIP_TRY(h, hipMemcpyAsync(saved, gcur, sizeof(saved), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_tmp_sp8, sp8.data(), sp8.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_tmp_tinfo, tinfo.data(), tinfo.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpyAsync(gcur, gs, sizeof(gs), hipMemcpyHostToDevice, h->stream));
    int rc = launch_field(h, e, h->d_tmp_sp8, h->d_tmp_tinfo, (int)nt, h->d_mfield);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(m_field, h->d_mfield, (size_t)L * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(gcur, saved, sizeof(saved), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return APS_OK;
}

}  // extern "C"
