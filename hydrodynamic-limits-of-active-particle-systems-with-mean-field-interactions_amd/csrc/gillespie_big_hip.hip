// gillespie_big_hip.hip -- gil_run_large of include/gillespie.h: the reference's exact event loop
// (PARTICLE_solver_CLASS.py:511-538) for ONE system far too large for a workgroup's LDS (the BASELINE size:
// N = 1e5 particles on L = 2e5 sites, where the reference manages 0.8 events/s because it recomputes the whole
// field and all N rates before every event).
//
// One persistent workgroup of 1024 threads; the state lives in global memory (L2-resident), only the weight table
// and small work areas in LDS.  What makes an event cheap:
//   * the smoothed histograms W, S are kept incrementally (exact weight grid) -- an event changes them on
//     the sites within the table's reach of one or two sites;
//   * a site -> particle map (K slots per site) finds the particles whose rates that changes without scanning all N;
//   * rates are summed in two levels (blocks of 256 particles, then the block sums): only the blocks holding
//     re-evaluated particles are re-summed, the choice of the particle descends the two levels.
// Event semantics, threshold order, exit handling and observation timing are those of gillespie_hip.hip (same
// channels() device function, same draws).  Parity: same-uniforms trajectories against the oracle and against the
// LDS-resident kernel (tests/test_gpu_gillespie.py).

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "gillespie.h"
#include "aps_common.hpp"

namespace {

constexpr int BT = 1024, BW = BT / 64;      // threads, waves of the workgroup
constexpr int PB = 256;                     // particles per rate block
constexpr int MAX_NB = 4096;                // rate blocks (N <= 2^20)
constexpr int TAB_LDS_MAX = 10000;          // table entries kept in LDS
std::string g_big_err;
enum { F_PLUS = 1, F_BOUND = 2, F_ALIVE = 4 };

struct BigArgs {
    Model m;
    gil_params p;
    int tlen, n_init, nblk, cb, tab_in_lds;
    double beta;
    const double *table, *times, *uniforms;
    const uint8_t *anchor;
    const int32_t *pos0; const int8_t *sigma0; const uint8_t *bound0;
    // global scratch
    int *pos, *occ, *occp, *slot, *work;
    uint8_t *flg;
    double *rate, *bsum, *W, *S;
    // outputs
    int32_t *pos_obs; int8_t *sigma_obs; uint8_t *flags_obs;
    int32_t *n_recorded; long long *n_events; double *t_final, *exits; int32_t *n_exits;
};

// W, S from scratch on all sites (one-time): a thread per site over all particles
__global__ __launch_bounds__(256) void big_field_init(const BigArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.m.L) return;
    double w = 0.0, s = 0.0;
    if (a.m.field_mode)
        for (int j = 0; j < a.n_init; ++j) {
            const double g = site_weight(a.m, a.table, a.tlen, x, a.pos0[j]);
            w += g; s += a.sigma0[j] > 0 ? g : -g;
        }
    a.W[x] = w; a.S[x] = s;
}

__device__ inline double big_rate(const BigArgs &a, const double *tab, int i, long long gs, long long gn) {
    (void)tab;
    const Model &M = a.m;
    const int L = M.L, p = a.pos[i];
    const uint8_t f = a.flg[i];
    double w, s;
    if (M.field_mode) { w = a.W[p]; s = a.S[p]; } else { w = (double)gn; s = (double)gs; }
    double mloc = 0.0;
    if (w > 0.0) { mloc = s / w; mloc = mloc > 1.0 ? 1.0 : (mloc < -1.0 ? -1.0 : mloc); }
    int l = p - 1, r = p + 1;
    if (M.periodic) { l = l < 0 ? l + L : l; r = r >= L ? r - L : r; }
    return channels(M, a.anchor ? a.anchor[p] != 0 : false, p, (f & F_PLUS) ? 1 : -1, (f & F_BOUND) != 0, mloc, a.beta,
                    a.occ[p], l >= 0 ? a.occ[l] : 0, r < L ? a.occ[r] : 0).total;
}

// lowest / highest thread of the workgroup for which `flag` holds, into *lo / *hi: one LDS atomic per wavefront
// (hundreds of same-address atomics from single lanes serialise at ~25 cycles each)
__device__ inline void note_first_last(bool flag, int *lo, int *hi) {
    const unsigned long long m = __ballot(flag);
    if (m && (threadIdx.x & 63) == 0) {
        const int base = (int)(threadIdx.x & ~63u);
        atomicMin(lo, base + __builtin_ctzll(m));
        atomicMax(hi, base + 63 - __builtin_clzll(m));
    }
}

// inclusive scan over the 1024 threads; `xw` = BW doubles of LDS
__device__ inline double block_scan_inclusive(double v, double *xw) {
    double inc = wave_scan_inclusive(v);
    const int t = threadIdx.x;
    __syncthreads();
    if ((t & 63) == 63) xw[t >> 6] = inc;
    __syncthreads();
    double before = 0.0;
    for (int w = 0; w < (t >> 6); ++w) before += xw[w];
    return inc + before;
}

__global__ __launch_bounds__(BT) void gil_big_kernel(const BigArgs a) {
    extern __shared__ double lds[];
    const Model &M = a.m;
    const int L = M.L, K = M.K, t = threadIdx.x, lane = t & 63, wave = t >> 6, nobs = a.p.n_obs, N = a.p.n_cap;
    double *tabl = lds;                                        // [tlen + 1] when it fits
    double *xw = tabl + (a.tab_in_lds ? ((a.tlen + 2) & ~1) : 0);   // [32] cross-wave scratch
    double *draws = xw + 32;                                   // [BT][4]
    double *dsel = draws + 4 * BT;                             // [4] cumulative rate before the chosen block etc.
    int *bflag = reinterpret_cast<int *>(dsel + 4);            // [MAX_NB] block needs re-summing
    int *blist = bflag + MAX_NB;                               // [MAX_NB] list of those blocks
    int *ctl = blist + MAX_NB;                                 // [32]
    const double *tab = a.tab_in_lds ? tabl : a.table;
    if (a.tab_in_lds) for (int i = t; i <= a.tlen; i += BT) tabl[i] = a.table[i];
    for (int j = t; j < MAX_NB; j += BT) bflag[j] = 0;
    // ---- load the system: particles, occupancy, site -> particle map
    for (int x = t; x < L; x += BT) { a.occ[x] = 0; a.occp[x] = 0; }
    for (size_t q = t; q < (size_t)L * K; q += BT) a.slot[q] = -1;
    __syncthreads();
    long long ls = 0, ln = 0;
    for (int i = t; i < N; i += BT) {
        const bool live = i < a.n_init;
        const int p = live ? a.pos0[i] : 0;
        const uint8_t f = live ? (uint8_t)(F_ALIVE | (a.sigma0[i] > 0 ? F_PLUS : 0) | ((a.bound0 && a.bound0[i]) ? F_BOUND : 0)) : 0;
        a.pos[i] = p; a.flg[i] = f; a.rate[i] = 0.0;
        if (live) {
            a.slot[(size_t)p * K + atomicAdd(&a.occ[p], 1)] = i;
            if (f & F_PLUS) atomicAdd(&a.occp[p], 1);
            ls += (f & F_PLUS) ? 1 : -1; ln += 1;
        }
    }
    __syncthreads();
    // global-mean mode: sum of spins, particles alive (every thread holds the totals)
    long long gsum_s = 0, gsum_n = 0;
    {
        long long *xl = reinterpret_cast<long long *>(xw);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { ls += __shfl_xor(ls, off); ln += __shfl_xor(ln, off); }
        if (lane == 0) { xl[wave] = ls; xl[BW + wave] = ln; }
        __syncthreads();
        for (int w = 0; w < BW; ++w) { gsum_s += xl[w]; gsum_n += xl[BW + w]; }
        __syncthreads();
    }
    for (int i = t; i < a.n_init; i += BT) a.rate[i] = big_rate(a, tab, i, gsum_s, gsum_n);
    __syncthreads();
    for (int j = wave; j < a.nblk; j += BW) {                  // block sums
        double v = 0.0;
        for (int k = lane; k < PB; k += 64) { const int i = j * PB + k; if (i < N) v += a.rate[i]; }
        v = wave_scan_inclusive(v);
        if (lane == 63) a.bsum[j] = v;
    }
    __syncthreads();
    double tnow = 0.0, t_next = nobs > 1 ? a.times[1] : INFINITY;
    long long n_ev = 0, ev_base = 0;
    int k_obs = 0, n_exit = 0;

    auto record = [&](int k) {                                 // state at observation k (ref :517-524)
        const size_t o = (size_t)k * N;
        for (int i = t; i < N; i += BT) {
            const uint8_t f = a.flg[i];
            if (a.pos_obs) a.pos_obs[o + i] = a.pos[i];
            if (a.sigma_obs) a.sigma_obs[o + i] = (f & F_PLUS) ? 1 : -1;
            if (a.flags_obs) a.flags_obs[o + i] = (uint8_t)(((f & F_BOUND) ? 1 : 0) | ((f & F_ALIVE) ? 2 : 0));
        }
    };
    record(0);
    k_obs = 1;
    bool have_event = false, dirty_all = false;
    int ev_a = 0, ev_b = 0;
#ifdef APS_STAMPS
    unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, s0 = __builtin_amdgcn_s_memtime();
#define BSTAMP(k) { const unsigned long long s1_ = __builtin_amdgcn_s_memtime(); st[k] += s1_ - s0; s0 = s1_; }
#else
#define BSTAMP(k)
#endif
    const int reach = (M.field_mode ? a.tlen - 1 : 0) + 1;
    while (tnow < a.p.T && k_obs < nobs && n_ev < a.p.max_events) {
        // ---- A: re-evaluate the rates the previous event changed
        if (have_event) {
            if (t == 0) { ctl[0] = 0; ctl[1] = 0; }
            __syncthreads();
            int lo, len;
            if (dirty_all) { lo = 0; len = 0; }                // handled below: every particle
            else {
                const bool wrapped = M.periodic && (ev_a - ev_b > 1 || ev_b - ev_a > 1);
                lo = (wrapped ? max(ev_a, ev_b) : min(ev_a, ev_b)) - reach;
                len = 2 * reach + 2;
                if (len >= L || (!M.periodic && reach >= L)) { lo = 0; len = L; }
            }
            auto append = [&](bool flag, int item) {           // one LDS atomic per wavefront: ballot + mbcnt compaction
                const unsigned long long m = __ballot(flag);
                if (!m) return;
                int base = 0;
                if (lane == 0) base = atomicAdd(&ctl[0], __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (flag) a.work[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = item;
            };
            if (dirty_all) {
                for (int i0 = 0; i0 < N; i0 += BT) {
                    const int i = i0 + t;
                    append(i < N && (a.flg[i] & F_ALIVE), i);
                }
            } else {
                constexpr int U = 4;                           // site iterations whose occupancy loads are in flight together
                for (int k0 = 0; k0 < len; k0 += U * BT) {     // sites in reach -> their particles, through the map
                    int xs[U], ns[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = k0 + u * BT + t;
                        int x = lo + k;
                        if (M.periodic) { x %= L; if (x < 0) x += L; }
                        const bool in = k < len && x >= 0 && x < L;
                        xs[u] = in ? x : 0;
                        ns[u] = in ? a.occ[x] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        for (int q = 0; __ballot(q < ns[u]); ++q) append(q < ns[u], q < ns[u] ? a.slot[(size_t)xs[u] * K + q] : 0);
                }
            }
            __syncthreads();
            BSTAMP(0)
            const int nwork = ctl[0];
            for (int base = 0; base < nwork; base += 4 * BT) {   // the list entries of four items per thread are fetched together
                int is[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int j = base + u * BT + t; is[u] = j < nwork ? a.work[j] : -1; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = is[u];
                    if (i < 0) continue;
                    a.rate[i] = big_rate(a, tab, i, gsum_s, gsum_n);
                    if (atomicExch(&bflag[i / PB], 1) == 0) blist[atomicAdd(&ctl[1], 1)] = i / PB;
                }
            }
            __syncthreads();
            const int nb_dirty = ctl[1];
            for (int q = wave; q < nb_dirty; q += BW) {        // re-sum the touched blocks
                const int j = blist[q];
                double v = 0.0;
                for (int k = lane; k < PB; k += 64) { const int i = j * PB + k; if (i < N) v += a.rate[i]; }
                v = wave_scan_inclusive(v);
                if (lane == 63) { a.bsum[j] = v; bflag[j] = 0; }
            }
            __syncthreads();
        }
        BSTAMP(1)
        // ---- B: total rate, draws, choice of the block and of the particle
        double mine = 0.0;
        for (int j = t * a.cb; j < min(a.nblk, (t + 1) * a.cb); ++j) mine += a.bsum[j];
        const double inc = block_scan_inclusive(mine, xw);
        BSTAMP(5)
        if (t == BT - 1) dsel[0] = inc;
        if ((n_ev - ev_base) >= BT || n_ev == 0) {
            ev_base = n_ev;
            const long long evn = n_ev + t;
            double u0, u1, u2, u3;
            if (a.uniforms) {
                const bool in = evn < a.p.max_events;
                const double *src = a.uniforms + (size_t)(in ? evn : 0) * 4;
                u0 = in ? src[0] : 0.0; u1 = in ? src[1] : 0.0; u2 = in ? src[2] : 0.0; u3 = in ? src[3] : 0.0;
            } else {
                uint32_t x[4], y[4];
                philox4x32_10((uint32_t)evn, (uint32_t)(evn >> 32), 0u, 0x47494C31u, M.seed_lo, M.seed_hi, x);
                philox4x32_10((uint32_t)evn, (uint32_t)(evn >> 32), 0u, 0x47494C32u, M.seed_lo, M.seed_hi, y);
                u0 = ((double)(x[0] >> 5) * 67108864.0 + (double)(x[1] >> 6)) * 0x1.0p-53;
                u1 = ((double)(x[2] >> 5) * 67108864.0 + (double)(x[3] >> 6)) * 0x1.0p-53;
                u2 = ((double)(y[0] >> 5) * 67108864.0 + (double)(y[1] >> 6)) * 0x1.0p-53;
                u3 = ((double)(y[2] >> 5) * 67108864.0 + (double)(y[3] >> 6)) * 0x1.0p-53;
            }
            draws[4 * t] = -log1p(-u0); draws[4 * t + 1] = u1; draws[4 * t + 2] = u2; draws[4 * t + 3] = u3;
        }
        if (t == 0) { ctl[2] = BT; ctl[3] = -1; ctl[4] = PB; ctl[5] = -1; ctl[13] = -1; ctl[14] = BT; ctl[15] = -1; ctl[16] = BT; }
        __syncthreads();
        BSTAMP(6)
        const double R = dsel[0];
        if (!(R > 0.0)) { tnow = INFINITY; break; }           // ref :355
        const double *dr = draws + 4 * (int)(n_ev - ev_base);
        const double tau = (1.0 / R) * dr[0], target = dr[1] * R, u2 = dr[2], u3 = dr[3];
        note_first_last(inc > target && mine > 0.0, &ctl[2], &ctl[13]);
        note_first_last(mine > 0.0, &ctl[14], &ctl[3]);
        __syncthreads();
        BSTAMP(7)
        const int tsel = ctl[2] < BT ? ctl[2] : ctl[3];        // target rounded past the total: last thread with any rate
        if (t == tsel) {                                       // which of this thread's blocks
            double run = inc - mine;
            int jsel = -1;
            double before = run;
            for (int j = t * a.cb; j < min(a.nblk, (t + 1) * a.cb); ++j) {
                const double b = a.bsum[j];
                if (b > 0.0) { jsel = j; before = run; run += b; if (run > target) break; }
            }
            ctl[6] = jsel; dsel[1] = before;
        }
        __syncthreads();
        BSTAMP(8)
        const int jsel = ctl[6];
        {                                                      // the particle inside the block: its first 256 threads scan it
            double r = 0.0;
            const int i = jsel * PB + t;
            if (t < PB && i < N) r = a.rate[i];
            const double binc = block_scan_inclusive(r, xw) + dsel[1];
            note_first_last(t < PB && r > 0.0 && binc > target, &ctl[4], &ctl[15]);
            note_first_last(t < PB && r > 0.0, &ctl[16], &ctl[5]);
            __syncthreads();
        }
        const int isel = jsel * PB + (ctl[4] < PB ? ctl[4] : ctl[5]);
        BSTAMP(2)
        // ---- C: one thread applies the event (ref :363-446) and keeps the site map
        if (t == 0) {
            const int i = isel, p = a.pos[i];
            uint8_t f = a.flg[i];
            const bool plus = (f & F_PLUS) != 0;
            double w, s;
            if (M.field_mode) { w = a.W[p]; s = a.S[p]; } else { w = (double)gsum_n; s = (double)gsum_s; }
            double mloc = 0.0;
            if (w > 0.0) { mloc = s / w; mloc = mloc > 1.0 ? 1.0 : (mloc < -1.0 ? -1.0 : mloc); }
            int l = p - 1, rr = p + 1;
            if (M.periodic) { l = l < 0 ? l + L : l; rr = rr >= L ? rr - L : rr; }
            const Channels c = channels(M, a.anchor ? a.anchor[p] != 0 : false, p, plus ? 1 : -1, (f & F_BOUND) != 0, mloc, a.beta,
                                        a.occ[p], l >= 0 ? a.occ[l] : 0, rr < L ? a.occ[rr] : 0);
            const double v = u2 * c.total;
            const double e_diff = c.diff, e_act = e_diff + c.act, e_bind = e_act + c.bind, e_unbind = e_bind + c.unbind,
                         e_exit = e_unbind + c.leave;
            int kind = 0, to = p;                              // 0 nothing, 1 hop, 2 flip, 3 exit
            if (v < e_diff) {
                if (c.left + c.right > 0.0) { kind = 1; to = (u3 < c.left / (c.left + c.right)) ? p - 1 : p + 1; }
            } else if (v < e_act) { kind = 1; to = p + 1; }
            else if (v < e_bind) f |= F_BOUND;
            else if (v < e_unbind) f &= (uint8_t)~F_BOUND;
            else if (v < e_exit) kind = 3;
            else kind = 2;
            auto unmap = [&](int site) {                       // take particle i out of the site's slots
                const int n = a.occ[site];
                for (int q = 0; q < n; ++q)
                    if (a.slot[(size_t)site * K + q] == i) { a.slot[(size_t)site * K + q] = a.slot[(size_t)site * K + n - 1]; break; }
                a.slot[(size_t)site * K + n - 1] = -1;
                a.occ[site] = n - 1;
            };
            if (kind == 1) {
                if (M.periodic) to = to < 0 ? to + L : (to >= L ? to - L : to);
                else to = to < 0 ? 0 : (to > L - 1 ? L - 1 : to);
                if (to == p) kind = 0;                         // clipped at a wall: nothing moved
                else {
                    unmap(p);
                    a.slot[(size_t)to * K + a.occ[to]] = i; a.occ[to] += 1;
                    if (plus) { a.occp[p] -= 1; a.occp[to] += 1; }
                    a.pos[i] = to;
                }
            } else if (kind == 2) {
                f ^= F_PLUS;
                a.occp[p] += plus ? -1 : 1;
            } else if (kind == 3) {
                f &= (uint8_t)~F_ALIVE;
                a.rate[i] = 0.0;
                unmap(p);
                if (plus) a.occp[p] -= 1;
                if (a.exits && n_exit < N) {
                    double *row = a.exits + (size_t)n_exit * 3;
                    row[0] = tnow; row[1] = (double)p; row[2] = (double)i;
                }
            }
            a.flg[i] = f;
            ctl[8] = kind; ctl[9] = p; ctl[10] = to; ctl[11] = plus ? 1 : -1; ctl[12] = i;
        }
        __syncthreads();
        const int kind = ctl[8], p_old = ctl[9], p_new = ctl[10], sg = ctl[11];
        if (kind == 3) n_exit += 1;
        BSTAMP(3)
        // ---- D: the event's change of the smoothed histograms
        if (!M.field_mode) {
            if (kind == 2) gsum_s -= 2 * sg;
            else if (kind == 3) { gsum_s -= sg; gsum_n -= 1; }
        } else if (kind != 0) {
            const int Rt = a.tlen - 1;
            const int centre = kind == 1 ? min(p_old, p_new) : p_old, span = kind == 1 ? 1 : 0;
            const bool wrap1 = kind == 1 && M.periodic && (p_old - p_new > 1 || p_new - p_old > 1);
            int lo = centre - Rt, len = 2 * Rt + 1 + span;
            if (wrap1 || len >= L || (!M.periodic && Rt >= L)) { lo = 0; len = L; }
            constexpr int U = 4;                               // sites per thread whose loads are in flight together
            for (int k0 = 0; k0 < len; k0 += U * BT) {
                int xs[U]; double ws[U], ss[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u * BT + t;
                    int x = lo + k;
                    if (M.periodic) { x %= L; if (x < 0) x += L; }
                    const bool in = k < len && x >= 0 && x < L;
                    xs[u] = in ? x : -1;
                    ws[u] = in ? a.W[x] : 0.0; ss[u] = in ? a.S[x] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int x = xs[u];
                    if (x >= 0) {
                        const double g0 = site_weight(M, tab, a.tlen, x, p_old);
                        if (kind == 1) {
                            const double g1 = site_weight(M, tab, a.tlen, x, p_new), d = g1 - g0;  // exact on the weight grid
                            a.W[x] = ws[u] + d; a.S[x] = ss[u] + (sg > 0 ? d : -d);
                        } else if (kind == 2) {
                            a.S[x] = ss[u] - (sg > 0 ? 2.0 * g0 : -2.0 * g0);
                        } else {
                            a.W[x] = ws[u] - g0; a.S[x] = ss[u] - (sg > 0 ? g0 : -g0);
                        }
                    }
                }
            }
        }
        // a particle that left is no longer in the site map, so the next event's work list will not reach its block:
        // re-sum that block (its rate is zero now) here
        if (kind == 3 && wave == 0) {
            const int j = ctl[12] / PB;
            double v = 0.0;
            for (int k = lane; k < PB; k += 64) { const int i = j * PB + k; if (i < N) v += a.rate[i]; }
            v = wave_scan_inclusive(v);
            if (lane == 63) a.bsum[j] = v;
        }
        __syncthreads();
        BSTAMP(4)
        have_event = true;
        ev_a = p_old; ev_b = p_new;
        dirty_all = !M.field_mode && (kind == 2 || kind == 3);
        // ---- E: time and observations (ref :514-538)
        n_ev += 1;
        tnow += tau;
        if (tnow > a.p.T) break;
        while (k_obs < nobs && t_next <= tnow) { record(k_obs); ++k_obs; t_next = k_obs < nobs ? a.times[k_obs] : INFINITY; }
    }
#ifdef APS_STAMPS
    if (t == 0 && a.exits) for (int k = 0; k < 12; ++k) a.exits[k] = (double)st[k];   // diagnostic build only
#endif
    if (t == 0) {
        if (a.n_recorded) a.n_recorded[0] = k_obs;
        if (a.n_events) a.n_events[0] = n_ev;
        if (a.t_final) a.t_final[0] = tnow;
        if (a.n_exits) a.n_exits[0] = n_exit;
    }
}

struct DevB {
    std::vector<void *> ptrs;
    ~DevB() { for (void *q : ptrs) (void)hipFree(q); }
    template <typename T> T *alloc(size_t n) {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        (void)hipMemset(q, 0, std::max<size_t>(n, 1) * sizeof(T));
        ptrs.push_back(q);
        return static_cast<T *>(q);
    }
    template <typename T> T *upload(const T *src, size_t n) {
        T *q = alloc<T>(n);
        if (q && n && hipMemcpy(q, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return q;
    }
};

}  // namespace

extern "C" {

int gil_run_large(const gil_params *p, int32_t n0, const int32_t *pos0, const int8_t *sigma0, const uint8_t *bound0,
                  const double *uniforms, int32_t *pos_obs, int8_t *sigma_obs, uint8_t *flags_obs, int32_t *n_recorded,
                  int64_t *n_events, double *t_final, double *exits, int32_t *n_exits, double *kernel_ms) {
    auto bad = [&](const char *m) { g_big_err = std::string("gil_run_large: ") + m; return GIL_ERR_ARG; };
    if (!p || !pos0 || !sigma0 || !p->beta || !p->times_obs) return bad("null argument");
    if (p->n_systems != 1) return bad("one system per call");
    if (p->L < 2 || p->L > (1 << 25)) return bad("L must be in [2, 2^25]");
    if (p->K < 1 || p->K > 32) return bad("site capacity K must be in [1, 32]");
    if ((int64_t)p->L * p->K > (1ll << 27)) return bad("L * K must not exceed 2^27 (site map)");
    if (p->n_cap < 1 || p->n_cap > MAX_NB * PB || n0 < 0 || n0 > p->n_cap || p->n_obs < 1 || p->max_events < 0) return bad("bad n_cap / n0 / n_obs / max_events");
    const int L = p->L, N = p->n_cap;
    {
        std::vector<int> occ((size_t)L, 0);
        for (int i = 0; i < n0; ++i) {
            if (pos0[i] < 0 || pos0[i] >= L) return bad("position outside [0, L)");
            if (++occ[(size_t)pos0[i]] > p->K) return bad("site capacity exceeded");
            if (sigma0[i] != 1 && sigma0[i] != -1) return bad("sigma must be +1 or -1");
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_big_err = "gil_run_large: no HIP device"; return GIL_ERR_NODEVICE; }
    if (p->device < 0 || p->device >= ndev) return bad("device ordinal out of range");
    if (hipSetDevice(p->device) != hipSuccess) { g_big_err = "hipSetDevice failed"; return GIL_ERR_HIP; }
    std::vector<double> table; int tlen = 0, q = 0;
    weight_table(p->sigma_grid, L, p->K, p->periodic != 0, table, tlen, q);
    BigArgs a{};
    a.p = *p; a.tlen = tlen; a.n_init = n0; a.nblk = (N + PB - 1) / PB; a.cb = (a.nblk + BT - 1) / BT;
    a.tab_in_lds = tlen + 1 <= TAB_LDS_MAX ? 1 : 0; a.beta = p->beta[0];
    Model &M = a.m;
    M.L = L; M.K = p->K; M.periodic = p->periodic ? 1 : 0; M.field_mode = p->sigma_grid > 0.0 ? 1 : 0;
    M.minus_anchor = p->minus_anchor ? 1 : 0; M.immobilize = p->immobilize ? 1 : 0; M.suppress_flip = p->suppress_flip ? 1 : 0;
    M.crowding = p->crowding ? 1 : 0; M.rate_diffusion = p->rate_diffusion; M.rate_active = p->rate_active;
    M.k_on = p->k_on; M.k_off = p->k_off; M.k_exit = p->k_exit; M.dt = 0.0;
    M.seed_lo = (uint32_t)p->seed; M.seed_hi = (uint32_t)(p->seed >> 32); M.ens_base = 0;
    DevB d;
    const size_t SO = (size_t)p->n_obs * N;
#define UPB(dst, src, n) do { a.dst = d.upload(src, n); if (!a.dst) { g_big_err = "gil_run_large: device upload failed (" #dst ")"; return GIL_ERR_HIP; } } while (0)
#define ALB(dst, T, n) do { a.dst = d.alloc<T>(n); if (!a.dst) { g_big_err = "gil_run_large: device allocation failed (" #dst ")"; return GIL_ERR_HIP; } } while (0)
    UPB(table, table.data(), table.size()); UPB(times, p->times_obs, (size_t)p->n_obs);
    UPB(pos0, pos0, (size_t)std::max(n0, 1)); UPB(sigma0, sigma0, (size_t)std::max(n0, 1));
    if (bound0) UPB(bound0, bound0, (size_t)std::max(n0, 1));
    if (p->anchor_mask) UPB(anchor, p->anchor_mask, (size_t)L);
    if (uniforms) UPB(uniforms, uniforms, (size_t)p->max_events * 4);
    M.flip_n = 0; M.flip_tab = nullptr;
    if (p->flip_table) {                                       // a caller's flip_rate_fn, tabulated (aps_set_flip_table's layout)
        if (p->flip_n < 1 || p->flip_n > (1 << 24)) return bad("flip_n must be in [1, 2^24]");
        M.flip_tab = d.upload(p->flip_table, (size_t)2 * ((size_t)p->flip_n + 1));
        if (!M.flip_tab) { g_big_err = "gil_run_large: device upload failed (flip_table)"; return GIL_ERR_HIP; }
        M.flip_n = p->flip_n;
    }
    ALB(pos, int, (size_t)N); ALB(occ, int, (size_t)L); ALB(occp, int, (size_t)L); ALB(slot, int, (size_t)L * p->K); ALB(work, int, (size_t)N);
    ALB(flg, uint8_t, (size_t)N); ALB(rate, double, (size_t)N); ALB(bsum, double, (size_t)a.nblk); ALB(W, double, (size_t)L); ALB(S, double, (size_t)L);
    if (pos_obs) ALB(pos_obs, int32_t, SO);
    if (sigma_obs) ALB(sigma_obs, int8_t, SO);
    if (flags_obs) ALB(flags_obs, uint8_t, SO);
    if (n_recorded) ALB(n_recorded, int32_t, 1);
    if (n_events) ALB(n_events, long long, 1);
    if (t_final) ALB(t_final, double, 1);
    if (exits) ALB(exits, double, (size_t)N * 3);
    if (n_exits) ALB(n_exits, int32_t, 1);
#undef UPB
#undef ALB
    const size_t lds = ((size_t)(a.tab_in_lds ? ((tlen + 2) & ~1) : 0) + 32 + 4 * BT + 4) * sizeof(double) + ((size_t)2 * MAX_NB + 32) * sizeof(int);
    if (lds > 160 * 1024) return bad("LDS budget exceeded");
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(&gil_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        g_big_err = "gil_run_large: cannot raise the dynamic LDS limit"; return GIL_ERR_HIP;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { g_big_err = "hipEventCreate failed"; return GIL_ERR_HIP; }
    hipLaunchKernelGGL(big_field_init, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, nullptr, a);
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(gil_big_kernel, dim3(1), dim3(BT), lds, nullptr, a);
    (void)hipEventRecord(e1, nullptr);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipDeviceSynchronize();
    float ms = 0.f;
    if (err == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) { g_big_err = std::string("gil_big_kernel: ") + hipGetErrorString(err); return GIL_ERR_HIP; }
    if (kernel_ms) *kernel_ms = ms;
#define DNB(host, dev, bytes) do { if (host && hipMemcpy(host, a.dev, (bytes), hipMemcpyDeviceToHost) != hipSuccess) { g_big_err = "gil_run_large: download failed (" #dev ")"; return GIL_ERR_HIP; } } while (0)
    DNB(pos_obs, pos_obs, SO * 4); DNB(sigma_obs, sigma_obs, SO); DNB(flags_obs, flags_obs, SO);
    DNB(n_recorded, n_recorded, 4); DNB(n_events, n_events, 8); DNB(t_final, t_final, 8); DNB(exits, exits, (size_t)N * 3 * 8); DNB(n_exits, n_exits, 4);
#undef DNB
    return GIL_OK;
}

const char *gil_large_last_error(void) { return g_big_err.c_str(); }

}  // extern "C"
