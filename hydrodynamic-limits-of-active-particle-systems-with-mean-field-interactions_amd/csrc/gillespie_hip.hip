// gillespie_hip.hip -- MI355X (gfx950) implementation of the C ABI in include/gillespie.h.
//
// The reference's exact event loop (PARTICLE_solver_CLASS.py:511-538) for many independent systems at once: one
// PERSISTENT workgroup per system, the whole system in LDS (particles, site occupancy, the smoothed histograms
// W = tot_conv and S = s_conv, the weight table, the per-particle rates).  One loop iteration = one event:
//   A  every thread evaluates the rate table of its particles (ref :254-352) from m = clip(S/W)[pos] and the
//      occupancy of the neighbouring sites, workgroup scan of the totals -> R
//   B  waiting time ~ Exp(R), particle ~ rates / R (searchsorted over the running sums, like Generator.choice),
//      event type by the reference's threshold order diffuse < active < bind < unbind < exit < flip (ref :358-367)
//   C  one thread applies the event (ref :371-446)
//   D  all threads add the event's change of W, S on the sites in reach (the reference recomputes the whole field
//      before every event, :512; the weights sit on the exact grid of DESIGN.md, so the incremental sums equal a
//      recomputation bit for bit)
//   E  t += tau; states / scalar sums of the observation times that were crossed go to HBM (ref :517-536)
// Randomness: Philox4x32-10 keyed by the seed, counter (event index, system) -- or numbers supplied by the caller.

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

std::string g_gil_err;
enum { F_PLUS = 1, F_BOUND = 2, F_ALIVE = 4 };
enum { GS_N = 0, GS_SPIN, GS_POS, GS_WALL, GS_MAXPOS, GS_FRONT, GS_ATTEMPT, GS_BLOCKED, GS_DISP, GS_DISP2, GS_NDISP, GS_EVENTS };

struct GilArgs {
    Model m;
    gil_params p;
    int tlen, chunk;
    const double *beta, *table, *times, *uniforms;
    const uint8_t *anchor, *block_table;
    const int32_t *front_lo, *n0, *pos0;
    const int8_t *sigma0;
    const uint8_t *bound0;
    int32_t *pos_obs; int8_t *sigma_obs; uint8_t *flags_obs; long long *scalars;
    int32_t *n_recorded; long long *n_events; double *t_final, *exits; int32_t *n_exits;
};

template <int NT>
__device__ inline long long wg_sum_ll(long long v, long long *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    long long s = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += red[w];
    return s;
}

// NT = threads per system: one wavefront (no real barriers, six systems per CU by LDS) for small systems, four for large ones
template <int NT>
__global__ __launch_bounds__(NT) void gil_kernel(const GilArgs a) {
    extern __shared__ double lds[];
    const Model &M = a.m;
    const int L = M.L, K = M.K, t = threadIdx.x, sys = blockIdx.x, ncap = a.p.n_cap, nobs = a.p.n_obs;
    double *W = lds, *S = W + L, *tab = S + L, *rate = tab + ((a.tlen + 2) & ~1), *red = rate + ncap + (ncap & 1);
    double *tinc = red + 8;                                   // [NT] inclusive scan of the threads' rate sums
    double *draws = tinc + NT;                                // [NT][4] -log1p(-u0), u1, u2, u3 of the next NT events
    long long *redl = reinterpret_cast<long long *>(draws + 4 * NT);   // [8]
    int *pos = reinterpret_cast<int *>(redl + 8);             // [ncap]
    int *ref = pos + ncap;                                    // [ncap] positions at the reference observation
    int *work = ref + ncap;                                   // [ncap] particles whose rates are re-evaluated before this event
    int *ctl = work + ncap;                                   // [16] broadcast slots
    uint8_t *flg = reinterpret_cast<uint8_t *>(ctl + 16);     // [ncap]
    uint8_t *occ = flg + ((ncap + 15) & ~15);                 // [L] particles per site
    uint8_t *occp = occ + ((L + 15) & ~15);                   // [L] plus particles per site (blocking table)
    const double beta = a.beta[sys];
    const int n_init = a.n0[sys];
    // ---- load the system
    for (int i = t; i <= a.tlen; i += NT) tab[i] = a.table[i];
    for (int x = t; x < L; x += NT) { occ[x] = 0; occp[x] = 0; }
    for (int i = t; i < ncap; i += NT) {
        const bool live = i < n_init;
        pos[i] = live ? a.pos0[(size_t)sys * ncap + i] : 0;
        flg[i] = live ? (uint8_t)(F_ALIVE | (a.sigma0[(size_t)sys * ncap + i] > 0 ? F_PLUS : 0) |
                                  ((a.bound0 && a.bound0[(size_t)sys * ncap + i]) ? F_BOUND : 0)) : 0;
        ref[i] = -1;
        rate[i] = 0.0;                                         // empty and departed slots keep rate zero
    }
    __syncthreads();
    if (t == 0) for (int i = 0; i < n_init; ++i) { occ[pos[i]]++; if (flg[i] & F_PLUS) occp[pos[i]]++; }
    // field from scratch: W(x) = sum_j w(x, p_j), S(x) = sum_j sigma_j w(x, p_j)
    for (int x = t; x < L; x += NT) {
        double w = 0.0, s = 0.0;
        if (M.field_mode)
            for (int j = 0; j < n_init; ++j) {
                const double g = site_weight(M, tab, a.tlen, x, pos[j]);
                w += g; s += (flg[j] & F_PLUS) ? g : -g;
            }
        W[x] = w; S[x] = s;
    }
    __syncthreads();
    long long gsum_s = 0, gsum_n = 0;                          // global-mean mode: sum of spins, particles alive
    if (!M.field_mode) {
        long long ls = 0, ln = 0;
        for (int i = t; i < n_init; i += NT) { ls += (flg[i] & F_PLUS) ? 1 : -1; ln += 1; }
        gsum_s = wg_sum_ll<NT>(ls, redl); gsum_n = wg_sum_ll<NT>(ln, redl);
    }
    double tnow = 0.0;
    long long n_ev = 0;
    int k_obs = 0, n_exit = 0;
    const int c0 = t * a.chunk, c1 = min(ncap, c0 + a.chunk);

    auto record = [&](int k) {                                 // observation k: state and scalar sums (ref :517-536)
        const size_t o = ((size_t)sys * nobs + k) * ncap;
        long long v[GIL_NSCALARS] = {0, 0, 0, 0, -1, 0, 0, 0, 0, 0, 0, 0};
        if (k == a.p.ref_obs) for (int i = t; i < ncap; i += NT) ref[i] = (flg[i] & F_ALIVE) ? pos[i] : -1;
        __syncthreads();
        for (int i = t; i < ncap; i += NT) {
            const uint8_t f = flg[i];
            if (a.pos_obs) a.pos_obs[o + i] = pos[i];
            if (a.sigma_obs) a.sigma_obs[o + i] = (f & F_PLUS) ? 1 : -1;
            if (a.flags_obs) a.flags_obs[o + i] = (uint8_t)(((f & F_BOUND) ? 1 : 0) | ((f & F_ALIVE) ? 2 : 0));
            if (!(f & F_ALIVE)) continue;
            const int p = pos[i];
            v[GS_N] += 1; v[GS_SPIN] += (f & F_PLUS) ? 1 : -1; v[GS_POS] += p; v[GS_WALL] += p >= a.p.x_wall;
            v[GS_MAXPOS] = max(v[GS_MAXPOS], (long long)p);
            if ((f & F_PLUS) && p < L - 1) {
                v[GS_ATTEMPT] += 1;
                const int cp = occp[p + 1], cm = occ[p + 1] - occp[p + 1];
                v[GS_BLOCKED] += a.block_table ? a.block_table[cp * (K + 1) + cm] : (cp + cm >= 1);
            }
            if (ref[i] >= 0) { const long long d = (long long)p - ref[i]; v[GS_DISP] += d; v[GS_DISP2] += d * d; v[GS_NDISP] += 1; }
        }
        long long mx = v[GS_MAXPOS];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
        __syncthreads();
        if ((t & 63) == 0) redl[4 + (t >> 6)] = mx;
        __syncthreads();
        mx = redl[4];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) mx = max(mx, redl[4 + w]);
        if (a.front_lo && mx >= 0) {
            const int lo = a.front_lo[mx];
            for (int i = t; i < ncap; i += NT) if ((flg[i] & F_ALIVE) && pos[i] >= lo) v[GS_FRONT] += 1;
        }
        for (int q = 0; q < GIL_NSCALARS; ++q) {
            if (q == GS_MAXPOS || q == GS_EVENTS) continue;
            const long long s = wg_sum_ll<NT>(v[q], redl);
            if (t == 0 && a.scalars) a.scalars[((size_t)sys * nobs + k) * GIL_NSCALARS + q] = s;
        }
        if (t == 0 && a.scalars) {
            a.scalars[((size_t)sys * nobs + k) * GIL_NSCALARS + GS_MAXPOS] = mx;
            a.scalars[((size_t)sys * nobs + k) * GIL_NSCALARS + GS_EVENTS] = n_ev;
        }
        __syncthreads();
    };

    record(0);                                                 // ref :489-508
    k_obs = 1;
    double t_next = nobs > 1 ? a.times[1] : INFINITY;          // next observation time (kept in a register: no load per event)
#ifdef APS_STAMPS
    unsigned long long st[5] = {0, 0, 0, 0, 0}, s0 = __builtin_amdgcn_s_memtime();
#define GSTAMP(k) { const unsigned long long s1_ = __builtin_amdgcn_s_memtime(); st[k] += s1_ - s0; s0 = s1_; }
#else
#define GSTAMP(k)
#endif
    long long ev_base = 0;                                     // first event of the block of draws held in LDS
    bool dirty_all = true;                                     // first event: every rate is evaluated
    int dirty_a = 0, dirty_b = 0;
    const int dirty_reach = (M.field_mode ? a.tlen - 1 : 0) + 1;
    while (tnow < a.p.T && k_obs < nobs && n_ev < a.p.max_events) {
        // ---- A: rates (ref :254-352).  The reference recomputes every particle's rates before every event; here only
        // the particles whose inputs the last event changed are re-evaluated (field within the table's reach of the
        // event's sites, occupancy of the neighbouring sites) -- the others' rates are the values already in LDS.
        // (1) every thread lists its particles that need it (lanes that find none do not hold up the others: the
        //     expensive rate evaluation then runs once over the compacted list, a lane per listed particle)
        int nwork = 0;
        if (NT == 64) {                                        // one wavefront: ballot + mbcnt compaction, no LDS counter
            for (int k = 0; k < a.chunk; ++k) {
                const int i = c0 + k;
                bool redo = i < c1 && (flg[i] & F_ALIVE);      // the rate of a particle that left was zeroed when it left
                if (redo && !dirty_all) {
                    int d0 = pos[i] - dirty_a, d1 = pos[i] - dirty_b;
                    d0 = d0 < 0 ? -d0 : d0; d1 = d1 < 0 ? -d1 : d1;
                    if (M.periodic) { d0 = min(d0, L - d0); d1 = min(d1, L - d1); }
                    redo = min(d0, d1) <= dirty_reach;
                }
                const unsigned long long m = __ballot(redo);
                if (redo) work[nwork + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = i;
                nwork += __popcll(m);
            }
        } else {
            if (t == 0) ctl[6] = 0;
            __syncthreads();
            for (int i = c0; i < c1; ++i) {
                if (!(flg[i] & F_ALIVE)) continue;
                bool redo = dirty_all;
                if (!redo) {
                    int d0 = pos[i] - dirty_a, d1 = pos[i] - dirty_b;
                    d0 = d0 < 0 ? -d0 : d0; d1 = d1 < 0 ? -d1 : d1;
                    if (M.periodic) { d0 = min(d0, L - d0); d1 = min(d1, L - d1); }
                    redo = min(d0, d1) <= dirty_reach;
                }
                if (redo) work[atomicAdd(&ctl[6], 1)] = i;
            }
            __syncthreads();
            nwork = ctl[6];
        }
        __syncthreads();
        for (int j = t; j < nwork; j += NT) {
            const int i = work[j], p = pos[i];
            const uint8_t f = flg[i];
            double w, s;
            if (M.field_mode) { w = W[p]; s = S[p]; } else { w = (double)gsum_n; s = (double)gsum_s; }
            double mloc = 0.0;
            if (w > 0.0) { mloc = s / w; mloc = mloc > 1.0 ? 1.0 : (mloc < -1.0 ? -1.0 : mloc); }
            int l = p - 1, rr = p + 1;
            if (M.periodic) { l = l < 0 ? l + L : l; rr = rr >= L ? rr - L : rr; }
            rate[i] = channels(M, a.anchor ? a.anchor[p] != 0 : false, p, (f & F_PLUS) ? 1 : -1, (f & F_BOUND) != 0, mloc, beta,
                               occ[p], l >= 0 ? occ[l] : 0, rr < L ? occ[rr] : 0).total;
        }
        __syncthreads();
        double mine = 0.0;
        for (int i = c0; i < c1; ++i) mine += rate[i];
        double inc;                                            // inclusive scan over the threads
        if (NT == 64) inc = wave_scan_inclusive(mine);
        else {
            inc = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(inc, off); if ((t & 63) >= off) inc += o; }
            __syncthreads();
            if ((t & 63) == 63) red[t >> 6] = inc;
            __syncthreads();
            double before = 0.0;
            for (int w = 0; w < (t >> 6); ++w) before += red[w];
            inc += before;
        }
        tinc[t] = inc;
        GSTAMP(0)
        // ---- B: draws (ref :358-362): the numbers of NT consecutive events are produced together, a lane per event
        if ((n_ev - ev_base) >= NT || n_ev == 0) {
            ev_base = n_ev;
            const long long evn = n_ev + t;
            double u0, u1, u2, u3;
            if (a.uniforms) {
                const bool in = evn < a.p.max_events;
                const double *src = a.uniforms + ((size_t)sys * a.p.max_events + (in ? evn : 0)) * 4;
                u0 = in ? src[0] : 0.0; u1 = in ? src[1] : 0.0; u2 = in ? src[2] : 0.0; u3 = in ? src[3] : 0.0;
            } else {
                uint32_t x[4], y[4];
                philox4x32_10((uint32_t)evn, (uint32_t)(evn >> 32), (uint32_t)sys, 0x47494C31u, M.seed_lo, M.seed_hi, x);
                philox4x32_10((uint32_t)evn, (uint32_t)(evn >> 32), (uint32_t)sys, 0x47494C32u, M.seed_lo, M.seed_hi, y);
                u0 = ((double)(x[0] >> 5) * 67108864.0 + (double)(x[1] >> 6)) * 0x1.0p-53;
                u1 = ((double)(x[2] >> 5) * 67108864.0 + (double)(x[3] >> 6)) * 0x1.0p-53;
                u2 = ((double)(y[0] >> 5) * 67108864.0 + (double)(y[1] >> 6)) * 0x1.0p-53;
                u3 = ((double)(y[2] >> 5) * 67108864.0 + (double)(y[3] >> 6)) * 0x1.0p-53;
            }
            draws[4 * t] = -log1p(-u0); draws[4 * t + 1] = u1; draws[4 * t + 2] = u2; draws[4 * t + 3] = u3;
        }
        if (t == 0) ctl[0] = NT;                               // first thread whose running sum exceeds the target
        __syncthreads();
        const double R = tinc[NT - 1];
        if (!(R > 0.0)) { tnow = INFINITY; break; }           // ref :355: tau = inf ends the loop
        const double *dr = draws + 4 * (int)(n_ev - ev_base);
        const double u[4] = {0.0, dr[1], dr[2], dr[3]};
        const double tau = (1.0 / R) * dr[0];
        const double target = u[1] * R;
        int tsel;
        if (NT == 64) {                                        // one wavefront: the choice is a ballot
            const unsigned long long above = __ballot(inc > target && mine > 0.0), any = __ballot(mine > 0.0);
            tsel = above ? __builtin_ctzll(above) : 63 - __builtin_clzll(any);   // target rounded past the total: last lane with any rate
        } else {
            if (inc > target && mine > 0.0) atomicMin(&ctl[0], t);
            __syncthreads();
            tsel = ctl[0];
            if (tsel >= NT) {                                  // target rounded past the total: last thread that has any rate
                if (t == 0) ctl[1] = -1;
                __syncthreads();
                if (mine > 0.0) atomicMax(&ctl[1], t);
                __syncthreads();
                tsel = ctl[1];
            }
        }
        GSTAMP(1)
        // ---- C: the chosen thread picks the particle and applies the event (ref :363-446)
        if (t == tsel) {
            double run = tinc[t] - mine;
            int isel = -1;
            for (int i = c0; i < c1; ++i) {
                if (rate[i] > 0.0) { isel = i; run += rate[i]; if (run > target) break; }
            }
            const int i = isel, p = pos[i];
            uint8_t f = flg[i];
            const bool plus = (f & F_PLUS) != 0;
            double w, s;
            if (M.field_mode) { w = W[p]; s = S[p]; } else { w = (double)gsum_n; s = (double)gsum_s; }
            double mloc = 0.0;
            if (w > 0.0) { mloc = s / w; mloc = mloc > 1.0 ? 1.0 : (mloc < -1.0 ? -1.0 : mloc); }
            int l = p - 1, rr = p + 1;
            if (M.periodic) { l = l < 0 ? l + L : l; rr = rr >= L ? rr - L : rr; }
            const Channels c = channels(M, a.anchor ? a.anchor[p] != 0 : false, p, plus ? 1 : -1, (f & F_BOUND) != 0, mloc, beta,
                                        occ[p], l >= 0 ? occ[l] : 0, rr < L ? occ[rr] : 0);
            const double v = u[2] * c.total;
            const double e_diff = c.diff, e_act = e_diff + c.act, e_bind = e_act + c.bind, e_unbind = e_bind + c.unbind,
                         e_exit = e_unbind + c.leave;
            int kind = 0, to = p;                              // 0 nothing, 1 hop, 2 flip, 3 exit
            if (v < e_diff) {
                if (c.left + c.right > 0.0) { kind = 1; to = (u[3] < c.left / (c.left + c.right)) ? p - 1 : p + 1; }
            } else if (v < e_act) { kind = 1; to = p + 1; }
            else if (v < e_bind) f |= F_BOUND;
            else if (v < e_unbind) f &= (uint8_t)~F_BOUND;
            else if (v < e_exit) kind = 3;
            else kind = 2;
            if (kind == 1) {
                if (M.periodic) to = to < 0 ? to + L : (to >= L ? to - L : to);
                else to = to < 0 ? 0 : (to > L - 1 ? L - 1 : to);
                occ[p]--; occ[to]++;
                if (plus) { occp[p]--; occp[to]++; }
                pos[i] = to;
                if (to == p) kind = 0;                         // clipped at a wall: nothing moved
            } else if (kind == 2) {
                f ^= F_PLUS;
                if (plus) occp[p]--; else occp[p]++;
            } else if (kind == 3) {
                f &= (uint8_t)~F_ALIVE;
                rate[i] = 0.0;
                occ[p]--; if (plus) occp[p]--;
                if (a.exits && n_exit < ncap) {
                    double *row = a.exits + ((size_t)sys * ncap + n_exit) * 3;
                    row[0] = tnow; row[1] = (double)p; row[2] = (double)i;
                }
            }
            flg[i] = f;
            ctl[2] = kind; ctl[3] = p; ctl[4] = to; ctl[5] = plus ? 1 : -1;
        }
        __syncthreads();
        GSTAMP(2)
        // ---- D: the event's change of the smoothed histograms
        const int kind = ctl[2], p_old = ctl[3], p_new = ctl[4], sg = ctl[5];
        if (kind == 3) n_exit += 1;
        // whose rates must be re-evaluated before the next event: everybody when the global mean moved, otherwise the
        // particles within the table's reach (+1 site for the occupancy of neighbours) of the event's sites
        dirty_a = p_old; dirty_b = p_new;
        dirty_all = !M.field_mode && (kind == 2 || kind == 3);
        if (!M.field_mode) {
            if (kind == 2) gsum_s -= 2 * sg;
            else if (kind == 3) { gsum_s -= sg; gsum_n -= 1; }
        } else if (kind != 0) {
            const int Rt = a.tlen - 1;
            const int centre = kind == 1 ? min(p_old, p_new) : p_old, span = kind == 1 ? 1 : 0;
            const bool wrap1 = kind == 1 && M.periodic && (p_old - p_new > 1 || p_new - p_old > 1);   // hop across the seam
            int lo = centre - Rt, len = 2 * Rt + 1 + span;
            if (wrap1 || len >= L || (!M.periodic && Rt >= L)) { lo = 0; len = L; }
            for (int k = t; k < len; k += NT) {
                int x = lo + k;
                if (M.periodic) { x %= L; if (x < 0) x += L; }
                else if (x < 0 || x >= L) continue;
                const double g0 = site_weight(M, tab, a.tlen, x, p_old);
                if (kind == 1) {
                    const double g1 = site_weight(M, tab, a.tlen, x, p_new), d = g1 - g0;      // exact on the weight grid
                    W[x] += d; S[x] += sg > 0 ? d : -d;
                } else if (kind == 2) {
                    S[x] -= sg > 0 ? 2.0 * g0 : -2.0 * g0;
                } else {
                    W[x] -= g0; S[x] -= sg > 0 ? g0 : -g0;
                }
            }
        }
        __syncthreads();
        GSTAMP(3)
        // ---- E: time and observations (ref :514-538)
        n_ev += 1;
        tnow += tau;
        if (tnow > a.p.T) break;
        while (k_obs < nobs && t_next <= tnow) { record(k_obs); ++k_obs; t_next = k_obs < nobs ? a.times[k_obs] : INFINITY; }
        GSTAMP(4)
    }
#ifdef APS_STAMPS
    if (t == 0 && sys == 0 && a.exits) for (int k = 0; k < 5; ++k) a.exits[k] = (double)st[k];   // diagnostic build only
#endif
    if (t == 0) {
        if (a.n_recorded) a.n_recorded[sys] = k_obs;
        if (a.n_events) a.n_events[sys] = n_ev;
        if (a.t_final) a.t_final[sys] = tnow;
        if (a.n_exits) a.n_exits[sys] = n_exit;
    }
}

struct Dev {
    std::vector<void *> ptrs;
    ~Dev() { for (void *q : ptrs) (void)hipFree(q); }
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

const char *gil_last_error(void) { return g_gil_err.c_str(); }

int gil_run_batch(const gil_params *p, const int32_t *n0, const int32_t *pos0, const int8_t *sigma0, const uint8_t *bound0,
                  const double *uniforms, int32_t *pos_obs, int8_t *sigma_obs, uint8_t *flags_obs, int64_t *scalars_obs,
                  int32_t *n_recorded, int64_t *n_events, double *t_final, double *exits, int32_t *n_exits, double *kernel_ms) {
    auto bad = [&](const char *m) { g_gil_err = std::string("gil_run_batch: ") + m; return GIL_ERR_ARG; };
    if (!p || !n0 || !pos0 || !sigma0 || !p->beta || !p->times_obs) return bad("null argument");
    if (p->L < 2 || p->L > GIL_MAX_L) return bad("L must be in [2, GIL_MAX_L]");
    if (p->K < 1 || p->K > 32) return bad("site capacity K must be in [1, 32]");
    if (p->n_systems < 1 || p->n_cap < 1 || p->n_cap > GIL_MAX_N || p->n_obs < 1 || p->max_events < 0) return bad("bad n_systems / n_cap / n_obs / max_events");
    const int S = p->n_systems, L = p->L, ncap = p->n_cap;
    for (int s = 0; s < S; ++s) {
        if (n0[s] < 0 || n0[s] > ncap) return bad("n0 outside [0, n_cap]");
        std::vector<int> occ((size_t)L, 0);
        for (int i = 0; i < n0[s]; ++i) {
            const int x = pos0[(size_t)s * ncap + i];
            if (x < 0 || x >= L) return bad("position outside [0, L)");
            if (++occ[(size_t)x] > p->K) return bad("site capacity exceeded");
            const int sg = sigma0[(size_t)s * ncap + i];
            if (sg != 1 && sg != -1) return bad("sigma must be +1 or -1");
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_gil_err = "gil_run_batch: no HIP device"; return GIL_ERR_NODEVICE; }
    if (p->device < 0 || p->device >= ndev) return bad("device ordinal out of range");
    if (hipSetDevice(p->device) != hipSuccess) { g_gil_err = "hipSetDevice failed"; return GIL_ERR_HIP; }

    std::vector<double> table; int tlen = 0, q = 0;
    weight_table(p->sigma_grid, L, p->K, p->periodic != 0, table, tlen, q);
    GilArgs a{};
    const int NT = ncap <= 1024 ? 64 : 256;                   // one wavefront per system while a lane owns at most 16 particles
    a.p = *p; a.tlen = tlen; a.chunk = (ncap + NT - 1) / NT;
    Model &M = a.m;
    M.L = L; M.K = p->K; M.periodic = p->periodic ? 1 : 0; M.field_mode = p->sigma_grid > 0.0 ? 1 : 0;
    M.minus_anchor = p->minus_anchor ? 1 : 0; M.immobilize = p->immobilize ? 1 : 0; M.suppress_flip = p->suppress_flip ? 1 : 0;
    M.crowding = p->crowding ? 1 : 0; M.rate_diffusion = p->rate_diffusion; M.rate_active = p->rate_active;
    M.k_on = p->k_on; M.k_off = p->k_off; M.k_exit = p->k_exit; M.dt = 0.0;
    M.seed_lo = (uint32_t)p->seed; M.seed_hi = (uint32_t)(p->seed >> 32); M.ens_base = 0;
    Dev d;
    const size_t SN = (size_t)S * ncap, SO = (size_t)S * p->n_obs;
#define UPL(dst, src, n) do { a.dst = d.upload(src, n); if (!a.dst) { g_gil_err = "gil_run_batch: device upload failed (" #dst ")"; return GIL_ERR_HIP; } } while (0)
#define OUTB(dst, host, n) do { if (host) { a.dst = d.alloc<std::remove_pointer<decltype(a.dst)>::type>(n); if (!a.dst) { g_gil_err = "gil_run_batch: device allocation failed (" #dst ")"; return GIL_ERR_HIP; } } } while (0)
    UPL(beta, p->beta, (size_t)S); UPL(table, table.data(), table.size()); UPL(times, p->times_obs, (size_t)p->n_obs);
    UPL(n0, n0, (size_t)S); UPL(pos0, pos0, SN); UPL(sigma0, sigma0, SN);
    if (bound0) UPL(bound0, bound0, SN);
    if (p->anchor_mask) UPL(anchor, p->anchor_mask, (size_t)L);
    if (p->front_lo) UPL(front_lo, p->front_lo, (size_t)L);
    if (p->block_table) UPL(block_table, p->block_table, (size_t)(p->K + 1) * (p->K + 1));
    M.flip_n = 0; M.flip_tab = nullptr;
    if (p->flip_table) {                                       // a caller's flip_rate_fn, tabulated (aps_set_flip_table's layout)
        if (p->flip_n < 1 || p->flip_n > (1 << 24)) return bad("flip_n must be in [1, 2^24]");
        M.flip_tab = d.upload(p->flip_table, (size_t)2 * ((size_t)p->flip_n + 1));
        if (!M.flip_tab) { g_gil_err = "gil_run_batch: device upload failed (flip_table)"; return GIL_ERR_HIP; }
        M.flip_n = p->flip_n;
    }
    if (uniforms) UPL(uniforms, uniforms, (size_t)S * p->max_events * 4);
    OUTB(pos_obs, pos_obs, SO * ncap); OUTB(sigma_obs, sigma_obs, SO * ncap); OUTB(flags_obs, flags_obs, SO * ncap);
    if (scalars_obs) { a.scalars = d.alloc<long long>(SO * GIL_NSCALARS); if (!a.scalars) { g_gil_err = "gil_run_batch: device allocation failed (scalars)"; return GIL_ERR_HIP; } }
    OUTB(n_recorded, n_recorded, (size_t)S); OUTB(t_final, t_final, (size_t)S); OUTB(exits, exits, SN * 3); OUTB(n_exits, n_exits, (size_t)S);
    if (n_events) { a.n_events = d.alloc<long long>((size_t)S); if (!a.n_events) { g_gil_err = "gil_run_batch: device allocation failed (n_events)"; return GIL_ERR_HIP; } }
#undef UPL
#undef OUTB
    const size_t lds = ((size_t)2 * L + ((tlen + 2) & ~1) + ncap + (ncap & 1) + 8 + 5 * NT + 8) * sizeof(double) +
                       ((size_t)3 * ncap + 16) * sizeof(int) + (size_t)((ncap + 15) & ~15) + (size_t)2 * ((L + 15) & ~15);
    if (lds > 160 * 1024) return bad("system does not fit the 160 KB of LDS");
    if (lds > 48 * 1024 && hipFuncSetAttribute(NT == 64 ? reinterpret_cast<const void *>(&gil_kernel<64>) : reinterpret_cast<const void *>(&gil_kernel<256>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        g_gil_err = "gil_run_batch: cannot raise the dynamic LDS limit"; return GIL_ERR_HIP;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { g_gil_err = "hipEventCreate failed"; return GIL_ERR_HIP; }
    (void)hipEventRecord(e0, nullptr);
    if (NT == 64) hipLaunchKernelGGL(gil_kernel<64>, dim3((unsigned)S), dim3(64), lds, nullptr, a);
    else hipLaunchKernelGGL(gil_kernel<256>, dim3((unsigned)S), dim3(256), lds, nullptr, a);
    (void)hipEventRecord(e1, nullptr);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipDeviceSynchronize();
    float ms = 0.f;
    if (err == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) { g_gil_err = std::string("gil_kernel: ") + hipGetErrorString(err); return GIL_ERR_HIP; }
    if (kernel_ms) *kernel_ms = ms;
#define DOWNL(host, dev, bytes) do { if (host && hipMemcpy(host, a.dev, (bytes), hipMemcpyDeviceToHost) != hipSuccess) { g_gil_err = "gil_run_batch: download failed (" #dev ")"; return GIL_ERR_HIP; } } while (0)
    DOWNL(pos_obs, pos_obs, SO * ncap * 4); DOWNL(sigma_obs, sigma_obs, SO * ncap); DOWNL(flags_obs, flags_obs, SO * ncap);
    DOWNL(scalars_obs, scalars, SO * GIL_NSCALARS * 8); DOWNL(n_recorded, n_recorded, (size_t)S * 4); DOWNL(n_events, n_events, (size_t)S * 8);
    DOWNL(t_final, t_final, (size_t)S * 8); DOWNL(exits, exits, SN * 3 * 8); DOWNL(n_exits, n_exits, (size_t)S * 4);
#undef DOWNL
    return GIL_OK;
}

}  // extern "C"
