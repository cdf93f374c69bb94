// ntt_conv.hpp -- the field update as ONE exact convolution per step (included by aps_hip.hip inside its
// anonymous namespace).
//
// Hot path replaced: compute_local_m_field (PARTICLE_solver_CLASS.py:216-246), whose smoothing the reference itself evaluates by
// FFT on a torus (:223-227) and by scipy's gaussian_filter1d between walls (:229-238).  Here the smoothed histograms W, S are
// state (tile_step.hpp); a step changes them by  dW = cW (*) w,  dS = cS (*) w  with cW, cS the step's deposits as sparse integer
// signals on the lattice (a hop: -1 at the old site, +1 at the new one; a flip: -+2 in cS) and w the weight table.  For a table
// far beyond LDS (BASELINE config 5: 80 001 taps) tile_step's sweep costs deposits x taps = 1.7e9 LDS gathers per step; the
// convolution theorem costs O(M log M) with M = 2^21 -- and a number-theoretic transform keeps it EXACT: every W, S, dW, dS is
// an integer (units of 2^-q) of magnitude below 2^29 (32-bit field), the prime P = 15 * 2^27 + 1 = 2 013 265 921 exceeds twice
// that, so the residues mod P determine the integers.  Same bits as the sweep, the oracle and every other formulation.
// The binary64 field (integers below 2^51 in units of its own 2^-q) takes the same transform modulo TWO primes -- the second
// one 27 * 2^26 + 1 = 1 811 939 329, product 2^61.7 -- and the last sweep puts the two residues together (Chinese remainder:
// x = r0 + P0 ((r1 - r0) P0^-1 mod P1), centred), exactly.
//
// Reflecting walls: the deposits within the table's reach of a wall are entered a second time at their mirror site
// (-1 - p, 2L - 1 - p), i.e. the signals live on [-Rt, L + Rt) and a plain linear convolution gives the reference's
// mode='reflect' sums on [0, L); M >= L + 2 Rt makes the circular convolution equal to the linear one there.
// Torus: the second entry is the same site one period on (p + L, p - L); the tap at distance L / 2 of an even ring counts once.
//
// Transform: M = R2 R1 R0 (R0 = 128 along memory, R1, R2 <= 128), index i = i0 + R0 i1 + R0 R1 i2, frequency
// k = k2 + R2 k1 + R2 R1 k0:  w^(ik) = w_R2^(i2 k2) . w^(R0 i1 k2) . w_R1^(i1 k1) . w^(i0 (k2 + R2 k1)) . w_R0^(i0 k0)
// -- three sweeps of small transforms held in LDS (one launch each: along i2, along i1, along i0) with two twiddle
// multiplications in between; in place, slot i_a ends up holding k_a.  The last forward sweep, the product with the table's
// spectrum and the first inverse sweep touch the same 128 contiguous words and are one kernel: five launches per convolution,
// both signals in every launch -- and from two 128 x 128 slabs on (m >= 15) everything between the two i2 sweeps is ONE launch
// (ntt_mid: a slab per workgroup, in LDS): three.  Arithmetic: residues as uint32 in memory, as doubles in registers and LDS,
// products by binary64 fma (exact: the error term of a * b and the quotient estimate both come from fma; full rate on CDNA,
// where 32-bit integer multiplies run at a quarter).
#pragma once

constexpr uint32_t NTT_PRIMES[2] = {2013265921u, 1811939329u};   // 15 * 2^27 + 1, 27 * 2^26 + 1
constexpr uint32_t NTT_ROOTS[2] = {31u, 13u};                    // a primitive root of each
constexpr uint32_t NTT_P = NTT_PRIMES[0];
constexpr int NTT_TILE = 4096;                   // words of one signal a workgroup holds in LDS
constexpr int NTT_THREADS = 256;

inline uint32_t ntt_mulmod_u64(uint32_t a, uint32_t b, uint32_t P) { return (uint32_t)(((unsigned long long)a * b) % P); }
inline uint32_t ntt_powmod(uint32_t b, unsigned long long e, uint32_t P) {
    uint32_t r = 1u;
    while (e) { if (e & 1ull) r = ntt_mulmod_u64(r, b, P); b = ntt_mulmod_u64(b, b, P); e >>= 1; }
    return r;
}

// Residues travel LAZILY: integers in [0, 2 P) (2 P < 2^32, so a uint32 holds them) as uint32 in global memory; in registers and
// LDS doubles of EITHER sign whose magnitude the butterflies let grow (a level doubles it at most: below 2^8 * 2 P < 2^39 after
// the seven levels of a 128-point transform) until the next product brings them back to [0, 2 P).  Quotients are estimated by
// floor(x / P - 2^-10): never above the true quotient (floor: also for negative x), at most one below it for every product that
// occurs here (|a| < 2^39, b < 2^32, P > 2^30.7: quotient below 2^40.3, three roundings of 2^-53 relative + the dropped low part
// l / P < 2^-13 -> error below 2^-11 < 2^-10), so a remainder lies in [0, 2 P) without any correction step.
constexpr double NTT_BIAS = 0.0009765625;                        // 2^-10
struct NttMod { double P, Pinv; };                               // the prime as a double and its reciprocal (rounded)
// a * b mod P, lazy: |a| < 2^39, 0 <= b < 2^32 -> [0, 2 P).  Exact: a b = h + l with l from fma; h - q P is a small integer (fma again).
__device__ __forceinline__ double ntt_mul(const double a, const double b, const NttMod md) {
    const double h = a * b, l = fma(a, b, -h);
    const double q = floor(fma(h, md.Pinv, -NTT_BIAS));
    return fma(-q, md.P, h) + l;
}
__device__ __forceinline__ double ntt_red(const double v, const NttMod md) {   // v an integer, |v| < 2^40 -> [0, 2 P), congruent
    const double q = floor(fma(v, md.Pinv, -NTT_BIAS));
    return fma(-q, md.P, v);
}
__device__ __forceinline__ uint32_t ntt_canon(const uint32_t v, const uint32_t P) { return v >= P ? v - P : v; }     // [0, 2 P) -> [0, P)

struct NttPrime {                          // the tables of one prime (device pointers)
    uint32_t P;
    NttMod md;
    const uint32_t *wr;                    // [2][3][64]  w_R^j for the three axes (R0, R1, R2), forward / inverse
    const uint32_t *t1;                    // [2][R1 R2]  w^(+-R0 e)            (twiddle between the i2 and the i1 sweep, e = i1 k2)
    const uint32_t *t2hi, *t2lo;           // [2][M / 1024], [2][1024]: w^(+-e) = hi[e >> 10] lo[e & 1023]   (e = i0 (k2 + R2 k1))
    const uint32_t *what;                  // [M] spectrum of the table in the transform's own output order, times 1 / M
    const uint32_t *whatp;                 // [M] the same in ntt_mid's slot order: [k2][i0-slot][i1-slot], slot s holds frequency brev(s)
};
struct NttPlan {
    int m, a0, a1, a2;                     // M = 2^m = R2 R1 R0, R_x = 2^a_x (a0 = 7; a2 = 0: two sweeps only)
    int L, Rt;                             // lattice sites, table reach; signal index = site + Rt
    int E, np;                             // ensembles; primes: 1 (32-bit field: int32 {W, S}) or 2 (binary64 field: double {W, S} = integers * unit)
    uint32_t crt_inv;                      // P0^-1 mod P1
    double unit;                           // 2^-q of the binary64 field
    NttPrime pr[2];                        // signals: [prime][ensemble][W | S][M] words
};

// B butterfly levels on the 2^B values of one thread.  The values are rows  n + (t << lo_shift), t = 0 .. 2^B - 1, of a transform
// of size 2^A whose level `s0 + s` they carry out (decimation in frequency: pair distance 2^(A - 1 - s0 - s) rows); the twiddle of
// the pair whose upper row is i is w_R^((i mod h) << level), taken from wtab (w_R^j, j < 64, as doubles).
// N0: n is known to be zero (the second pass): the pairs with j = 0 need no product; otherwise every pair multiplies (w_R^0 = 1
// for the rare n = j = 0: a branch per butterfly would cost more than the product)
template <int B, bool N0>
__device__ __forceinline__ void ntt_reg_levels(double (&x)[1 << B], const double *__restrict__ wtab, const int n, const int lo_shift, const int s0, const NttMod md) {
    // a level at most doubles the magnitude of its values (sums and differences; a product is back in [0, 2 P))
#pragma unroll
    for (int s = 0; s < B; ++s) {
        constexpr int NV = 1 << B;
        const int ht = NV >> (s + 1);
#pragma unroll
        for (int pr = 0; pr < NV / 2; ++pr) {
            const int j = pr & (ht - 1), u = ((pr - j) << 1) + j, v = u + ht;
            const double xa = x[u], xb = x[v];
            x[u] = xa + xb;
            const double d = xa - xb;                                      // either sign: the quotient estimate of ntt_mul floors
            const int e = (n + (j << lo_shift)) << (s0 + s);               // < 64
            if (N0 && j == 0) x[v] = d; else x[v] = ntt_mul(d, wtab[e], md);
        }
    }
}

// Transform of size 2^A (natural order in, bit-reversed order out) of the NC = 2^lg_nc columns of one signal held in LDS, element
// (row r, column c) at buf[r * ld + c] (doubles, lazy residues): two register passes (2^AH = 16 rows a thread, then 2^AL = 8).
template <int A>
__device__ __forceinline__ void ntt_lds_transform(double *buf, const int lg_nc, const int ld, const double *__restrict__ wtab, const int t, const NttMod md) {
    constexpr int AH = A < 4 ? A : 4, AL = A - AH;
    const int NC = 1 << lg_nc;
    for (int w = t; w < (NC << AL); w += NTT_THREADS) {                    // pass 1: rows n + (tt << AL)
        const int c = w & (NC - 1), n = w >> lg_nc;
        double x[1 << AH];
#pragma unroll
        for (int tt = 0; tt < (1 << AH); ++tt) x[tt] = buf[(n + (tt << AL)) * ld + c];
        ntt_reg_levels<AH, false>(x, wtab, n, AL, 0, md);
#pragma unroll
        for (int tt = 0; tt < (1 << AH); ++tt) buf[(n + (tt << AL)) * ld + c] = AL > 0 ? x[tt] : ntt_red(x[tt], md);   // (reduced by the last pass)
    }
    __syncthreads();
    if constexpr (AL > 0) {
        for (int w = t; w < (NC << AH); w += NTT_THREADS) {                // pass 2: rows (u << AL) + v
            const int c = w & (NC - 1), u = w >> lg_nc;
            double x[1 << AL];
#pragma unroll
            for (int v = 0; v < (1 << AL); ++v) x[v] = buf[((u << AL) + v) * ld + c];
            ntt_reg_levels<AL, true>(x, wtab, 0, 0, AH, md);
#pragma unroll
            for (int v = 0; v < (1 << AL); ++v) buf[((u << AL) + v) * ld + c] = ntt_red(x[v], md);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ int ntt_bitrev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// ---- sweep along a strided axis (i2: stride R0 R1, or i1: stride R0): a workgroup takes NC consecutive words (same other digits)
// for all R = 2^A values of the axis digit, of one signal (blockIdx.y).
// first forward sweep (INIT): the input are the deposit signals -- int32 coefficients, cleared behind the read -- instead of
// residues; last inverse sweep (FINAL): the result is added to {W, S} of the sites.  Both take every prime of the plan in turn
// (blockIdx.z = ensemble): the coefficients are read once, and the last sweep needs all residues of a word to put the integer
// together; in between a launch covers the primes by blockIdx.z = prime * E + ensemble.
// AXIS 1: forward: multiply by w^(R0 i1 k2), transform over i1; inverse: transform, multiply by w^(-R0 i1 k2).
// NP: primes of the plan (2 is only instantiated for the i2 sweeps: the binary64 field wants m >= 15)
template <int AXIS, bool INV, int A, int NP>
__global__ __launch_bounds__(NTT_THREADS) void ntt_strided(const NttPlan pl, uint32_t *__restrict__ data, int *__restrict__ csig, void *__restrict__ ws, const int init_or_final) {
    __shared__ double buf[NTT_TILE + 128];
    __shared__ double wtab[64];
    const int t = threadIdx.x, sgl = blockIdx.y;                 // one signal (0: W, 1: S) per workgroup
    constexpr int R = 1 << A;
    constexpr int lg_nc = 12 - A, NC = 1 << lg_nc;               // consecutive words per axis value (>= 32)
    static_assert(NTT_TILE == 4096, "lg_nc assumes 4096 words per tile");
    const size_t M = (size_t)1 << pl.m;
    const size_t stride = AXIS == 2 ? ((size_t)1 << (pl.a0 + pl.a1)) : ((size_t)1 << pl.a0);
    const size_t per_outer = stride / NC;                        // tiles per value of the digits above the axis
    const size_t outer = blockIdx.x / per_outer, inner0 = (blockIdx.x % per_outer) * NC;
    const size_t base = outer * stride * R + inner0;             // word index of (axis digit 0, first column)
    const bool first = !INV && init_or_final, last = INV && init_or_final;
    const int ens = (first || last) ? (int)blockIdx.z : (int)blockIdx.z % pl.E;
    const int npr = (first || last) ? NP : 1;
    constexpr int ld = NC + 1;
    const int k2 = (int)outer;                                   // AXIS 1: the digit above (slot i2 holds k2)
    constexpr int NI = NTT_TILE / NTT_THREADS;                   // words per thread: all their loads are issued before the first is used
    int coef[NP > 1 ? NI : 1];                                   // INIT with two primes: the deposit coefficients, read once for both
    uint32_t res0[NP > 1 ? NI : 1];                              // FINAL with two primes: the residues mod the first one
    if (NP > 1 && first) {
        int *c0 = csig + ((size_t)ens * 2 + sgl) * M;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int w = t + u * NTT_THREADS, c = w & (NC - 1), r = w >> lg_nc;
            coef[u] = c0[base + (size_t)r * stride + c];
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {                           // cleared for the next step
            const int w = t + u * NTT_THREADS, c = w & (NC - 1), r = w >> lg_nc;
            if (coef[u]) c0[base + (size_t)r * stride + c] = 0;
        }
    }
#pragma unroll 1
    for (int q = 0; q < npr; ++q) {
        const int pi = (first || last) ? q : (int)blockIdx.z / pl.E;
        const NttPrime pp = pi ? pl.pr[1] : pl.pr[0];            // (a select of scalars: indexing the argument would put the array into registers)
        const NttMod md = pp.md;
        uint32_t *sig0 = data + (((size_t)pi * pl.E + ens) * 2 + sgl) * M;
        if (q) __syncthreads();                                  // everyone has read the previous prime's result out of LDS
        int tl = t;                                              // opaque per iteration: hoisted out of the loop, the 16 word addresses (and as many
        if (NP > 1) asm volatile("" : "+v"(tl));                 // LDS offsets) of a thread would stay in registers across both transforms
        if (t < 64) wtab[t] = (double)pp.wr[(INV ? 192 : 0) + (AXIS == 2 ? 128 : 64) + t];
        // ---- load (row r = axis digit, column c): + twiddle of the i1 sweep going forward
        {
            uint32_t raw[NI], twv[NI];
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int w = tl + u * NTT_THREADS, c = w & (NC - 1), r = w >> lg_nc;
                const size_t g = base + (size_t)r * stride + c;
                raw[u] = first ? (NP > 1 ? 0u : (uint32_t)csig[((size_t)ens * 2 + sgl) * M + g]) : sig0[g];
                twv[u] = (AXIS == 1 && !INV) ? pp.t1[(size_t)r * k2] : 1u;       // w^(R0 i1 k2), i1 = r
            }
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int w = tl + u * NTT_THREADS, c = w & (NC - 1), r = w >> lg_nc;
                double v0;
                if (first) {                                     // deposit coefficients: small signed integers
                    const int x0 = NP > 1 ? coef[u] : (int)raw[u];
                    if (NP == 1 && x0) csig[((size_t)ens * 2 + sgl) * M + base + (size_t)r * stride + c] = 0;   // cleared for the next step
                    v0 = (double)x0;
                } else v0 = (double)raw[u];
                if (AXIS == 1 && !INV) v0 = ntt_mul(v0, (double)twv[u], md);
                buf[r * ld + c] = v0;
            }
        }
        __syncthreads();
        ntt_lds_transform<A>(buf, lg_nc, ld, wtab, tl, md);
        // ---- store, un-permuting the bit-reversed output: slot r gets the value of frequency r
        uint32_t twv[NI];
        int old_w[NP == 1 ? NI : 1];                             // FINAL, int32 field: the sites' W (or S) this thread will add to (binary64: read where it is added)
        const bool combine = last && q == npr - 1;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int w = tl + u * NTT_THREADS, c = w & (NC - 1), r = w >> lg_nc;
            twv[u] = (AXIS == 1 && INV) ? pp.t1[((size_t)1 << (pl.a1 + pl.a2)) + (size_t)r * k2] : 1u;   // w^(-R0 i1 k2): the result index r is i1
            if constexpr (NP == 1) {
                old_w[u] = 0;
                if (combine) {
                    const long long site = (long long)(base + (size_t)r * stride + c) - pl.Rt;
                    if (site >= 0 && site < pl.L) old_w[u] = reinterpret_cast<const int *>(reinterpret_cast<const int2 *>(ws) + (size_t)ens * pl.L + site)[sgl];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int w = tl + u * NTT_THREADS, c = w & (NC - 1), r = w >> lg_nc;
            double d0 = buf[ntt_bitrev(r, A) * ld + c];
            const size_t g = base + (size_t)r * stride + c;
            if (AXIS == 1 && INV) d0 = ntt_mul(d0, (double)twv[u], md);
            const uint32_t v0 = (uint32_t)d0;
            if (!last) { sig0[g] = v0; continue; }
            // natural order again: word g is the change of W (or S) at site g - Rt
            const long long site = (long long)g - pl.Rt;
            const uint32_t vc = ntt_canon(v0, pp.P);
            if (!combine) { if constexpr (NP > 1) res0[u] = vc; continue; }
            if (site < 0 || site >= pl.L) continue;
            if constexpr (NP == 1) {
                if (vc) reinterpret_cast<int *>(reinterpret_cast<int2 *>(ws) + (size_t)ens * pl.L + site)[sgl] = old_w[u] + (vc > pp.P / 2 ? (int)(vc - pp.P) : (int)vc);
            } else {
                // Chinese remainder: x = r0 + P0 t, t = (r1 - r0) P0^-1 mod P1; centred: |x| < P0 P1 / 2 by construction of the weight grid
                const uint32_t P0 = pl.pr[0].P, P1 = pp.P, r0 = res0[u];
                const uint32_t r0m = r0 >= P1 ? r0 - P1 : r0;                     // P0 < 2 P1
                const uint32_t dd = vc >= r0m ? vc - r0m : vc + P1 - r0m;
                const uint32_t tt = ntt_canon((uint32_t)ntt_mul((double)dd, (double)pl.crt_inv, md), P1);
                const unsigned long long x = (unsigned long long)r0 + (unsigned long long)P0 * tt, PP = (unsigned long long)P0 * P1;
                const long long xs = x > PP / 2 ? (long long)(x - PP) : (long long)x;
                if (xs) reinterpret_cast<double *>(reinterpret_cast<double2 *>(ws) + (size_t)ens * pl.L + site)[sgl] += (double)xs * pl.unit;   // (exact: both on the grid 2^-q)
            }
        }
    }
}

// ---- the contiguous axis (i0, R0 = 128): forward sweep (with its twiddle w^(i0 (k2 + R2 k1))), product with the table's spectrum,
// inverse sweep (twiddle w^(-i0 ...)): one kernel, a workgroup takes 32 rows of 128 contiguous words of both signals.
// FWD_ONLY: stop after the forward sweep (building the table's spectrum).
template <bool FWD_ONLY>
__global__ __launch_bounds__(NTT_THREADS) void ntt_contig(const NttPlan pl, uint32_t *__restrict__ data) {
    __shared__ double buf[NTT_TILE + 128];
    __shared__ double wtab[2][64];
    const int t = threadIdx.x, sgl = blockIdx.y;
    constexpr int A0 = 7, R0 = 1 << A0, lg_nr = 12 - A0, NR = 1 << lg_nr;   // rows per tile
    const size_t M = (size_t)1 << pl.m;
    const size_t row0 = (size_t)blockIdx.x * NR;                 // row = (i1-slot, i2-slot) = k1 + R1 k2
    const NttPrime pp = (int)blockIdx.z >= pl.E ? pl.pr[1] : pl.pr[0];              // blockIdx.z = prime * E + ensemble: the signals are laid out in that order
    const NttMod md = pp.md;
    uint32_t *sig0 = data + ((size_t)blockIdx.z * 2 + sgl) * M;
    constexpr int ld = NR + 1;                                   // element (transform row i0, tile row r) at i0 * ld + r
    const int R1m = (1 << pl.a1) - 1;
    if (t < 128) wtab[t >> 6][t & 63] = (double)pp.wr[(t >> 6) * 192 + (t & 63)];
    constexpr int NI = NTT_TILE / NTT_THREADS;
    auto tw2_index = [&](const size_t row, const int i0, size_t &ihi, size_t &ilo) {
        const int k1 = (int)(row & (size_t)R1m), k2 = (int)(row >> pl.a1);
        const unsigned long long e = (unsigned long long)i0 * ((unsigned long long)k2 + ((unsigned long long)k1 << pl.a2));   // < M
        ihi = (size_t)(e >> 10); ilo = (size_t)(e & 1023ull);
    };
    uint32_t whatv[NI];                                          // the table's spectrum at this thread's words: asked for now, used after the forward sweep
    {
        uint32_t raw[NI], hi[NI], lo[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int w = t + u * NTT_THREADS, i0 = w & (R0 - 1), r = w >> A0;
            const size_t g = (row0 + r) * R0 + i0;
            size_t ihi, ilo;
            tw2_index(row0 + r, i0, ihi, ilo);
            raw[u] = sig0[g]; hi[u] = pp.t2hi[ihi]; lo[u] = pp.t2lo[ilo];
            whatv[u] = FWD_ONLY ? 0u : pp.what[g];
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int w = t + u * NTT_THREADS, i0 = w & (R0 - 1), r = w >> A0;
            buf[i0 * ld + r] = ntt_mul((double)raw[u], ntt_mul((double)hi[u], (double)lo[u], md), md);
        }
    }
    __syncthreads();
    ntt_lds_transform<A0>(buf, lg_nr, ld, wtab[0], t, md);
    if (FWD_ONLY) {
        for (int w = t; w < NTT_TILE; w += NTT_THREADS) {
            const int i0 = w & (R0 - 1), r = w >> A0;
            const int src = ntt_bitrev(i0, A0);
            sig0[(row0 + r) * R0 + i0] = ntt_canon((uint32_t)buf[src * ld + r], pp.P);      // (the table's spectrum is kept canonical)
        }
        return;
    }
    // product with the table's spectrum, un-permuting on the way (slot i0 <- frequency i0): through registers
    double keep0[NTT_TILE / NTT_THREADS];
#pragma unroll
    for (int u = 0; u < NTT_TILE / NTT_THREADS; ++u) {
        const int w = t + u * NTT_THREADS;
        const int i0 = w & (R0 - 1), r = w >> A0;
        const int src = ntt_bitrev(i0, A0);
        keep0[u] = ntt_mul(buf[src * ld + r], (double)whatv[u], md);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NTT_TILE / NTT_THREADS; ++u) {
        const int w = t + u * NTT_THREADS;
        const int i0 = w & (R0 - 1), r = w >> A0;
        buf[i0 * ld + r] = keep0[u];
    }
    __syncthreads();
    ntt_lds_transform<A0>(buf, lg_nr, ld, wtab[1], t, md);
    {
        uint32_t hi[NI], lo[NI];
        const size_t nhi = M >> 10;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int w = t + u * NTT_THREADS, i0 = w & (R0 - 1), r = w >> A0;
            size_t ihi, ilo;
            tw2_index(row0 + r, i0, ihi, ilo);
            hi[u] = pp.t2hi[nhi + ihi]; lo[u] = pp.t2lo[1024 + ilo];
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int w = t + u * NTT_THREADS, i0 = w & (R0 - 1), r = w >> A0;
            const double v0 = ntt_mul(buf[ntt_bitrev(i0, A0) * ld + r], ntt_mul((double)hi[u], (double)lo[u], md), md);
            sig0[(row0 + r) * R0 + i0] = (uint32_t)v0;
        }
    }
}

// ---- the three middle launches as ONE (a0 = a1 = 7, a2 >= 1): a workgroup of 1024 threads keeps the 128 x 128 words of one
// i2-slot of one signal in LDS (element (i1-slot s1, i0-slot s0) at buf[s1 * LD + s0], doubles) and runs  twiddle 1, sweep over
// i1, twiddle 2, sweep over i0, product with the table's spectrum, inverse sweep over i0, twiddle 2^-1, inverse sweep over i1,
// twiddle 1^-1  on it -- one read and one write of the signal instead of three of each.  No un-permuting anywhere: a forward
// sweep leaves frequency brev(s) in slot s; the inverse sweep takes its logical row r from slot brev(r), which puts its output
// (index brev(r) at logical row r) back in natural order; the spectrum is stored in the slots' order (whatp).
// Register hand-overs: the load is already the first pass of the first sweep (thread t holds words t + 1024 tt = rows n + 8 tt of
// column t & 127), the second pass of the forward i0 sweep of two neighbouring groups is the first pass of the inverse one (16
// adjacent slots), the last pass stores straight to memory: 7 LDS reads and 6 writes of the tile in all.
// No reduction anywhere: every run of seven butterfly levels ends in a product (twiddle or spectrum), which takes |a| < 2^39.
// Twiddles w^(+-e): along a thread's 16 rows the exponents are an arithmetic progression -- two table look-ups and 18 products.
constexpr int NTT_MID_THREADS = 1024;
constexpr int NTT_MID_LD = 129;
constexpr size_t NTT_MID_LDS = ((size_t)128 * NTT_MID_LD + 128) * sizeof(double);
constexpr int NTT_MID_WORDS = 128 * 128;

__device__ __forceinline__ double ntt_tw2(const NttPrime &pp, const int m, const bool inv, const unsigned e) {     // w^(+-e), e < M, lazy
    const size_t nhi = ((size_t)1 << m) >> 10;
    return ntt_mul((double)pp.t2hi[(inv ? nhi : 0) + (e >> 10)], (double)pp.t2lo[(inv ? 1024u : 0u) + (e & 1023u)], pp.md);
}
__device__ __forceinline__ void ntt_geom16(double (&tw)[16], const double base, const double ratio, const NttMod md) {   // base * ratio^i, depth 4
    double pw = ratio;
    tw[0] = base;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int i = 0; i < (1 << b); ++i) tw[i + (1 << b)] = ntt_mul(tw[i], pw, md);
        if (b < 3) pw = ntt_mul(pw, pw, md);
    }
}
__host__ __device__ constexpr int ntt_brev3(int v) { return ((v & 1) << 2) | (v & 2) | ((v >> 2) & 1); }
__host__ __device__ constexpr int ntt_brev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v >> 1) & 2) | ((v >> 3) & 1); }

__global__ __launch_bounds__(NTT_MID_THREADS) void ntt_mid(const NttPlan pl, uint32_t *__restrict__ data) {
    extern __shared__ double ntt_mid_lds[];
    constexpr int LD = NTT_MID_LD;
    double *const buf = ntt_mid_lds, *const wf = buf + 128 * LD, *const wi = wf + 64;
    const int t = threadIdx.x, c = t & 127;
    const int n = __builtin_amdgcn_readfirstlane(t >> 7);        // 0 .. 7, the same for the 64 lanes of a wave
    const int sgl = blockIdx.y, k2 = blockIdx.x;
    const size_t M = (size_t)1 << pl.m;
    const NttPrime pp = (int)blockIdx.z >= pl.E ? pl.pr[1] : pl.pr[0];              // blockIdx.z = prime * E + ensemble
    const NttMod md = pp.md;
    uint32_t *const slab = data + ((size_t)blockIdx.z * 2 + sgl) * M + (size_t)k2 * NTT_MID_WORDS;
    const uint32_t *const wslab = pp.whatp + (size_t)k2 * NTT_MID_WORDS;
    const uint32_t *const t1f = pp.t1, *const t1i = pp.t1 + ((size_t)1 << (pl.a1 + pl.a2));
    if (t < 128) (t < 64 ? wf : wi)[t & 63] = (double)pp.wr[(t >> 6) * 192 + (t & 63)];
    double x[16];
    {   // load (word t + 1024 tt = (i1 = n + 8 tt, i0 = c)), twiddle 1 = w^(R0 i1 k2)
        uint32_t raw[16];
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) raw[tt] = slab[t + tt * 1024];
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) x[tt] = ntt_mul((double)raw[tt], (double)t1f[(size_t)(n + 8 * tt) * k2], md);
    }
    uint32_t wv[16];                                            // the table's spectrum at this thread's slots of the product: asked for now
#pragma unroll
    for (int j = 0; j < 16; ++j) wv[j] = wslab[(16 * n + j) * 128 + c];
    __syncthreads();
    // forward over i1, pass 1 (rows n + 8 tt of column c) -- in the registers of the load
    ntt_reg_levels<4, false>(x, wf, n, 3, 0, md);
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) buf[(n + 8 * tt) * LD + c] = x[tt];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {                             // pass 2: rows 8 u + v
        const int u = n + 8 * it;
        double y[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) y[v] = buf[(8 * u + v) * LD + c];
        ntt_reg_levels<3, true>(y, wf, 0, 0, 4, md);
#pragma unroll
        for (int v = 0; v < 8; ++v) buf[(8 * u + v) * LD + c] = y[v];
    }
    __syncthreads();
    {   // forward over i0 of the i1-slot c (k1 = brev(c)), pass 1: i0 = n + 8 tt, twiddle 2 = w^(i0 q), q = k2 + R2 k1
        const unsigned q = (unsigned)k2 + ((unsigned)ntt_bitrev(c, 7) << pl.a2);
        double tw[16];
        ntt_geom16(tw, ntt_tw2(pp, pl.m, false, (unsigned)n * q), ntt_tw2(pp, pl.m, false, 8u * q), md);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) x[tt] = ntt_mul(buf[c * LD + n + 8 * tt], tw[tt], md);
        ntt_reg_levels<4, false>(x, wf, n, 3, 0, md);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) buf[c * LD + n + 8 * tt] = x[tt];
    }
    __syncthreads();
    {   // pass 2 of the groups u = 2 n, 2 n + 1 (slots 16 n .. 16 n + 15), the product, pass 1 of the inverse sweep (its n' = brev3(n))
        double y0[8], y1[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) { y0[v] = buf[c * LD + 16 * n + v]; y1[v] = buf[c * LD + 16 * n + 8 + v]; }
        ntt_reg_levels<3, true>(y0, wf, 0, 0, 4, md);
        ntt_reg_levels<3, true>(y1, wf, 0, 0, 4, md);
        double y[16];
#pragma unroll
        for (int v = 0; v < 8; ++v) { y[v] = ntt_mul(y0[v], (double)wv[v], md); y[8 + v] = ntt_mul(y1[v], (double)wv[8 + v], md); }
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) x[tt] = y[ntt_brev4(tt)];                 // logical row n' + 8 tt sits in slot 16 n + brev4(tt)
        ntt_reg_levels<4, false>(x, wi, ntt_brev3(n), 3, 0, md);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) buf[c * LD + 16 * n + ntt_brev4(tt)] = x[tt];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {                             // inverse over i0, pass 2: logical rows 8 u + v in slots 16 brev3(v) + brev4(u)
        const int u = n + 8 * it, ub = ntt_brev4(u);
        double y[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) y[v] = buf[c * LD + 16 * ntt_brev3(v) + ub];
        ntt_reg_levels<3, true>(y, wi, 0, 0, 4, md);
#pragma unroll
        for (int v = 0; v < 8; ++v) buf[c * LD + 16 * ntt_brev3(v) + ub] = y[v];
    }
    __syncthreads();
    {   // inverse over i1 at i0 = c, pass 1: logical rows k1 = n + 8 tt in slots 16 brev3(n) + brev4(tt); twiddle 2^-1 = w^(-c (k2 + R2 k1))
        const int nb = 16 * ntt_brev3(n);
        double tw[16];
        ntt_geom16(tw, ntt_tw2(pp, pl.m, true, (unsigned)c * ((unsigned)k2 + ((unsigned)n << pl.a2))), ntt_tw2(pp, pl.m, true, (unsigned)c << (3 + pl.a2)), md);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) x[tt] = ntt_mul(buf[(nb + ntt_brev4(tt)) * LD + c], tw[tt], md);
        ntt_reg_levels<4, false>(x, wi, n, 3, 0, md);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) buf[(nb + ntt_brev4(tt)) * LD + c] = x[tt];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {                             // pass 2, twiddle 1^-1 = w^(-R0 i1 k2), store: slot = i1 again
        const int u = n + 8 * it, ub = ntt_brev4(u);
        double y[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) y[v] = buf[(16 * ntt_brev3(v) + ub) * LD + c];
        ntt_reg_levels<3, true>(y, wi, 0, 0, 4, md);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int i1 = 16 * ntt_brev3(v) + ub;
            slab[i1 * 128 + c] = (uint32_t)ntt_mul(y[v], (double)t1i[(size_t)i1 * k2], md);
        }
    }
}

// launch of a strided sweep whose radix is only known at run time (the i2 sweep: 2^a2, a2 = 1 .. 7)
template <int AXIS, bool INV, int NP>
inline void ntt_launch_strided(int a, dim3 grid, dim3 block, hipStream_t stream, hipEvent_t e0, hipEvent_t e1, bool timed, const NttPlan &pl, uint32_t *data, int *csig, void *ws, int flag) {
#define NTT_CASE(AA) case AA: if (timed) hipExtLaunchKernelGGL((ntt_strided<AXIS, INV, AA, NP>), grid, block, 0, stream, e0, e1, 0, pl, data, csig, ws, flag); \
                              else hipLaunchKernelGGL((ntt_strided<AXIS, INV, AA, NP>), grid, block, 0, stream, pl, data, csig, ws, flag); break;
    switch (a) { NTT_CASE(1) NTT_CASE(2) NTT_CASE(3) NTT_CASE(4) NTT_CASE(5) NTT_CASE(6) NTT_CASE(7) default: break; }
#undef NTT_CASE
}

// ---- host side: tables of a plan for one prime (everything mod P by 64-bit integer arithmetic)
struct NttTables { std::vector<uint32_t> wr, t1, t2hi, t2lo; };
inline void ntt_split(int m, int &a0, int &a1, int &a2) { a0 = 7; a1 = std::min(7, m - 7); a2 = m - 7 - a1; }
inline void ntt_build_tables(int m, uint32_t P, uint32_t G, NttTables &T) {
    int a0, a1, a2;
    ntt_split(m, a0, a1, a2);
    const unsigned long long M = 1ull << m;
    const uint32_t w = ntt_powmod(G, (P - 1ull) / M, P), wi = ntt_powmod(w, P - 2ull, P);
    T.wr.assign(2 * 192, 1u);
    for (int inv = 0; inv < 2; ++inv) {
        const uint32_t ww = inv ? wi : w;
        const int as[3] = {a0, a1, a2};
        for (int ax = 0; ax < 3; ++ax) {
            const uint32_t wR = ntt_powmod(ww, M >> as[ax], P);       // w_R = w^(M / R)
            uint32_t cur = 1u;
            for (int j = 0; j < 64; ++j) { T.wr[(size_t)inv * 192 + ax * 64 + j] = cur; cur = ntt_mulmod_u64(cur, wR, P); }
        }
    }
    const size_t n1 = (size_t)1 << (a1 + a2);
    T.t1.assign(2 * n1, 1u);
    for (int inv = 0; inv < 2; ++inv) {
        const uint32_t step = ntt_powmod(inv ? wi : w, 1ull << a0, P);    // w^(R0)
        uint32_t cur = 1u;
        for (size_t e = 0; e < n1; ++e) { T.t1[inv * n1 + e] = cur; cur = ntt_mulmod_u64(cur, step, P); }
    }
    const size_t nhi = std::max<size_t>(M >> 10, 1);
    T.t2hi.assign(2 * nhi, 1u); T.t2lo.assign(2 * 1024, 1u);
    for (int inv = 0; inv < 2; ++inv) {
        const uint32_t ww = inv ? wi : w, step = ntt_powmod(ww, 1024ull, P);
        uint32_t cur = 1u;
        for (size_t e = 0; e < nhi; ++e) { T.t2hi[inv * nhi + e] = cur; cur = ntt_mulmod_u64(cur, step, P); }
        cur = 1u;
        for (size_t e = 0; e < 1024; ++e) { T.t2lo[inv * 1024 + e] = cur; cur = ntt_mulmod_u64(cur, ww, P); }
    }
}
