// aps_common.hpp -- pieces shared by the translation units of libaps_hip.so (synchronous stepper aps_hip.hip,
// exact event loop gillespie_hip.hip): the deterministic exp, Philox4x32-10, the rate table of one particle
// (step_gillespie's vector section, ref :261-351) and the exact-grid weight table.  Everything sits in an
// anonymous namespace: each translation unit gets its own copy.
#pragma once

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// deterministic exp: mul, fma, rint and an exponent insert only -> identical bits on host and device
__host__ __device__ inline double aps_exp(double x) {
    const double LOG2E = 0x1.71547652b82fep+0, LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
    x = x > 700.0 ? 700.0 : (x < -700.0 ? -700.0 : x);
    const double kf = rint(x * LOG2E);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 0x1.6124613a86d09p-33;                       // 1/13!
    p = fma(p, r, 0x1.1eed8eff8d898p-29);
    p = fma(p, r, 0x1.ae64567f544e4p-26);
    p = fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = fma(p, r, 0x1.71de3a556c734p-19);
    p = fma(p, r, 0x1.a01a01a01a01ap-16);
    p = fma(p, r, 0x1.a01a01a01a01ap-13);
    p = fma(p, r, 0x1.6c16c16c16c17p-10);
    p = fma(p, r, 0x1.1111111111111p-7);
    p = fma(p, r, 0x1.5555555555555p-5);
    p = fma(p, r, 0x1.5555555555555p-3);
    p = fma(p, r, 0x1.0000000000000p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const long long k = (long long)kf, k1 = k / 2, k2 = k - k1;
    union { unsigned long long u; double d; } s1, s2;
    s1.u = (unsigned long long)(k1 + 1023) << 52;
    s2.u = (unsigned long long)(k2 + 1023) << 52;
    return p * s1.d * s2.d;
}

// Philox4x32-10 (Salmon et al., Random123): counter-based, so a particle's draw depends only on
// (seed, step, particle index, ensemble) -- never on which thread, GPU or tiling evaluated it.
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                     uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;   // one v_mad_u64_u32 each
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Model {               // by-value kernel argument: everything the rate code needs
    int L, K, periodic, field_mode, minus_anchor, immobilize, suppress_flip, crowding;
    double rate_diffusion, rate_active, k_on, k_off, k_exit, dt;
    uint32_t seed_lo, seed_hi;
    int ens_base;
    // a caller's flip_rate_fn (ref :59-62, applied at :261-262), tabulated by the host: flip_tab[(spin > 0 ? 0 : flip_n + 1) + i]
    // = fn(spin, -1 + 2 i / flip_n), i = 0 .. flip_n; NULL: the Curie-Weiss rate exp(-beta spin m) of ref :60
    int flip_n;
    const double *flip_tab;
};

struct Channels { double diff, act, flip, bind, unbind, leave, left, right, total; };

// Rate table of one particle: the arithmetic of step_gillespie's vector section (ref :261-351),
// operation for operation (same association order as the NumPy expressions).
__device__ inline Channels channels(const Model &M, bool anchored_site, int p, int spin, bool bound, double mloc,
                                    double beta, int occ_self, int occ_left, int occ_right) {
    const int L = M.L, K = M.K;
    const bool plus = spin > 0;
    double flip;
    if (M.flip_tab) {                                        // linear interpolation of the caller's tabulated rate in m (m is clipped to [-1, 1])
        const double u = (mloc + 1.0) * (0.5 * (double)M.flip_n);
        int i = (int)u;
        i = i < 0 ? 0 : (i >= M.flip_n ? M.flip_n - 1 : i);
        const double f = u - (double)i;
        const double *tb = M.flip_tab + (plus ? 0 : M.flip_n + 1);
        const double a = tb[i], b = tb[i + 1];
        flip = a + f * (b - a);
    } else flip = aps_exp(-beta * (double)spin * mloc);
    if (M.suppress_flip && bound) flip = 0.0;
    // target sites: forward = right neighbour for +, own site for -; walls clip, torus wraps
    const bool wall_l = !M.periodic && p == 0, wall_r = !M.periodic && p == L - 1;
    const int o_l = wall_l ? occ_self : occ_left, o_r = wall_r ? occ_self : occ_right;
    const int o_f = plus ? o_r : occ_self;
    const bool open_l = !wall_l && o_l < K, open_r = !wall_r && o_r < K, open_f = plus && open_r;
    double hl = M.rate_diffusion * (double)open_l, hr = M.rate_diffusion * (double)open_r;
    double act = (plus || !M.minus_anchor) ? M.rate_active : 0.0;
    double leave = 0.0;
    const bool held = M.immobilize && !plus && anchored_site && bound;
    if (held) { act = 0.0; hl = 0.0; hr = 0.0; leave = M.k_exit; }
    double diff = hl + hr;
    if (!(plus && open_f)) act = 0.0;
    if (M.crowding) {
        double ff = 1.0 - (double)o_f / (double)K, fl = 1.0 - (double)o_l / (double)K, fr = 1.0 - (double)o_r / (double)K;
        ff = ff < 0.0 ? 0.0 : (ff > 1.0 ? 1.0 : ff);
        fl = fl < 0.0 ? 0.0 : (fl > 1.0 ? 1.0 : fl);
        fr = fr < 0.0 ? 0.0 : (fr > 1.0 ? 1.0 : fr);
        act *= ff;
        hl = M.rate_diffusion * (double)open_l * fl;
        hr = M.rate_diffusion * (double)open_r * fr;
        diff = hl + hr;
    }
    if (held) { diff = 0.0; act = 0.0; }
    Channels c;
    c.bind = (!bound && !plus && anchored_site && occ_self < K) ? M.k_on : 0.0;
    c.unbind = bound ? M.k_off : 0.0;
    c.diff = diff; c.act = act; c.flip = flip; c.leave = leave; c.left = hl; c.right = hr;
    c.total = ((((diff + act) + flip) + c.bind) + c.unbind) + leave;
    return c;
}

// Weight table (DESIGN.md "Weight table"): unnormalised taps of gaussian_filter1d(truncate=4) with the
// reflected images folded in, rounded to the grid 2^-q that keeps every possible partial sum exact.
// Weight table (DESIGN.md "Weight table"): unnormalised taps of gaussian_filter1d(truncate=4) with the
// reflected images folded in, rounded to the grid 2^-q that keeps every possible partial sum exact.
// sum_bits: bits a sum of weights may take in grid units (51: exact binary64 field; 29: the int32 field of the fp32 mode)
inline void weight_table(double sigma_grid, int L_, int K, bool periodic, std::vector<double> &table, int &tlen, int &q_out, int sum_bits = 51) {
    struct { double sigma_grid; int L, K; bool periodic; } p{sigma_grid, L_, K, periodic};
    table.clear();
    q_out = 0;
    if (!(p.sigma_grid > 0.0)) { tlen = 0; table.push_back(0.0); return; }
    const double s2 = p.sigma_grid * p.sigma_grid;
    const int64_t L = p.L;
    const int64_t lw = p.periodic ? L / 2 : (int64_t)(4.0 * p.sigma_grid + 0.5);
    const int64_t tmax = p.periodic ? L / 2 : std::min(lw, L);
    std::vector<double> w((size_t)tmax + 1);
    double wmax = 0.0;
    for (int64_t t = 0; t <= tmax; ++t) {
        double acc = 0.0;
        if (p.periodic) {
            const double a = (double)t;
            acc = aps_exp(-(0.5 * a * a) / s2);
        } else {
            for (int64_t k = 0;; ++k) {
                const int64_t d1 = t + 2 * L * k, d2 = 2 * L * k - t;
                bool any = false;
                if (d1 <= lw) { const double a = (double)d1; acc += aps_exp(-(0.5 * a * a) / s2); any = true; }
                if (k > 0 && d2 <= lw) { const double a = (double)d2; acc += aps_exp(-(0.5 * a * a) / s2); any = true; }
                if (!any) break;
            }
        }
        w[(size_t)t] = acc;
        wmax = std::max(wmax, acc);
    }
    const double nterm = p.periodic ? (double)p.K * (2.0 * (double)tmax + 1.0)
                                    : (lw < L ? (double)p.K * (2.0 * (double)lw + 1.0) : 2.0 * (double)p.K * (double)L);
    const double bound = std::ceil(nterm * wmax);
    int bits = 0;
    while (std::ldexp(1.0, bits) <= bound) ++bits;
    int q = std::min(45, sum_bits - bits);
    if (sum_bits < 32) {                                     // integer field: every single weight below 2^23 (24-bit multiply-add)
        int wb = 0;
        while (std::ldexp(1.0, wb) <= std::ceil(wmax)) ++wb;
        q = std::min(q, 22 - wb);
    }
    const double up = std::ldexp(1.0, q), down = std::ldexp(1.0, -q);
    int n = 0;
    for (int64_t t = 0; t <= tmax; ++t) {
        w[(size_t)t] = std::rint(w[(size_t)t] * up) * down;
        if (w[(size_t)t] != 0.0) n = (int)t + 1;
    }
    w.resize((size_t)n);
    w.push_back(0.0);
    table = std::move(w);
    tlen = n;
    q_out = q;
}


// value of the lane n below within the same row of 16 lanes (0.0 when that lane is outside the row): two 32-bit DPP moves
template <int N>
__device__ __forceinline__ double row_shr(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x110 | N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x110 | N, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
}

__device__ __forceinline__ double read_lane(double v, int lane) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// inclusive scan over one wavefront without touching LDS: Hillis-Steele inside the rows of 16 lanes (DPP), then the three
// row totals are added to the rows above them
__device__ __forceinline__ double wave_scan_inclusive(double v) {
    v += row_shr<1>(v); v += row_shr<2>(v); v += row_shr<4>(v); v += row_shr<8>(v);
    const int row = (threadIdx.x & 63) >> 4;
    const double t0 = read_lane(v, 15), t1 = read_lane(v, 31), t2 = read_lane(v, 47);
    double add = 0.0;
    if (row > 0) add = t0;
    if (row > 1) add += t1;
    if (row > 2) add += t2;
    return v + add;
}

// weight of source site p seen from site x, images folded in (same lookups as the stepper's field kernels)
__device__ inline double site_weight(const Model &M, const double *tab, int tlen, int x, int p) {
    const int L = M.L;
    int d = x > p ? x - p : p - x;
    if (M.periodic) { d = min(d, L - d); return tab[min(d, tlen)]; }
    const int s = x + p + 1;
    return tab[min(d, tlen)] + tab[min(min(s, 2 * L - s), tlen)];
}

}  // namespace
