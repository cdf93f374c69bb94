/* ORACLE (test infrastructure; never shipped, never the thing measured as the product).
 *
 * Plain-C restatement of the fixed-dt ("synchronous") lattice-gas stepper that the HIP path
 * implements.  It is written in the reference's own LATTICE formulation -- histogram the particles
 * onto the L sites, smooth the histogram with the truncated Gaussian, read the field back at the
 * particle sites -- i.e. the algorithm of compute_local_m_field + step_gillespie's rate section
 * (PARTICLE_solver_CLASS.py:216-246, :254-352), whereas the HIP kernel uses the equivalent
 * all-pairs form.  Two independent formulations agreeing bit for bit is the parity check.
 *
 * What is pinned against the reference (tests/test_oracle_sync.py):
 *   orc_field_sites   vs fixture G1 (reference m-field)          <= 1e-11 (weight-grid rounding)
 *   orc_rates         vs fixture G2 (reference rates/R vectors)  <= 1e-14 rel
 *   whole stepper     vs fixture G4 (reference ensemble statistics), statistical, dt -> 0
 * The Philox generator is pinned by the Random123 known-answer vectors, orc_exp by libm.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/build.py).  -ffp-contract=off matters:
 * every floating-point operation below is a single IEEE-754 operation (explicit fma() where a fused
 * one is meant), so the GPU executes the identical sequence and gets identical bits.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t L;               /* lattice sites                                   ref :41  */
    int32_t K;               /* site capacity                                   ref :43  */
    int32_t periodic;        /*                                                 ref :81  */
    int32_t field_mode;      /* 0 = global mean (sigma<=0, ref :219-221), 1 = kernel */
    int32_t tlen;            /* entries in the weight table (distance 0..tlen-1) */
    int32_t minus_anchor;    /* ref :83  */
    int32_t immobilize;      /* ref :82  */
    int32_t suppress_flip;   /* ref :54  */
    int32_t crowding;        /* ref :55  */
    int32_t flip_n;          /* intervals of flip_tab (0: the Curie-Weiss rate of ref :60) */
    double rate_diffusion;   /* already scaled, ref :45-50 */
    double rate_active;
    double beta;
    double k_on, k_off, k_exit;
    double dt;
    uint64_t seed;
    uint32_t ensemble;
    uint32_t reserved2;
    const double *flip_tab;  /* [2][flip_n + 1] or NULL: a caller's flip_rate_fn (ref :59-62, applied at :261-262) on the grid
                                m = -1 + 2 i / flip_n, row 0: sigma = +1, row 1: sigma = -1; interpolated linearly in m */
} orc_params;

/* ------------------------------------------------------------------ Philox4x32-10 (Random123) */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint32_t h0, l0, h1, l1;
        mulhilo(0xD2511F53u, c0, &h0, &l0);
        mulhilo(0xCD9E8D57u, c2, &h1, &l1);
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------ deterministic exp */
/* exp(x) from IEEE operations only (mul, fma, rint, exponent insert): the same bits on any
 * IEEE-754 machine.  |x| <= 700.  ~1 ulp. */
double orc_exp(double x) {
    const double LOG2E = 0x1.71547652b82fep+0;
    const double LN2_HI = 0x1.62e42fee00000p-1;
    const double LN2_LO = 0x1.a39ef35793c76p-33;
    static const double C[14] = {           /* 1/n!, n = 0..13 */
        0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p-1, 0x1.5555555555555p-3,
        0x1.5555555555555p-5, 0x1.1111111111111p-7, 0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13,
        0x1.a01a01a01a01ap-16, 0x1.71de3a556c734p-19, 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26,
        0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33};
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    double kf = rint(x * LOG2E);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = C[13];
    for (int n = 12; n >= 0; --n) p = fma(p, r, C[n]);
    int64_t k = (int64_t)kf;
    /* 2^k by exponent insert, in two factors so that k down to -1074 would still be fine */
    int64_t k1 = k / 2, k2 = k - k1;
    uint64_t b1 = (uint64_t)(k1 + 1023) << 52, b2 = (uint64_t)(k2 + 1023) << 52;
    double s1, s2;
    memcpy(&s1, &b1, 8);
    memcpy(&s2, &b2, 8);
    return p * s1 * s2;
}

/* ------------------------------------------------------------------ weight table
 * Unnormalised Gaussian taps of scipy's gaussian_filter1d (radius lw = int(4*sigma_g + 0.5); the
 * normalisation cancels in the ratio s_conv/tot_conv, ref :241-243), reflected images folded in
 * ("wrapped" over the 2L-periodic even extension) and rounded to the grid 2^-q so that every partial
 * sum of weights is exactly representable in binary64 -> sums are independent of summation order.
 *   periodic:  W[t] = exp(-0.5 t^2 / sigma_g^2), t = circular distance, 0 <= t <= L/2       (ref :114-116)
 *   reflect :  W[t] = sum_k w(|t + 2Lk|), |t + 2Lk| <= lw,  t = circular distance mod 2L, 0 <= t <= L
 * Returns the number of entries written (trailing zeros trimmed), q in *q_out; -1 if cap too small. */
/* sum_bits = bits a sum of weights may occupy in units of the grid: 51 for the exact binary64 field (two spare bits for
 * add / subtract sequences of the incremental formulations), 29 for the 32-bit integer field of the library's `fp32` mode
 * (same two spare bits under the int32 sign bit). */
int32_t orc_build_table_bits(double sigma_g, int32_t L, int32_t K, int32_t periodic, int32_t sum_bits, double *out, int32_t cap,
                             int32_t *q_out);

int32_t orc_build_table(double sigma_g, int32_t L, int32_t K, int32_t periodic, double *out, int32_t cap,
                        int32_t *q_out) {
    return orc_build_table_bits(sigma_g, L, K, periodic, 51, out, cap, q_out);
}

int32_t orc_build_table_bits(double sigma_g, int32_t L, int32_t K, int32_t periodic, int32_t sum_bits, double *out, int32_t cap,
                             int32_t *q_out) {
    const double s2 = sigma_g * sigma_g;
    int64_t lw = periodic ? (int64_t)(L / 2) : (int64_t)(4.0 * sigma_g + 0.5);
    int64_t tmax = periodic ? (int64_t)(L / 2) : (lw < (int64_t)L ? lw : (int64_t)L);
    if (tmax + 1 > cap) return -1;
    double wmax = 0.0;
    for (int64_t t = 0; t <= tmax; ++t) {
        double acc = 0.0;
        if (periodic) {
            double a = (double)t;
            acc = orc_exp(-(0.5 * a * a) / s2);
        } else {
            for (int64_t k = 0;; ++k) {             /* k = 0, then -k and +k together */
                int64_t d1 = t + 2 * (int64_t)L * k, d2 = 2 * (int64_t)L * k - t;
                int any = 0;
                if (d1 <= lw) { double a = (double)d1; acc += orc_exp(-(0.5 * a * a) / s2); any = 1; }
                if (k > 0 && d2 <= lw) { double a = (double)d2; acc += orc_exp(-(0.5 * a * a) / s2); any = 1; }
                if (!any) break;
            }
        }
        out[t] = acc;
        if (acc > wmax) wmax = acc;
    }
    /* number of non-zero terms one target can collect, times the largest weight */
    double nterm = periodic ? (double)K * (2.0 * (double)tmax + 1.0)
                            : (lw < (int64_t)L ? (double)K * (2.0 * (double)lw + 1.0)
                                               : 2.0 * (double)K * (double)L);
    double bound = ceil(nterm * wmax);
    int bits = 0;
    while (ldexp(1.0, bits) <= bound) ++bits;       /* bit length of the integer bound */
    int q = sum_bits - bits;
    if (q > 45) q = 45;
    if (sum_bits < 32) {                            /* integer field: every single weight below 2^23 (24-bit multiply-add) */
        int wb = 0;
        while (ldexp(1.0, wb) <= ceil(wmax)) ++wb;
        if (q > 22 - wb) q = 22 - wb;
    }
    const double up = ldexp(1.0, q), down = ldexp(1.0, -q);
    int32_t n = 0;
    for (int64_t t = 0; t <= tmax; ++t) {
        out[t] = rint(out[t] * up) * down;
        if (out[t] != 0.0) n = (int32_t)t + 1;
    }
    *q_out = q;
    return n;
}

/* ------------------------------------------------------------------ field */
static inline int64_t cdist(int64_t a, int64_t period) {
    int64_t t = a % period;
    if (t < 0) t += period;
    return t < period - t ? t : period - t;
}

/* S(x) = sum_y s[y] * Wsum(x,y), Wtot(x) = sum_y tot[y] * Wsum(x,y) for the listed target sites. */
static void field_at(const orc_params *P, const double *wtab, const int32_t *cp, const int32_t *cm,
                     const int32_t *targets, int64_t nt, double *S, double *W) {
    const int64_t L = P->L, T = P->tlen;
    if (P->field_mode == 0) {
        int64_t s = 0, w = 0;
        for (int64_t y = 0; y < L; ++y) { s += cp[y] - cm[y]; w += cp[y] + cm[y]; }
        for (int64_t i = 0; i < nt; ++i) { S[i] = (double)s; W[i] = (double)w; }
        return;
    }
    /* occupied sites once */
    int32_t *occ_y = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    int64_t ny = 0;
    for (int64_t y = 0; y < L; ++y) if (cp[y] + cm[y] > 0) occ_y[ny++] = (int32_t)y;
    const int windowed = (T - 1 < L);        /* table shorter than the box: sources are in windows */
    for (int64_t i = 0; i < nt; ++i) {
        const int64_t x = targets[i];
        double s = 0.0, w = 0.0;
        if (P->periodic) {
            for (int64_t j = 0; j < ny; ++j) {
                int64_t y = occ_y[j], t = cdist(x - y, L);
                if (t < T) { double g = wtab[t]; s += g * (double)(cp[y] - cm[y]); w += g * (double)(cp[y] + cm[y]); }
            }
        } else if (windowed) {
            int64_t lo = x - (T - 1), hi = x + (T - 1);
            for (int64_t y = (lo < 0 ? 0 : lo); y <= hi && y < L; ++y) {
                int32_t tot = cp[y] + cm[y];
                if (!tot) continue;
                double g = wtab[y > x ? y - x : x - y];
                s += g * (double)(cp[y] - cm[y]); w += g * (double)tot;
            }
            /* image at -1-y (left wall):  distance x + y + 1 */
            for (int64_t y = 0; y < L && x + y + 1 < T; ++y) {
                int32_t tot = cp[y] + cm[y];
                if (!tot) continue;
                double g = wtab[x + y + 1];
                s += g * (double)(cp[y] - cm[y]); w += g * (double)tot;
            }
            /* image at 2L-1-y (right wall): distance 2L - 1 - x - y */
            for (int64_t y = L - 1; y >= 0 && 2 * L - 1 - x - y < T; --y) {
                int32_t tot = cp[y] + cm[y];
                if (!tot) continue;
                double g = wtab[2 * L - 1 - x - y];
                s += g * (double)(cp[y] - cm[y]); w += g * (double)tot;
            }
        } else {
            for (int64_t j = 0; j < ny; ++j) {
                int64_t y = occ_y[j];
                int64_t t1 = cdist(x - y, 2 * L), t2 = cdist(x + y + 1, 2 * L);
                double g = (t1 < T ? wtab[t1] : 0.0) + (t2 < T ? wtab[t2] : 0.0);
                s += g * (double)(cp[y] - cm[y]); w += g * (double)(cp[y] + cm[y]);
            }
        }
        S[i] = s; W[i] = w;
    }
    free(occ_y);
}

static inline double clip_ratio(double s, double w) {
    if (!(w > 0.0)) return 0.0;              /* ref :241-243: zero where tot_conv <= 0 */
    double m = s / w;
    return m > 1.0 ? 1.0 : (m < -1.0 ? -1.0 : m);   /* ref :245 */
}

static void histogram(int32_t L, int64_t n, const int32_t *pos, const int8_t *spin, const uint8_t *alive,
                      int32_t *cp, int32_t *cm) {
    memset(cp, 0, sizeof(int32_t) * (size_t)L);
    memset(cm, 0, sizeof(int32_t) * (size_t)L);
    for (int64_t i = 0; i < n; ++i)
        if (alive[i]) { if (spin[i] > 0) cp[pos[i]]++; else cm[pos[i]]++; }
}

/* m-field on all L sites (observation path, ref :496/:512/:525) + the site histograms. */
void orc_field_sites(const orc_params *P, const double *wtab, int64_t n, const int32_t *pos,
                     const int8_t *spin, const uint8_t *alive, int32_t *cp, int32_t *cm, double *m,
                     double *S_out, double *W_out) {
    const int32_t L = P->L;
    histogram(L, n, pos, spin, alive, cp, cm);
    int32_t *tg = (int32_t *)calloc((size_t)L, sizeof(int32_t));
    double *S = (double *)malloc(sizeof(double) * (size_t)L), *W = (double *)malloc(sizeof(double) * (size_t)L);
    for (int32_t x = 0; x < L; ++x) tg[x] = x;
    field_at(P, wtab, cp, cm, tg, L, S, W);
    for (int32_t x = 0; x < L; ++x) m[x] = clip_ratio(S[x], W[x]);
    if (S_out) memcpy(S_out, S, sizeof(double) * (size_t)L);
    if (W_out) memcpy(W_out, W, sizeof(double) * (size_t)L);
    free(tg); free(S); free(W);
}

/* S, W and the four occupancies (self, forward, left, right target) per particle -- the quantities
 * the all-pairs HIP kernel accumulates.  Dead particles get zeros. */
void orc_pair_sums(const orc_params *P, const double *wtab, int64_t n, const int32_t *pos, const int8_t *spin,
                   const uint8_t *alive, double *S, double *W, int32_t *occ4) {
    const int32_t L = P->L;
    int32_t *cp = (int32_t *)malloc(sizeof(int32_t) * (size_t)L), *cm = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    histogram(L, n, pos, spin, alive, cp, cm);
    field_at(P, wtab, cp, cm, pos, n, S, W);
    for (int64_t i = 0; i < n; ++i) {
        if (!alive[i]) { S[i] = W[i] = 0.0; occ4[4*i] = occ4[4*i+1] = occ4[4*i+2] = occ4[4*i+3] = 0; continue; }
        int64_t p = pos[i], f = p + (spin[i] > 0), l = p - 1, r = p + 1;
        if (P->periodic) { f = (f + L) % L; l = (l + L) % L; r = (r + L) % L; }
        else { f = f < 0 ? 0 : (f > L - 1 ? L - 1 : f); l = l < 0 ? 0 : l; r = r > L - 1 ? L - 1 : r; }
        occ4[4*i] = cp[p] + cm[p]; occ4[4*i+1] = cp[f] + cm[f]; occ4[4*i+2] = cp[l] + cm[l]; occ4[4*i+3] = cp[r] + cm[r];
    }
    free(cp); free(cm);
}

/* ------------------------------------------------------------------ rates (ref :254-352) */
enum { CH_DIFF = 0, CH_ACT, CH_FLIP, CH_BIND, CH_UNBIND, CH_EXIT, CH_LEFT, CH_RIGHT, CH_TOTAL, CH_N };

typedef struct { double ch[CH_N]; int32_t fwd, left, right; } chan_t;

static void channels(const orc_params *P, const uint8_t *anchor, int32_t p, int spin, int bound, double m,
                     const int32_t *occ, chan_t *c, double (*expfn)(double)) {
    const int32_t L = P->L, K = P->K;
    const int plus = spin > 0;
    double flip;
    if (P->flip_tab) {                                                  /* ref :59-62 with a tabulated callable, :261-262 */
        const double u = (m + 1.0) * (0.5 * (double)P->flip_n);
        int32_t i = (int32_t)u;
        i = i < 0 ? 0 : (i >= P->flip_n ? P->flip_n - 1 : i);
        const double fr = u - (double)i;
        const double *tb = P->flip_tab + (plus ? 0 : P->flip_n + 1);
        const double a = tb[i], b = tb[i + 1];
        flip = a + fr * (b - a);
    } else flip = expfn(-P->beta * (double)spin * m);                   /* ref :60, :262 */
    if (P->suppress_flip && bound) flip = 0.0;                          /* ref :266-267 */
    int32_t f = p + (plus ? 1 : 0), l = p - 1, r = p + 1;               /* ref :276-291 */
    if (P->periodic) { f %= L; l = (l + L) % L; r %= L; }
    else { if (f > L - 1) f = L - 1; if (l < 0) l = 0; if (r > L - 1) r = L - 1; }
    const int open_f = occ[f] < K && f != p, open_l = occ[l] < K && l != p, open_r = occ[r] < K && r != p;
    double hl = P->rate_diffusion * (double)open_l, hr = P->rate_diffusion * (double)open_r;   /* ref :304-305 */
    double act = (plus || !P->minus_anchor) ? P->rate_active : 0.0;     /* ref :269-272 */
    double leave = 0.0;
    const int held = P->immobilize && !plus && anchor[p] && bound;      /* ref :307-312 */
    if (held) { act = 0.0; hl = 0.0; hr = 0.0; leave = P->k_exit; }
    double diff = hl + hr;
    if (!(plus && open_f)) act = 0.0;                                   /* ref :317-319 */
    if (P->crowding) {                                                  /* ref :322-336 */
        double ff = 1.0 - (double)occ[f] / (double)K, fl = 1.0 - (double)occ[l] / (double)K,
               fr = 1.0 - (double)occ[r] / (double)K;
        ff = ff < 0.0 ? 0.0 : (ff > 1.0 ? 1.0 : ff);
        fl = fl < 0.0 ? 0.0 : (fl > 1.0 ? 1.0 : fl);
        fr = fr < 0.0 ? 0.0 : (fr > 1.0 ? 1.0 : fr);
        act *= ff;
        hl = P->rate_diffusion * (double)open_l * fl;
        hr = P->rate_diffusion * (double)open_r * fr;
        diff = hl + hr;
    }
    if (held) { diff = 0.0; act = 0.0; }                                /* ref :338-340 */
    const double attach = (!bound && !plus && anchor[p] && occ[p] < K) ? P->k_on : 0.0;   /* ref :343-345 */
    const double detach = bound ? P->k_off : 0.0;                       /* ref :347-348 */
    c->ch[CH_DIFF] = diff; c->ch[CH_ACT] = act; c->ch[CH_FLIP] = flip; c->ch[CH_BIND] = attach;
    c->ch[CH_UNBIND] = detach; c->ch[CH_EXIT] = leave; c->ch[CH_LEFT] = hl; c->ch[CH_RIGHT] = hr;
    c->ch[CH_TOTAL] = ((((diff + act) + flip) + attach) + detach) + leave;      /* ref :351 */
    c->fwd = f; c->left = l; c->right = r;
}

/* Rate vectors from a GIVEN m-field (the reference passes m_field into step_gillespie), with libm exp
 * (use_libm=1, to compare with numpy's exp in fixture G2) or the deterministic exp.
 * out: 9 x n doubles, channel-major in the order of the enum above. */
void orc_rates(const orc_params *P, const uint8_t *anchor, int64_t n, const int32_t *pos, const int8_t *spin,
               const uint8_t *bound, const double *m_field, int32_t use_libm, double *out) {
    const int32_t L = P->L;
    uint8_t *alive = (uint8_t *)malloc((size_t)n);
    memset(alive, 1, (size_t)n);
    int32_t *cp = (int32_t *)malloc(sizeof(int32_t) * (size_t)L), *cm = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    int32_t *occ = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    histogram(L, n, pos, spin, alive, cp, cm);
    for (int32_t x = 0; x < L; ++x) occ[x] = cp[x] + cm[x];
    for (int64_t i = 0; i < n; ++i) {
        chan_t c;
        channels(P, anchor, pos[i], spin[i], bound[i], m_field[pos[i]], occ, &c, use_libm ? exp : orc_exp);
        for (int k = 0; k < CH_N; ++k) out[(int64_t)k * n + i] = c.ch[k];
    }
    free(alive); free(cp); free(cm); free(occ);
}

/* ------------------------------------------------------------------ one synchronous step
 * proposal codes */
enum { EV_NONE = 0, EV_LEFT = 1, EV_RIGHT = 2, EV_FWD = 3, EV_BIND = 4, EV_UNBIND = 5, EV_EXIT = 6, EV_FLIP = 7 };

/* Phase 1 of a step for the particles lo <= i < hi only (what one rank of a sharded job evaluates):
 * proposal byte = event code | (free capacity of the hop target - 1) << 3, the format the ranks exchange.
 * S/W are optional outputs for [lo, hi). */
void orc_sync_propose(const orc_params *P, const double *wtab, const uint8_t *anchor, int64_t n, const int32_t *pos,
                      const int8_t *spin, const uint8_t *bound, const uint8_t *alive, uint64_t step, int64_t lo,
                      int64_t hi, uint8_t *prop, double *Sout, double *Wout) {
    const int32_t L = P->L, K = P->K;
    int32_t *cp = (int32_t *)malloc(sizeof(int32_t) * (size_t)L), *cm = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    int32_t *occ = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    const int64_t m = hi - lo;
    double *S = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1)), *W = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    histogram(L, n, pos, spin, alive, cp, cm);
    for (int32_t x = 0; x < L; ++x) occ[x] = cp[x] + cm[x];
    field_at(P, wtab, cp, cm, pos + lo, m, S, W);
    const uint32_t key[2] = {(uint32_t)P->seed, (uint32_t)(P->seed >> 32)};
    for (int64_t i = lo; i < hi; ++i) {
        prop[i] = EV_NONE;
        if (!alive[i]) continue;
        chan_t c;
        channels(P, anchor, pos[i], spin[i], bound[i], clip_ratio(S[i - lo], W[i - lo]), occ, &c, orc_exp);
        const double r = c.ch[CH_TOTAL];
        const uint32_t ctr[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)i, P->ensemble};
        uint32_t x[4];
        orc_philox4x32_10(ctr, key, x);
        const double u0 = ((double)(x[0] >> 5) * 67108864.0 + (double)(x[1] >> 6)) * 0x1.0p-53;
        const double u1 = (double)x[2] * 0x1.0p-32, u2 = (double)x[3] * 0x1.0p-32;
        const double p_fire = 1.0 - orc_exp(-(r * P->dt));
        if (!(u0 < p_fire)) continue;
        const double v = u1 * r;                                        /* ref :362 */
        const double e_diff = c.ch[CH_DIFF], e_act = e_diff + c.ch[CH_ACT], e_bind = e_act + c.ch[CH_BIND],
                     e_unbind = e_bind + c.ch[CH_UNBIND], e_exit = e_unbind + c.ch[CH_EXIT];   /* ref :363-367 */
        int ev = EV_NONE, target = -1;
        if (v < e_diff) {
            const double a = c.ch[CH_LEFT], b = c.ch[CH_RIGHT];
            if (a + b <= 0.0) continue;
            if (u2 < a / (a + b)) { ev = EV_LEFT; target = c.left; }    /* ref :378-381 */
            else { ev = EV_RIGHT; target = c.right; }
        } else if (v < e_act) { ev = EV_FWD; target = c.fwd; }
        else if (v < e_bind) ev = EV_BIND;
        else if (v < e_unbind) ev = EV_UNBIND;
        else if (v < e_exit) ev = EV_EXIT;
        else ev = EV_FLIP;
        int cap = target >= 0 ? K - occ[target] : 1;
        cap = cap < 1 ? 1 : (cap > 32 ? 32 : cap);
        prop[i] = (uint8_t)(ev | ((cap - 1) << 3));
    }
    if (Sout) memcpy(Sout, S, sizeof(double) * (size_t)m);
    if (Wout) memcpy(Wout, W, sizeof(double) * (size_t)m);
    free(cp); free(cm); free(occ); free(S); free(W);
}

/* Phase 2: apply ALL n proposals.  Hops into a site are granted in increasing particle index while fewer
 * than `cap` (the free capacity at step start, carried in the proposal byte) have been granted; everything
 * else always succeeds.  A pure function of (state, proposals): every rank computes the same result. */
int32_t orc_sync_commit(const orc_params *P, int64_t n, int32_t *pos, int8_t *spin, uint8_t *bound, uint8_t *alive,
                        uint64_t step, const uint8_t *prop, uint8_t *accepted_out, double *exit_log, int64_t exit_cap,
                        int64_t *n_exit) {
    const int32_t L = P->L;
    int32_t *taken = (int32_t *)calloc((size_t)L, sizeof(int32_t));
    int32_t rc = 0;
    const double t_now = (double)step * P->dt;
    for (int64_t i = 0; i < n; ++i) {
        int ok = 0;
        const int ev = prop[i] & 7, cap = (prop[i] >> 3) + 1;
        if (alive[i]) switch (ev) {
            case EV_LEFT: case EV_RIGHT: case EV_FWD: {
                int32_t s = ev == EV_LEFT ? pos[i] - 1 : pos[i] + 1;
                if (P->periodic) s = s < 0 ? s + L : (s >= L ? s - L : s);
                if (taken[s] < cap) { taken[s]++; pos[i] = s; ok = 1; }
                break;
            }
            case EV_BIND: bound[i] = 1; ok = 1; break;
            case EV_UNBIND: bound[i] = 0; ok = 1; break;
            case EV_FLIP: spin[i] = (int8_t)-spin[i]; ok = 1; break;
            case EV_EXIT:
                alive[i] = 0; ok = 1;
                if (*n_exit < exit_cap) {
                    exit_log[3 * *n_exit] = t_now; exit_log[3 * *n_exit + 1] = (double)pos[i];
                    exit_log[3 * *n_exit + 2] = (double)i; ++*n_exit;
                } else rc = -1;
                break;
            default: break;
        }
        if (accepted_out) accepted_out[i] = (uint8_t)ok;
    }
    free(taken);
    return rc;
}

/* One synchronous step = propose for everybody + commit.  Index i is the particle's original index (Philox
 * counter word 2).  Optional outputs (may be NULL): prop[n], accepted[n], Sout/Wout[n].  Exits are appended
 * as (time, position, index) triples to exit_log; returns 0, or -1 if exit_log overflowed. */
int32_t orc_sync_step(const orc_params *P, const double *wtab, const uint8_t *anchor, int64_t n, int32_t *pos,
                      int8_t *spin, uint8_t *bound, uint8_t *alive, uint64_t step, uint8_t *prop_out,
                      uint8_t *accepted_out, double *Sout, double *Wout, double *exit_log, int64_t exit_cap,
                      int64_t *n_exit) {
    uint8_t *prop = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    orc_sync_propose(P, wtab, anchor, n, pos, spin, bound, alive, step, 0, n, prop, Sout, Wout);
    const int32_t rc = orc_sync_commit(P, n, pos, spin, bound, alive, step, prop, accepted_out, exit_log, exit_cap, n_exit);
    if (prop_out) for (int64_t i = 0; i < n; ++i) prop_out[i] = prop[i] & 7;
    free(prop);
    return rc;
}

/* nsteps steps in a row starting at step index step0 (CPU baseline loop / statistical runs). */
int32_t orc_sync_run(const orc_params *P, const double *wtab, const uint8_t *anchor, int64_t n, int32_t *pos,
                     int8_t *spin, uint8_t *bound, uint8_t *alive, uint64_t step0, int64_t nsteps,
                     double *exit_log, int64_t exit_cap, int64_t *n_exit) {
    int32_t rc = 0;
    for (int64_t s = 0; s < nsteps; ++s)
        rc |= orc_sync_step(P, wtab, anchor, n, pos, spin, bound, alive, step0 + (uint64_t)s, 0, 0, 0, 0,
                            exit_log, exit_cap, n_exit);
    return rc;
}
