// pde_hip.hip -- MI355X (gfx950) implementation of the C ABI in include/pde.h.
//
// Replaces the time loop of the reference's IMEXPDE (IMEX_PDE_solver_class.py:236-290): one PERSISTENT workgroup
// per system keeps rho_plus, rho_minus, the diffused fields and the magnetisation in LDS and runs all nsteps
// without leaving the kernel (a step is ~10 phases separated by workgroup barriers; nothing but the requested
// series is written to HBM).  Systems of a batch differ only in beta (the reference's sweep parameter).
//
//   implicit diffusion  (I - gamma dt Lap / dx^2) x = rho          ref :68-82, :192-193
//       constant matrix -> Thomas factorisation once on the host; the two triangular sweeps are first-order linear
//       recurrences y_i = a_i y_{i-1} + b_i, evaluated as a workgroup-wide scan of affine maps (chunk per thread,
//       Hillis-Steele over the 256 chunk maps in LDS); periodic corners by Sherman-Morrison
//   magnetisation       local ratio | circular Gaussian convolution | global mean      ref :156-168
//       the reference multiplies rfft's; here the periodic kernel (same normalised taps, cut where they fall below
//       1e-17 of the centre tap) is applied directly with a sliding 4-site register window
//   reaction/advection/clip/renormalise                                                   ref :195-233
//   observables per step: mean m, var(total), lowest rfft modes (direct DFT), snapshots   ref :243-255
//   tracers: Euler-Maruyama flip + drift + noise, windowed v_eff / D_eff                  ref :257-287
//
// Arithmetic: binary64.  Not bit-identical to the reference (different linear solver, summation orders and libm):
// tests/test_gpu_pde.py holds the stated tolerances against oracle/pde_numpy.py, which IS bit-identical to it.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "pde.h"

namespace {

constexpr int NT = 256;                 // threads per workgroup = chunks of the recurrences
std::string g_err;

struct PdeArgs {
    pde_params p;
    int n_snap, ktaps, chunk;           // chunk = sites per thread in the scans
    double dx, sm_coef, sm_denom;       // Sherman-Morrison: x = y - z * (y_0 + sm_coef y_{L-1}) / sm_denom
    const double *beta, *rho_p0, *rho_m0, *tracer_x0;
    const int8_t *tracer_s0;
    const double *rand_u, *rand_n;
    const double *fw, *finv, *fu, *fz;  // factorisation: multipliers, 1/pivots, upper diagonal, S-M vector z   [L]
    const double *ktab;                 // [ktaps + 1] normalised kernel taps by distance
    const double *twc, *tws;            // [L] cos / sin(2 pi j / L)
    double *rho_p, *rho_m, *m_series, *var_series, *v_eff, *D_eff, *snapshots, *m_snapshots, *fft_re, *fft_im, *tracer_x;
    int8_t *tracer_s;
    double *work;                       // systems beyond LDS: [n_systems][work_stride] fields + kernel taps in global memory (else null)
    long long work_stride;
    double *hist;                       // [n_systems][window][n_tracers] ring of unwrapped tracer positions
    double *trx; int8_t *trs;           // [n_systems][n_tracers] working tracer state
};

__device__ inline double cw_rate(double beta, double sigma, double m) {      // ref :64-66
    const double r = exp(-beta * sigma * m);
    return r < 1e-8 ? 1e-8 : (r > 1e8 ? 1e8 : r);
}

// Philox4x32-10 (Random123), as in the particle stepper
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sum over the workgroup; every thread gets the result.  `red` = NT doubles of LDS scratch.
__device__ inline double block_sum(double v, double *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();                                          // scratch may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += red[w];
    return s;
}

// Exclusive scan of 2 x NT affine maps x -> A x + B (two fields at once) in LDS, Hillis-Steele.  Logical order:
// thread t is element t (forward) or NT-1-t (backward).  Returns the composition of all maps BEFORE this thread's
// element applied to the start value 0, i.e. the value entering this thread's chunk.
__device__ inline void scan_affine2(double A0, double B0, double A1, double B1, double4 *buf, bool backward, double &in0, double &in1) {
    const int j = backward ? NT - 1 - (int)threadIdx.x : (int)threadIdx.x;
    double4 *cur = buf, *nxt = buf + NT;
    __syncthreads();
    cur[j] = make_double4(A0, B0, A1, B1);
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
        double4 me = cur[j];
        if (j >= off) {                                       // me after prev:  x -> me.A (prev.A x + prev.B) + me.B
            const double4 pv = cur[j - off];
            me = make_double4(me.x * pv.x, me.x * pv.y + me.y, me.z * pv.z, me.z * pv.w + me.w);
        }
        nxt[j] = me;
        __syncthreads();
        double4 *sw = cur; cur = nxt; nxt = sw;
    }
    if (j == 0) { in0 = 0.0; in1 = 0.0; }
    else { const double4 pv = cur[j - 1]; in0 = pv.y; in1 = pv.w; }
}

// x = A^{-1} d for both fields: d in (dp, dm), result overwrites them.
__device__ inline void diffuse2(const PdeArgs &a, double *dp, double *dm, double4 *scan, double *red) {
    const int L = a.p.L, t = threadIdx.x, c0 = t * a.chunk, c1 = min(L, c0 + a.chunk);
    // forward: y_i = d_i - w_i y_{i-1}
    double A = 1.0, Bp = 0.0, Bm = 0.0;
    for (int i = c0; i < c1; ++i) { const double w = a.fw[i]; Bp = dp[i] - w * Bp; Bm = dm[i] - w * Bm; A = -w * A; }
    double yp, ym;
    scan_affine2(A, Bp, A, Bm, scan, false, yp, ym);
    for (int i = c0; i < c1; ++i) { const double w = a.fw[i]; yp = dp[i] - w * yp; ym = dm[i] - w * ym; dp[i] = yp; dm[i] = ym; }
    // backward: x_i = inv_i y_i - (u_i inv_i) x_{i+1}
    A = 1.0; Bp = 0.0; Bm = 0.0;
    for (int i = c1 - 1; i >= c0; --i) { const double iv = a.finv[i], q = -a.fu[i] * iv; Bp = dp[i] * iv + q * Bp; Bm = dm[i] * iv + q * Bm; A = q * A; }
    double xp, xm;
    scan_affine2(A, Bp, A, Bm, scan, true, xp, xm);
    __syncthreads();
    for (int i = c1 - 1; i >= c0; --i) { const double iv = a.finv[i], q = -a.fu[i] * iv; xp = dp[i] * iv + q * xp; xm = dm[i] * iv + q * xm; dp[i] = xp; dm[i] = xm; }
    __syncthreads();
    if (a.p.periodic) {                                       // Sherman-Morrison correction for the two corner entries
        const double fp = (dp[0] + a.sm_coef * dp[L - 1]) / a.sm_denom, fm = (dm[0] + a.sm_coef * dm[L - 1]) / a.sm_denom;
        __syncthreads();
        for (int i = t; i < L; i += NT) { const double z = a.fz[i]; dp[i] -= z * fp; dm[i] -= z * fm; }
        __syncthreads();
    }
    (void)red;
}

__global__ __launch_bounds__(NT) void pde_kernel(const PdeArgs a) {
    extern __shared__ double lds[];
    const int L = a.p.L, t = threadIdx.x, sys = blockIdx.x, ntr = a.p.n_tracers, nsteps = a.p.nsteps;
    // the five fields and the kernel taps: in LDS when they fit (L <= ~3000), else in this system's slab of global memory
    // (one workgroup = one CU: its L1 / the L2 serve it, workgroup barriers order the accesses); the scan scratch stays in LDS
    double *fields = a.work ? a.work + (size_t)sys * (size_t)a.work_stride : lds;
    double *rp = fields, *rm = rp + L, *xp = rm + L, *xm = xp + L, *mf = xm + L, *ktab = mf + L;
    double *red = a.work ? lds : ktab + ((a.ktaps + 2) & ~1);
    double4 *scan = a.work ? reinterpret_cast<double4 *>(lds + ((NT + 3) & ~3))
                           : reinterpret_cast<double4 *>(lds + ((5 * L + ((a.ktaps + 2) & ~1) + NT + 3) & ~3));   // 32-byte aligned
    const double beta = a.beta[sys], dx = a.dx, dt = a.p.dt, lam = a.p.lam;
    for (int i = t; i < L; i += NT) { rp[i] = a.rho_p0[(size_t)sys * L + i]; rm[i] = a.rho_m0[(size_t)sys * L + i]; }
    for (int i = t; i <= a.ktaps; i += NT) ktab[i] = a.ktab[i];
    double *trx = a.trx + (size_t)sys * ntr;
    int8_t *trs = a.trs + (size_t)sys * ntr;
    for (int i = t; i < ntr; i += NT) { trx[i] = a.tracer_x0[(size_t)sys * ntr + i]; trs[i] = a.tracer_s0[(size_t)sys * ntr + i]; }
    __syncthreads();
    const double noise_amp = sqrt(2.0 * a.p.gamma * dt);
    for (int n = 0; n <= nsteps; ++n) {
        // ---- magnetisation of the current state (ref :156-168): used by the observables, the tracers and step()
        double m_global = 0.0;
        if (a.p.kernel_mode == 0) {
            for (int i = t; i < L; i += NT) mf[i] = (rp[i] - rm[i]) / (rp[i] + rm[i] + 1e-12);
        } else if (a.p.kernel_mode == 2) {
            double s = 0.0, w = 0.0;
            for (int i = t; i < L; i += NT) { s += rp[i] - rm[i]; w += rp[i] + rm[i]; }
            s = block_sum(s, red); w = block_sum(w, red);
            m_global = s / (w + 1e-12);
            for (int i = t; i < L; i += NT) mf[i] = m_global;
        } else {                                               // circular convolution, 4 consecutive sites per thread
            for (int base = 4 * t; base < L; base += 4 * NT) {
                double num[4] = {0, 0, 0, 0}, den[4] = {0, 0, 0, 0};
                // window holds s, tot at sites base + j + (0..3); slide j from -ktaps to +ktaps
                int idx = base - a.ktaps;
                idx %= L; if (idx < 0) idx += L;
                double sp[4], sm[4];
#pragma unroll
                for (int k = 0; k < 3; ++k) { sp[k + 1] = rp[idx]; sm[k + 1] = rm[idx]; idx = idx + 1 == L ? 0 : idx + 1; }
                for (int j = -a.ktaps; j <= a.ktaps; ++j) {
                    sp[0] = sp[1]; sp[1] = sp[2]; sp[2] = sp[3]; sm[0] = sm[1]; sm[1] = sm[2]; sm[2] = sm[3];
                    sp[3] = rp[idx]; sm[3] = rm[idx]; idx = idx + 1 == L ? 0 : idx + 1;
                    // site base + k sees source base + k + j  <=>  window slot k holds it when the window starts at base + j
                    const double kv = ktab[j < 0 ? -j : j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { num[k] += kv * (sp[k] - sm[k]); den[k] += kv * (sp[k] + sm[k]); }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) if (base + k < L) mf[base + k] = num[k] / (den[k] + 1e-12);
            }
        }
        __syncthreads();
        // ---- observables (ref :243-255)
        {
            double sm_ = 0.0, st = 0.0;
            for (int i = t; i < L; i += NT) { sm_ += mf[i]; st += rp[i] + rm[i]; }
            sm_ = block_sum(sm_, red); st = block_sum(st, red);
            const double mean_t = st / L;
            double sv = 0.0;
            for (int i = t; i < L; i += NT) { const double d = rp[i] + rm[i] - mean_t; sv += d * d; }
            sv = block_sum(sv, red);
            if (t == 0) {
                if (a.m_series) a.m_series[(size_t)sys * (nsteps + 1) + n] = a.p.kernel_mode == 2 ? m_global : sm_ / L;
                if (a.var_series) a.var_series[(size_t)sys * (nsteps + 1) + n] = sv / L;
            }
            if (a.fft_re)
                for (int k = t; k < a.p.n_fft_modes; k += NT) {   // rfft(total)[k] / L = sum total_i (cos - i sin)(2 pi k i / L) / L
                    double re = 0.0, im = 0.0;
                    int ph = 0;
                    for (int i = 0; i < L; ++i) {
                        const double v = rp[i] + rm[i];
                        re += v * a.twc[ph]; im -= v * a.tws[ph];
                        ph += k; if (ph >= L) ph -= L;
                    }
                    a.fft_re[((size_t)sys * (nsteps + 1) + n) * a.p.n_fft_modes + k] = re / L;
                    a.fft_im[((size_t)sys * (nsteps + 1) + n) * a.p.n_fft_modes + k] = im / L;
                }
            if (n % a.p.snapshot_interval == 0) {
                const size_t o = ((size_t)sys * a.n_snap + n / a.p.snapshot_interval) * L;
                for (int i = t; i < L; i += NT) {
                    if (a.snapshots) a.snapshots[o + i] = rp[i] + rm[i];
                    if (a.m_snapshots) a.m_snapshots[o + i] = rp[i] - rm[i];
                }
            }
        }
        // ---- tracers (ref :257-287)
        if (ntr > 0) {
            double sdr = 0.0;
            const bool windowed = n >= a.p.window;
            double *hist = a.hist + (size_t)sys * a.p.window * ntr;
            for (int i = t; i < ntr; i += NT) {
                double xu = trx[i];
                double xw = fmod(xu, a.p.xlim);                // numpy's % : floor modulo
                if (xw != 0.0 && xw < 0.0) xw += a.p.xlim;
                int idx = (int)(xw / dx) % L;
                const double m_loc = mf[idx];
                int s = trs[i];
                double u, g;
                if (a.rand_u) {
                    const size_t o = ((size_t)sys * (nsteps + 1) + n) * ntr + i;
                    u = a.rand_u[o]; g = a.rand_n[o];
                } else {
                    uint32_t x[4];
                    philox4x32_10((uint32_t)n, (uint32_t)i, (uint32_t)sys, 0x7AC3u, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32), x);
                    u = ((double)(x[0] >> 5) * 67108864.0 + (double)(x[1] >> 6)) * 0x1.0p-53;
                    const double u1 = ((double)x[2] + 0.5) * 0x1.0p-32, u2 = ((double)x[3] + 0.5) * 0x1.0p-32;
                    g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
                }
                const double rate = cw_rate(beta, (double)s, m_loc);
                if (u < rate * dt) s = -s;
                xu += lam * (double)s * dt + noise_amp * g;
                trx[i] = xu; trs[i] = (int8_t)s;
                hist[(size_t)(n % a.p.window) * ntr + i] = xu;
            }
            __syncthreads();
            if (windowed) {                                    // dr = x_n - x_{n - window + 1}  (ref: history[-window])
                const double *old = hist + (size_t)((n + 1) % a.p.window) * ntr;
                for (int i = t; i < ntr; i += NT) sdr += trx[i] - old[i];
                const double mean_dr = block_sum(sdr, red) / ntr;
                double sv = 0.0;
                for (int i = t; i < ntr; i += NT) { const double d = trx[i] - old[i] - mean_dr; sv += d * d; }
                sv = block_sum(sv, red) / ntr;
                if (t == 0) {
                    if (a.v_eff) a.v_eff[(size_t)sys * (nsteps + 1) + n] = mean_dr / (a.p.window * dt);
                    if (a.D_eff) a.D_eff[(size_t)sys * (nsteps + 1) + n] = sv / (2 * a.p.window * dt);
                }
            } else if (t == 0) {
                const double nan = __longlong_as_double(0x7ff8000000000000ll);
                if (a.v_eff) a.v_eff[(size_t)sys * (nsteps + 1) + n] = nan;
                if (a.D_eff) a.D_eff[(size_t)sys * (nsteps + 1) + n] = nan;
            }
        }
        if (n == nsteps) break;
        // ---- step() (ref :190-233)
        for (int i = t; i < L; i += NT) { xp[i] = rp[i]; xm[i] = rm[i]; }
        __syncthreads();
        diffuse2(a, xp, xm, scan, red);
        double m0 = 0.0;
        for (int i = t; i < L; i += NT) m0 += xp[i] + xm[i];
        m0 = block_sum(m0, red);
        const int per = a.p.periodic;
        if (!a.p.anchored_minus) {
            for (int i = t; i < L; i += NT) {
                const double dpl = i > 0 ? (xp[i] - xp[i - 1]) / dx : (per ? (xp[0] - xp[L - 1]) / dx : 0.0);     // right-moving: backward difference
                const double dmr = i < L - 1 ? (xm[i + 1] - xm[i]) / dx : (per ? (xm[0] - xm[L - 1]) / dx : 0.0); // left-moving: forward difference
                const double Rp = cw_rate(beta, -1.0, mf[i]) * xm[i] - cw_rate(beta, 1.0, mf[i]) * xp[i];
                const double np_ = xp[i] + dt * (-lam * dpl + Rp), nm_ = xm[i] + dt * (lam * dmr + (-Rp));
                rp[i] = np_ < 0.0 ? 0.0 : np_; rm[i] = nm_ < 0.0 ? 0.0 : nm_;
            }
        } else {
            for (int i = t; i < L; i += NT) {                  // reaction first, into rp (star_p) / rm (star_m)
                const double Rp = cw_rate(beta, -1.0, mf[i]) * xm[i] - cw_rate(beta, 1.0, mf[i]) * xp[i];
                const double sp_ = xp[i] + dt * Rp, sm_ = xm[i] + dt * (-Rp);
                rp[i] = sp_ < 0.0 ? 0.0 : sp_; rm[i] = sm_ < 0.0 ? 0.0 : sm_;
            }
            __syncthreads();
            for (int i = t; i < L; i += NT) {                  // advection of star_p, result into xp
                const double dpl = i > 0 ? (rp[i] - rp[i - 1]) / dx : (per ? (rp[0] - rp[L - 1]) / dx : 0.0);
                const double v = rp[i] + dt * (-lam * dpl);
                xp[i] = v < 0.0 ? 0.0 : v;
            }
            __syncthreads();
            for (int i = t; i < L; i += NT) rp[i] = xp[i];
        }
        __syncthreads();
        double m1 = 0.0;
        for (int i = t; i < L; i += NT) m1 += rp[i] + rm[i];
        m1 = block_sum(m1, red);
        const double sc = m0 / m1;
        for (int i = t; i < L; i += NT) { rp[i] *= sc; rm[i] *= sc; }
        __syncthreads();
    }
    for (int i = t; i < L; i += NT) {
        if (a.rho_p) a.rho_p[(size_t)sys * L + i] = rp[i];
        if (a.rho_m) a.rho_m[(size_t)sys * L + i] = rm[i];
    }
    for (int i = t; i < ntr; i += NT) {
        if (a.tracer_x) a.tracer_x[(size_t)sys * ntr + i] = trx[i];
        if (a.tracer_s) a.tracer_s[(size_t)sys * ntr + i] = trs[i];
    }
}

struct DevBuf {            // frees everything it allocated when it goes out of scope
    std::vector<void *> ptrs;
    ~DevBuf() { for (void *q : ptrs) (void)hipFree(q); }
    template <typename T> T *alloc(size_t n) {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
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

const char *pde_last_error(void) { return g_err.c_str(); }

int pde_solve_batch(const pde_params *p, int32_t n_systems, const double *beta, const double *rho_p0, const double *rho_m0,
                    const double *tracer_x0, const int8_t *tracer_s0, const double *rand_u, const double *rand_n,
                    double *rho_p, double *rho_m, double *m_series, double *var_series, double *v_eff_series,
                    double *D_eff_series, double *snapshots, double *m_snapshots, double *fft_re, double *fft_im,
                    double *tracer_x, int8_t *tracer_s, double *kernel_ms) {
    auto bad = [&](const char *m) { g_err = std::string("pde_solve_batch: ") + m; return PDE_ERR_ARG; };
    if (!p || !beta || !rho_p0 || !rho_m0 || n_systems < 1) return bad("null argument or n_systems < 1");
    if (p->L < 4 || p->L > PDE_MAX_L) return bad("L must be in [4, PDE_MAX_L]");
    if (p->nsteps < 0 || !(p->dt > 0.0) || !(p->xlim > 0.0)) return bad("nsteps >= 0, dt > 0, xlim > 0 required");
    if (p->snapshot_interval < 1) return bad("snapshot_interval must be >= 1");
    if (p->kernel_mode < 0 || p->kernel_mode > 2) return bad("kernel_mode must be 0, 1 or 2");
    if (p->n_tracers < 0 || (p->n_tracers > 0 && (!tracer_x0 || !tracer_s0 || p->window < 1))) return bad("tracers need initial positions, states and window >= 1");
    if ((rand_u == nullptr) != (rand_n == nullptr)) return bad("rand_u and rand_n come together");
    if (p->n_fft_modes < 0 || p->n_fft_modes > p->L / 2 + 1 || ((fft_re == nullptr) != (fft_im == nullptr))) return bad("bad fft request");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_err = "pde_solve_batch: no HIP device"; return PDE_ERR_NODEVICE; }
    if (p->device < 0 || p->device >= ndev) return bad("device ordinal out of range");
    if (hipSetDevice(p->device) != hipSuccess) { g_err = "hipSetDevice failed"; return PDE_ERR_HIP; }

    const int L = p->L, ntr = p->n_tracers, ns = p->nsteps + 1;
    const double dx = p->xlim / L;
    // ---- constant matrix: Thomas factorisation (and the Sherman-Morrison vector for periodic corners)
    const double av = p->gamma * p->dt / (dx * dx), bv = 1.0 + 2.0 * av;
    std::vector<double> lo(L, -av), di(L, bv), up(L, -av), fw(L, 0.0), finv(L), fz(L, 0.0);
    lo[0] = 0.0; up[L - 1] = 0.0;
    double sm_coef = 0.0, sm_denom = 1.0;
    if (!p->periodic) { up[0] = -2.0 * av; lo[L - 1] = -2.0 * av; }
    else {
        const double corner = -av, gam = -bv;
        di[0] = bv - gam; di[L - 1] = bv - corner * corner / gam;
        sm_coef = corner / gam;
    }
    std::vector<double> piv(L);
    piv[0] = di[0];
    for (int i = 1; i < L; ++i) { fw[i] = lo[i] / piv[i - 1]; piv[i] = di[i] - fw[i] * up[i - 1]; }
    for (int i = 0; i < L; ++i) finv[i] = 1.0 / piv[i];
    if (p->periodic) {                                         // A' z = u,  u = (gam, 0, ..., 0, corner)
        const double corner = -av, gam = -bv;
        std::vector<double> y(L, 0.0);
        y[0] = gam; y[L - 1] = corner;
        for (int i = 1; i < L; ++i) y[i] -= fw[i] * y[i - 1];
        fz[L - 1] = y[L - 1] * finv[L - 1];
        for (int i = L - 2; i >= 0; --i) fz[i] = (y[i] - up[i] * fz[i + 1]) * finv[i];
        sm_denom = 1.0 + fz[0] + sm_coef * fz[L - 1];
    }
    // ---- kernel taps (ref :84-93), normalised over the whole ring, cut where negligible
    std::vector<double> ktab(1, 1.0);
    int ktaps = 0;
    if (p->kernel_mode == 1) {
        std::vector<double> full(L);
        double sum = 0.0;
        for (int i = 0; i < L; ++i) { const double d = std::min(i, L - i) * dx / p->kernel_sigma; full[i] = std::exp(-0.5 * d * d); sum += full[i]; }
        ktaps = 0;
        for (int i = 0; i <= L / 2; ++i) if (full[i] >= 1e-17 * full[0]) ktaps = i;
        ktab.assign(ktaps + 1, 0.0);
        for (int i = 0; i <= ktaps; ++i) ktab[i] = full[i] / sum;
        if (L % 2 == 0 && ktaps == L / 2) ktab[ktaps] *= 0.5;  // the antipodal site is met from both sides of the sweep
    }
    std::vector<double> twc(L), tws(L);
    for (int j = 0; j < L; ++j) { const double ang = 6.283185307179586476925 * (double)j / (double)L; twc[j] = std::cos(ang); tws[j] = std::sin(ang); }

    DevBuf d;
    PdeArgs a{};
    a.p = *p; a.dx = dx; a.sm_coef = sm_coef; a.sm_denom = sm_denom; a.ktaps = ktaps;
    a.chunk = (L + NT - 1) / NT; a.n_snap = p->nsteps / p->snapshot_interval + 1;
    const size_t SL = (size_t)n_systems * L, SN = (size_t)n_systems * ns, ST = (size_t)n_systems * ntr;
#define UP(dst, src, n) do { a.dst = d.upload(src, n); if (!a.dst) { g_err = "pde_solve_batch: device upload failed (" #dst ")"; return PDE_ERR_HIP; } } while (0)
#define OUT(dst, host, n) do { if (host) { a.dst = d.alloc<std::remove_pointer<decltype(a.dst)>::type>(n); if (!a.dst) { g_err = "pde_solve_batch: device allocation failed (" #dst ")"; return PDE_ERR_HIP; } } } while (0)
    UP(beta, beta, (size_t)n_systems); UP(rho_p0, rho_p0, SL); UP(rho_m0, rho_m0, SL);
    UP(fw, fw.data(), (size_t)L); UP(finv, finv.data(), (size_t)L); UP(fu, up.data(), (size_t)L); UP(fz, fz.data(), (size_t)L);
    UP(ktab, ktab.data(), ktab.size()); UP(twc, twc.data(), (size_t)L); UP(tws, tws.data(), (size_t)L);
    if (ntr) {
        UP(tracer_x0, tracer_x0, ST); UP(tracer_s0, tracer_s0, ST);
        if (rand_u) { UP(rand_u, rand_u, SN * ntr); UP(rand_n, rand_n, SN * ntr); }
        a.hist = d.alloc<double>((size_t)n_systems * p->window * ntr);
        a.trx = d.alloc<double>(ST); a.trs = d.alloc<int8_t>(ST);
        if (!a.hist || !a.trx || !a.trs) { g_err = "pde_solve_batch: device allocation failed (tracers)"; return PDE_ERR_HIP; }
    }
    OUT(rho_p, rho_p, SL); OUT(rho_m, rho_m, SL); OUT(m_series, m_series, SN); OUT(var_series, var_series, SN);
    OUT(v_eff, v_eff_series, SN); OUT(D_eff, D_eff_series, SN);
    OUT(snapshots, snapshots, (size_t)n_systems * a.n_snap * L); OUT(m_snapshots, m_snapshots, (size_t)n_systems * a.n_snap * L);
    OUT(fft_re, fft_re, SN * p->n_fft_modes); OUT(fft_im, fft_im, SN * p->n_fft_modes);
    OUT(tracer_x, tracer_x, ST); OUT(tracer_s, tracer_s, ST);
#undef UP
#undef OUT
    size_t lds = (size_t)((5 * L + ((ktaps + 2) & ~1) + NT + 3) & ~3) * sizeof(double) + (size_t)2 * NT * sizeof(double4);
    if (lds > 160 * 1024) {                                    // beyond LDS: the fields live in global memory, the scans' scratch in LDS
        a.work_stride = (long long)((5 * (size_t)L + ((ktaps + 2) & ~1) + 3) & ~(size_t)3);
        a.work = d.alloc<double>((size_t)n_systems * (size_t)a.work_stride);
        if (!a.work) { g_err = "pde_solve_batch: device allocation failed (work)"; return PDE_ERR_HIP; }
        lds = (size_t)((NT + 3) & ~3) * sizeof(double) + (size_t)2 * NT * sizeof(double4);
    }
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(&pde_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        g_err = "pde_solve_batch: cannot raise the dynamic LDS limit"; return PDE_ERR_HIP;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { g_err = "hipEventCreate failed"; return PDE_ERR_HIP; }
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(pde_kernel, dim3((unsigned)n_systems), dim3(NT), lds, nullptr, a);
    (void)hipEventRecord(e1, nullptr);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipDeviceSynchronize();
    float ms = 0.f;
    if (err == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) { g_err = std::string("pde_kernel: ") + hipGetErrorString(err); return PDE_ERR_HIP; }
    if (kernel_ms) *kernel_ms = ms;
#define DOWN(host, dev, n) do { if (host && hipMemcpy(host, a.dev, (n), hipMemcpyDeviceToHost) != hipSuccess) { g_err = "pde_solve_batch: download failed (" #dev ")"; return PDE_ERR_HIP; } } while (0)
    DOWN(rho_p, rho_p, SL * 8); DOWN(rho_m, rho_m, SL * 8); DOWN(m_series, m_series, SN * 8); DOWN(var_series, var_series, SN * 8);
    DOWN(v_eff_series, v_eff, SN * 8); DOWN(D_eff_series, D_eff, SN * 8);
    DOWN(snapshots, snapshots, (size_t)n_systems * a.n_snap * L * 8); DOWN(m_snapshots, m_snapshots, (size_t)n_systems * a.n_snap * L * 8);
    DOWN(fft_re, fft_re, SN * p->n_fft_modes * 8); DOWN(fft_im, fft_im, SN * p->n_fft_modes * 8);
    DOWN(tracer_x, tracer_x, ST * 8); DOWN(tracer_s, tracer_s, ST);
#undef DOWN
    return PDE_OK;
}

}  // extern "C"
