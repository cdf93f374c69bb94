/* pde.h -- C ABI of the MI355X-native hydrodynamic-limit solver (part of libaps_hip.so).
 *
 * Replaces the time loop of the reference's IMEXPDE (IMEX_PDE_solver_class.py): `solve()` :236-290 with `step()`
 * :190-233 (implicit diffusion, upwind advection, Curie-Weiss reaction, clip, mass renormalisation), the
 * per-step observables :243-255 and the Euler-Maruyama tracer particles :257-287 -- for a BATCH of independent
 * systems (one workgroup each) that share every parameter except beta, the parameter the reference's sweep
 * drivers vary (IMEX_PDE_solver_run_sweep.py:26-40).  Plain C types, caller-allocated host buffers.
 * All functions return 0 on success and a negative code on failure; pde_last_error() gives the text.
 */
#ifndef PDE_H
#define PDE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDE_OK 0
#define PDE_ERR_ARG (-1)
#define PDE_ERR_HIP (-2)
#define PDE_ERR_NODEVICE (-4)

#define PDE_MAX_L (1 << 22)     /* one workgroup per system; its fields sit in LDS up to L ~ 3000, in global memory beyond */
#define PDE_LDS_L 3072           /* largest L whose fields fit one workgroup's LDS (plain local kernel) */

/* Mirrors the reference constructor keywords (ref :13-61). */
typedef struct pde_params {
    int32_t L;                  /* grid points, 4 <= L <= PDE_MAX_L                      ref :29 */
    int32_t nsteps;             /* int(T / dt)                                            ref :35 */
    int32_t periodic;           /* bc: 1 = "periodic", 0 = "neumann"                      ref :41 */
    int32_t anchored_minus;     /* active_model: 1 = "anchored_minus", 0 = "bidirectional" ref :42 */
    int32_t kernel_mode;        /* 0 local ratio, 1 periodic Gaussian kernel, 2 global mean  ref :156-168 */
    int32_t snapshot_interval;  /*                                                        ref :47 */
    int32_t n_tracers;          /*                                                        ref :125 */
    int32_t window;             /* int(0.05 / dt): steps of the v_eff / D_eff window      ref :238-239 */
    int32_t n_fft_modes;        /* lowest rfft modes of the total density recorded per step (0: none)  ref :249-251 */
    int32_t device;
    int32_t reserved[2];
    double xlim, dt, gamma, lam, kernel_sigma;
    uint64_t seed;              /* Philox key of the tracer noise when no random numbers are supplied */
} pde_params;

const char *pde_last_error(void);

/* Runs n_systems systems from their initial states to step nsteps.
 *   beta[n_systems]                      Curie-Weiss inverse temperature of each system          (ref :39)
 *   rho_p0, rho_m0 [n_systems][L]        initial densities (ref initialize() :96-121)
 *   tracer_x0 [n_systems][n_tracers], tracer_s0 (+1/-1)  initial tracers (ref :125-129); may be NULL iff n_tracers == 0
 *   rand_u, rand_n [n_systems][nsteps+1][n_tracers]      optional: the uniform and normal numbers the reference
 *                                        would draw (np.random.rand / randn, ref :268, :273); NULL: Philox on the device
 * Outputs (any may be NULL):
 *   rho_p, rho_m [n_systems][L]; m_series, var_series, v_eff_series, D_eff_series [n_systems][nsteps+1] (NaN where
 *   the reference leaves NaN); snapshots, m_snapshots [n_systems][nsteps/snapshot_interval + 1][L];
 *   fft_re, fft_im [n_systems][nsteps+1][n_fft_modes]; tracer_x (unwrapped) [n_systems][n_tracers], tracer_s. */
int pde_solve_batch(const pde_params *p, int32_t n_systems, const double *beta, const double *rho_p0, const double *rho_m0,
                    const double *tracer_x0, const int8_t *tracer_s0, const double *rand_u, const double *rand_n,
                    double *rho_p, double *rho_m, double *m_series, double *var_series, double *v_eff_series,
                    double *D_eff_series, double *snapshots, double *m_snapshots, double *fft_re, double *fft_im,
                    double *tracer_x, int8_t *tracer_s, double *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* PDE_H */
