/* gillespie.h -- C ABI of the device-resident EXACT event loop (part of libaps_hip.so).
 *
 * Replaces ParticleSystem.run's `while t < T` loop as the reference wrote it (PARTICLE_solver_CLASS.py:511-538):
 * one Gillespie event per iteration -- compute_local_m_field (:216-246, kept incrementally: an event changes the
 * smoothed histograms only within the kernel's reach of one or two sites), the rate section of step_gillespie
 * (:254-352), the waiting time / particle / event choice (:358-367) and the state update (:371-446) -- for a BATCH
 * of independent systems, one persistent workgroup each, the whole system in LDS.  This is the shape of the
 * reference's sweep drivers (a serial `for beta: for run:` double loop, ..._sweep_beta.py:75-95, :895-897).
 * Plain C types, caller-allocated host buffers; 0 on success, negative on failure, gil_last_error() has the text.
 */
#ifndef GILLESPIE_H
#define GILLESPIE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIL_OK 0
#define GIL_ERR_ARG (-1)
#define GIL_ERR_HIP (-2)
#define GIL_ERR_NODEVICE (-4)

#define GIL_MAX_L 4096          /* one system lives in one workgroup's LDS */
#define GIL_MAX_N 2048
#define GIL_NSCALARS 12

/* Mirrors the reference constructor keywords (ref :14-138) after the rate scaling of :45-50. */
typedef struct gil_params {
    int32_t L, K, periodic, minus_anchor, immobilize, suppress_flip, crowding;   /* as in aps_params */
    int32_t n_systems;          /* independent systems = workgroups */
    int32_t n_cap;              /* particle slots per system (<= GIL_MAX_N) */
    int32_t n_obs;              /* number of observation times (ref :461) */
    int32_t device;
    int32_t x_wall;             /* first site counted as "at the right wall" in the scalar sums */
    int32_t ref_obs;            /* observation index whose positions are the origin of the displacement sums (-1: none) */
    int32_t flip_n;             /* intervals of flip_table (0: none) */
    double sigma_grid, rate_diffusion, rate_active, k_on, k_off, k_exit;
    double T;                   /* the loop ends when t > T or after the last observation time (ref :511-538) */
    uint64_t seed;              /* Philox key; counter = (event index, system index) */
    int64_t max_events;         /* safety bound per system (and number of rows of `uniforms` when supplied) */
    const double *beta;         /* [n_systems] */
    const uint8_t *anchor_mask; /* [L] or NULL */
    const double *times_obs;    /* [n_obs], increasing, times_obs[0] is recorded before the first event */
    const int32_t *front_lo;    /* [L] or NULL: lowest site of the front window when the right-most particle sits at site s */
    const uint8_t *block_table; /* [(K+1)*(K+1)] or NULL: does a right neighbour with (plus, minus) particles block? */
    const double *flip_table;   /* [2][flip_n + 1] or NULL: a caller's flip_rate_fn tabulated over m in [-1, 1] (row 0: sigma = +1, row 1: -1),
                                   interpolated linearly instead of exp(-beta sigma m) (ref :59-62, :261-262; aps_set_flip_table) */
} gil_params;

const char *gil_last_error(void);

/* Inputs per system s: n0[s] particles pos0/sigma0/bound0[s*n_cap ...] (bound0 may be NULL = unbound).
 * uniforms: optional [n_systems][max_events][4] numbers in [0,1) used instead of Philox for (waiting time,
 * particle, event, left/right) -- the parity tests feed the same numbers to the CPU oracle.
 * Outputs (any may be NULL): per observation k and system s
 *   pos_obs, sigma_obs [s][k][n_cap], flags_obs (bit 0 bound, bit 1 alive): the state after the event that crossed
 *   times_obs[k] (ref :517-524);   scalars_obs [s][k][GIL_NSCALARS] = { n, sum sigma, sum pos, #(pos >= x_wall),
 *   max pos, #(front window), #plus movers, #blocked movers, sum d, sum d^2, #d, events so far };
 * n_recorded[s], n_events[s], t_final[s]; exits [s][n_cap][3] = (time, site, particle) with n_exits[s] (ref :424-436). */
int gil_run_batch(const gil_params *p, const int32_t *n0, const int32_t *pos0, const int8_t *sigma0, const uint8_t *bound0,
                  const double *uniforms, int32_t *pos_obs, int8_t *sigma_obs, uint8_t *flags_obs, int64_t *scalars_obs,
                  int32_t *n_recorded, int64_t *n_events, double *t_final, double *exits, int32_t *n_exits,
                  double *kernel_ms);

/* The same loop for ONE system too large for a workgroup's LDS (n_systems must be 1; L <= 2^25, L*K <= 2^27,
 * n_cap <= 2^20): the BASELINE size N = 1e5, where the reference recomputes the whole field and all N rates before
 * every event (PARTICLE_solver_CLASS.py:512-513; 0.8 events/s measured).  One persistent workgroup of 1024 threads,
 * state in global memory, a site -> particle map and two-level rate sums so that an event touches only what it
 * changes.  Arguments as in gil_run_batch for a single system (no scalar sums); uniforms [max_events][4] optional. */
int gil_run_large(const gil_params *p, int32_t n0, const int32_t *pos0, const int8_t *sigma0, const uint8_t *bound0,
                  const double *uniforms, int32_t *pos_obs, int8_t *sigma_obs, uint8_t *flags_obs, int32_t *n_recorded,
                  int64_t *n_events, double *t_final, double *exits, int32_t *n_exits, double *kernel_ms);
const char *gil_large_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GILLESPIE_H */
