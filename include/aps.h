/* aps.h -- C ABI of the MI355X-native active-particle stepper (libaps_hip.so).
 *
 * This is the drop-in boundary for the hot path of the reference's ParticleSystem
 * (PARTICLE_solver_CLASS.py): plain C types, caller-allocated host buffers, an opaque handle.
 * The Python face `ParticleSystem` (package file particle_system.py) binds it with ctypes; the
 * binding a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Each entry point names the reference code it replaces (file = PARTICLE_solver_CLASS.py).
 * All functions return 0 on success and a negative code on failure; aps_last_error() gives the text.
 * A handle is not thread-safe; different handles may be used from different threads.
 */
#ifndef APS_H
#define APS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APS_OK 0
#define APS_ERR_ARG (-1)      /* bad argument / unsupported parameter combination */
#define APS_ERR_HIP (-2)      /* a HIP runtime call failed (text in aps_last_error) */
#define APS_ERR_STATE (-3)    /* call not valid in the current state (e.g. no state uploaded) */
#define APS_ERR_NODEVICE (-4) /* no usable GPU */

typedef struct aps_handle aps_handle;

/* Mirrors the reference constructor keywords (ref :14-138) after the rate scaling of :45-50.
 * One handle carries n_ensembles independent systems of the same shape (BASELINE config 4). */
typedef struct aps_params {
    int32_t L;                  /* lattice sites, 2 <= L <= 2^25                     ref :41 */
    int32_t K;                  /* site capacity, 1..32                              ref :43 */
    int32_t periodic;           /* 0 = reflecting walls, 1 = torus                   ref :81 */
    int32_t minus_anchor;       /*                                                   ref :83 */
    int32_t immobilize;         /* immobilize_when_anchored                          ref :82 */
    int32_t suppress_flip;      /* suppress_flip_when_bound                          ref :54 */
    int32_t crowding;           /* crowding_suppresses_rates                         ref :55 */
    int32_t n_ensembles;        /* >= 1 */
    int64_t n_particles;        /* particles per ensemble (capacity of the state arrays) */
    double sigma_grid;          /* local_kernel_sigma / dx; <= 0 selects the global mean   ref :84, :219 */
    double rate_diffusion;      /* scaled                                            ref :45-50 */
    double rate_active;
    double k_on, k_off, k_exit; /*                                                   ref :52-56 */
    double dt;                  /* fixed step of the synchronous scheme */
    uint64_t seed;              /* Philox key */
    const double *beta;         /* [n_ensembles]                                     ref :51 */
    const uint8_t *anchor_mask; /* [L] is_anchor_site, or NULL for none              ref :88-104 */
    int32_t device;             /* HIP device ordinal */
    int32_t rank, world;        /* shard of this handle, world = 1: everything.  TILES: the rank's contiguous range of site tiles
                                   (aps_owned_sites); LATTICE / PAIRS: its block of particle slots */
    int32_t sort_by_site;       /* 1: keep particles ordered by site internally (tile culling) */
    int32_t ensemble_base;      /* Philox counter word 3 of local ensemble e is ensemble_base + e */
    int32_t method;             /* APS_METHOD_*: which formulation of the mean field the stepper uses */
    int32_t fp32;               /* 0: exact binary64 field (weights on the grid 2^-q, q = 51 - bits of the largest sum).
                                   1: "float32" field of BASELINE config 5: the same construction on the coarser grid on which
                                   every sum fits 32 bits (q = 29 - bits); W, S and the table are int32 in the stepping kernel
                                   (APS_METHOD_TILES).  Still exact integer sums -- independent of summation order, tiling and GPU
                                   count -- but a different, coarser weight table: |m - m_reference| ~ 1e-6 instead of 2e-11 */
    int32_t halo_interval;      /* site-sharded TILES handles: steps per halo exchange.  k > 1 keeps (k - 1) * reach tiles of the
                                   neighbours' state as a ghost zone that this rank steps redundantly (a range shrinking by
                                   `reach` tiles per step), so that one larger message every k steps replaces k small ones;
                                   same bits for every k.  0: chosen by the library (aps_halo_info reports it) */
} aps_params;

/* Two formulations of compute_local_m_field (ref :216-246); both give the same bits (exact weight grid):
 *   PAIRS    all-pairs tile kernel: every step each particle sums w(d_ij) over all particles in reach
 *   LATTICE  the reference's own histogram -> smoothing -> gather, with the smoothed histograms
 *            tot_conv = W(x), s_conv = S(x) (ref :224-238) kept on the L sites and updated incrementally by the
 *            accepted events of each step; site occupancy (ref :248-252) likewise
 *   TILES    the LATTICE formulation with a site-centric state (one word per particle slot of a site) and the whole
 *            step -- field update, rates, draws, exclusion, state update -- in ONE kernel over site tiles; same bits
 *   AUTO     world = 1: TILES (LATTICE once the binary64 state exceeds 512 MB, the streaming regime); world > 1: LATTICE with
 *            particle-index shards -- site-range shards are chosen by asking for TILES (bench.py does, and falls back when the
 *            table's reach does not fit the ranks' ranges); PAIRS if the deposit lists would exceed 16 GB */
#define APS_METHOD_AUTO 0
#define APS_METHOD_PAIRS 1
#define APS_METHOD_LATTICE 2
#define APS_METHOD_TILES 3

int aps_device_count(void);
const char *aps_last_error(const aps_handle *h);   /* h may be NULL: error of the last failed aps_create */

/* replaces ParticleSystem.__init__ (ref :14-138): builds the weight table, allocates device state */
int aps_create(const aps_params *p, aps_handle **out);
void aps_destroy(aps_handle *h);

/* Run all subsequent launches and copies on a caller-owned hipStream_t, taken literally: NULL means the
 * legacy default stream (which is what torch.cuda.current_stream() is unless a side stream is active).
 * Until this is called the handle uses a private non-blocking stream. */
int aps_set_stream(aps_handle *h, void *hip_stream);

/* State exchange in ORIGINAL particle order (the order init_particles returned, ref :191-195).
 * sigma is +1/-1, bound and alive are 0/1; alive may be NULL (= all alive). n <= n_particles. */
int aps_set_state(aps_handle *h, int32_t ensemble, const int32_t *pos, const int8_t *sigma,
                  const uint8_t *bound, const uint8_t *alive, int64_t n);
int aps_get_state(aps_handle *h, int32_t ensemble, int32_t *pos, int8_t *sigma, uint8_t *bound,
                  uint8_t *alive, int64_t n);

/* Parity hook for the all-pairs kernel: per particle S = sum sigma_j w(d_ij), W = sum w(d_ij)
 * (ref :216-246 in all-pairs form) and occ4 = occupancy of {own, forward, left, right} target site
 * (ref :294-301).  Arrays are [n] / [n] / [4n] in original order. */
int aps_pair_accumulate(aps_handle *h, int32_t ensemble, double *S, double *W, int32_t *occ4, int64_t n);

/* Parity hook for the lattice formulation: the same quantities as aps_pair_accumulate, read from the
 * incrementally maintained lattice arrays at the particles' sites (m_field[pos], occ_total[target], ref :261, :294-301). */
int aps_lattice_accumulate(aps_handle *h, int32_t ensemble, double *S, double *W, int32_t *occ4, int64_t n);

/* The maintained lattice arrays themselves: W = tot_conv, S = s_conv (ref :224-238, unnormalised taps on the weight
 * grid) and occ = occ_total (ref :248-252), each [L]; any pointer may be NULL.  LATTICE handles only. */
int aps_get_lattice(aps_handle *h, int32_t ensemble, double *W, double *S, int32_t *occ);

/* APS_METHOD_PAIRS, _LATTICE or _TILES: what AUTO resolved to. */
int aps_method(aps_handle *h);

/* replaces the body of the `while t < T` loop (ref :511-516): nsteps synchronous steps of dt.
 * Sharded handles (world > 1) need aps_comm_init first (or the caller drives propose/exchange/commit). */
int aps_step(aps_handle *h, int64_t nsteps);

/* The two halves of one step, for callers that exchange proposals between ranks themselves:
 * aps_propose fills this rank's block of the proposal buffer, aps_commit applies ALL blocks. */
int aps_propose(aps_handle *h);
int aps_commit(aps_handle *h);
/* Device address/size of the whole proposal buffer and of this rank's block inside it. */
int aps_exchange_buffer(aps_handle *h, void **dev_ptr, int64_t *total_bytes, int64_t *my_offset,
                        int64_t *my_bytes);

/* Use caller-owned device memory (>= total_bytes of aps_exchange_buffer) as the proposal buffer, e.g. a
 * torch tensor that torch.distributed all-gathers in place.  NULL returns to the internal buffer. */
int aps_bind_exchange_buffer(aps_handle *h, void *dev_ptr, int64_t nbytes);

/* In-library data path for sharded handles: rank 0 obtains a 128-byte RCCL unique id, the caller broadcasts
 * it to all ranks (any transport), every rank calls aps_comm_init; afterwards aps_step() runs
 * propose -> ncclAllGather (in place, 1 byte per particle, on the handle's stream) -> commit per step. */
int aps_comm_unique_id(uint8_t *out128);
int aps_comm_init(aps_handle *h, const uint8_t *id128);
/* APS_METHOD_TILES handles shard by SITE RANGE instead: rank r steps the tiles of sites [lo, hi) (aps_owned_sites) and,
 * after each step's kernel, exchanges with its two neighbour ranks the three boundary sites of cells, two of {W, S} and the
 * deposit lists within the table's reach (ncclSend / ncclRecv inside aps_step once aps_comm_init was called).  A caller
 * that moves the halo itself calls aps_propose (the kernel), transfers, then aps_commit; aps_halo_copy is that transfer
 * between two handles living on ONE device (tests, single-process multi-handle runs).  On such handles aps_get_state
 * reports alive = 2 for particles that currently sit on another rank's sites, the lattice arrays are valid on the own sites,
 * and the scalar observables / exit log cover the own particles: the caller adds them up over the ranks. */
int aps_owned_sites(aps_handle *h, int32_t *lo, int32_t *hi);
int aps_halo_copy(aps_handle *dst, aps_handle *src_neighbour);
/* The same halo through host memory, for any transport: aps_halo_pack copies this rank's first (side 0, for the left
 * neighbour) or last (side 1, for the right neighbour) block into `host` (host = NULL: only the size), aps_halo_unpack
 * stores a received block (from_side 0: the RIGHT neighbour's first block, 1: the LEFT neighbour's last block). */
int aps_halo_pack(aps_handle *h, int32_t side, uint8_t *host, int64_t cap, int64_t *nbytes);
int aps_halo_unpack(aps_handle *h, int32_t from_side, const uint8_t *host, int64_t nbytes);
/* Steps per halo exchange of this handle (aps_params.halo_interval, or the library's choice), how many steps it has taken
 * since the last one, and whether the exchange is due NOW, i.e. between the aps_propose just made and its aps_commit
 * (due = 1 exactly when age + 1 == interval; a handle that is not site-sharded reports interval 0, due 0).  A caller that
 * moves the halo itself transfers only when due; aps_commit refuses a due step whose blocks have not all arrived, and
 * aps_halo_copy / pack / unpack refuse a step that is not due.  With interval k the blocks hold (k - 1) * reach whole
 * tiles of cells and {W, S} (+ three / two sites) and k * reach tiles of deposit lists per side. */
int aps_halo_info(aps_handle *h, int32_t *interval, int32_t *age, int32_t *due);
/* Byte counts of the four blocks: send_bytes[0] / [1] = this rank's first / last block (0: no neighbour on that side),
 * recv_bytes[0] = the RIGHT neighbour's first block, recv_bytes[1] = the LEFT neighbour's last block. */
int aps_halo_sizes(aps_handle *h, int64_t send_bytes[2], int64_t recv_bytes[2]);

/* The halo of site-sharded TILES handles by PEER STORES, the preferred transport inside aps_step: every rank exports its landing
 * buffers (aps_ipc_export: device memory + a HIP IPC handle, described by a 256-byte blob that the caller hands to the two
 * neighbour ranks by any means), maps its neighbours' (aps_ipc_connect; left / right blob, NULL where a reflecting wall is the
 * neighbour; handles of one process are connected by address) and from then on aps_step, after the kernel of a due step,
 * launches ONE push kernel that stores this rank's two packed blocks straight into the neighbours' landing buffers (over xGMI
 * between GPUs) followed by one arrival word each, and ONE pull kernel that waits (bounded: 20 s, APS_HALO_TIMEOUT_MS) for the
 * neighbours' arrival words and unpacks -- no RCCL kernel, no host call, no host synchronisation per exchange.  Landing buffers
 * are double buffered by exchange parity.  All ranks must call aps_step with the same step counts; a halo that does not arrive
 * makes aps_step return APS_ERR_STATE.  aps_exchange_kind: 0 the caller moves the halo, 1 ncclSend / ncclRecv, 2 peer stores. */
#define APS_IPC_BLOB_BYTES 256
int aps_ipc_export(aps_handle *h, uint8_t *blob256);
int aps_ipc_connect(aps_handle *h, const uint8_t *left_blob256, const uint8_t *right_blob256);
int aps_exchange_kind(aps_handle *h);

/* Number of ranks the communicator of this handle actually spans (ncclCommCount): what a bench line reports as
 * evidence that the exchange ran between that many processes. */
int aps_comm_ranks(aps_handle *h, int32_t *nranks);
/* nbytes through the calls the halo exchange makes (ncclGroupStart, ncclSend, ncclRecv, ncclGroupEnd on the handle's
 * stream) from this rank to itself, compared on the host: the transport smoke test a one-GPU box can run. */
int aps_comm_selftest(aps_handle *h, int64_t nbytes);

/* replaces the observation block (ref :517-536): site histograms and the m-field on all L sites. */
int aps_observe(aps_handle *h, int32_t ensemble, int64_t *counts_p, int64_t *counts_m, double *m_field);

/* Driver-side observables on the device (the sums behind compute_v_eff_and_window, compute_rho_eff,
 * compute_blocking_probability, compute_mean_magnetizatoin, compute_D_eff_active of
 * PARTICLE_solver_BIOLOGY_EXCLUSION_sweep_beta.py:123-229, :316-319, :500-525), all exact integers over the live
 * particles of one ensemble: out11 = { n, sum sigma, sum pos, #(pos >= x_wall), max pos (-1 if none),
 * #(range_lo <= pos <= range_hi), #(plus particles on sites < L-1), of those: blocked by the right neighbour,
 * sum d, sum d^2, #d } with d = pos - reference pos of the same particle (aps_mark_reference; zeros if none marked).
 * block_table[(K+1)*(K+1)], indexed [plus count * (K+1) + minus count] of the right neighbour site, says whether
 * that occupancy blocks (NULL: any particle blocks). */
int aps_mark_reference(aps_handle *h, int32_t ensemble);
int aps_observe_scalars(aps_handle *h, int32_t ensemble, int32_t x_wall, int32_t range_lo, int32_t range_hi,
                        const uint8_t *block_table, int64_t *out11);

/* The same for ALL ensembles of the handle in one pass (two launches, one download): out11 is [n_ensembles][11];
 * range_lo_hi [n_ensembles][2] gives each ensemble's own site range (NULL: empty range). */
int aps_observe_scalars_all(aps_handle *h, int32_t x_wall, const int32_t *range_lo_hi, const uint8_t *block_table,
                            int64_t *out11);

/* Structure observables on the device (the inputs of extract_structure_observables_from_out,
 * PARTICLE_solver_BIOLOGY_local_structure.py:55-103, taken from the current state of one ensemble instead of from the M x L
 * arrays of run(): ref :527-535): out[0] = live particles n, out[1] = sum over sites of (particles on the site)^2,
 * out[2], out[3] = sum and sum of squares over the L sites of the local magnetisation m(x) (ref :216-246),
 * out[4 + 2k], out[5 + 2k] = real and imaginary part of sum_x count(x) exp(-2 pi i k x / L), k = 0 .. k_max - 1
 * (np.fft.fft(total)[k] times n dx).  out holds 4 + 2 k_max doubles; 1 <= k_max <= L. */
int aps_observe_structure(aps_handle *h, int32_t ensemble, int32_t k_max, double *out);

/* Coarse-grained histograms on the device: live plus / minus particles per bin of ceil(L / nbins) consecutive sites (the grid
 * of the hydrodynamic-limit PDE when particle and PDE densities are compared on the same domain; rho_+- of ref :205-213
 * summed over a bin).  plus, minus: [nbins] exact counts. */
int aps_observe_bins(aps_handle *h, int32_t ensemble, int32_t nbins, int64_t *plus, int64_t *minus);

/* m-field for a caller-supplied histogram: compute_local_m_field(counts_p, counts_m) (ref :216-246) */
int aps_field_from_counts(aps_handle *h, int32_t ensemble, const int64_t *counts_p, const int64_t *counts_m,
                          double *m_field);

/* replaces the rate section of step_gillespie (ref :254-352) for CALLER-supplied arrays (n particles, the
 * m-field and the site histograms the reference passes in): out9n[c*n + i], c = diffusion, active, flip, bind,
 * unbind, exit, left, right, total.  The event choice itself (ref :358-448) stays with the caller's generator. */
int aps_rates_from_field(aps_handle *h, int32_t ensemble, const int32_t *pos, const int8_t *sigma, const uint8_t *bound,
                         int64_t n, const double *m_field, const int64_t *counts_p, const int64_t *counts_m,
                         double *out9n);

int aps_time(aps_handle *h, double *t, int64_t *step_index);

/* Exit log (ref :424-436): rows of (time, position, particle index), ordered by (time, index). */
int aps_get_exits(aps_handle *h, int32_t ensemble, double *rows3, int64_t cap_rows, int64_t *n_rows);

/* Weight table actually used by the kernels (for parity tests): entries, and the grid exponent q. */
int aps_get_table(aps_handle *h, double *out, int32_t cap, int32_t *tlen, int32_t *q);

/* Re-establish the site-sorted internal order (no effect on results; speeds up tile culling). */
int aps_resort(aps_handle *h);

/* Measurement: run nsteps steps (launched one by one, no graph replay) with HIP events around every launch of the
 * field kernel alone -- pair_accumulate (PAIRS) or field_update (LATTICE) -- on the handle's stream; returns
 * their summed duration, the number of launches and the work done: pair evaluations (PAIRS) or deposits (LATTICE). */
int aps_step_timed(aps_handle *h, int64_t nsteps, double *kernel_ms, int64_t *launches, double *work);

/* How the last aps_step call ran: steps replayed from captured hipGraphs (runs of 32, 16, 8, 4, 2 and 1 steps, for either parity of the first step) and steps launched
 * kernel by kernel. */
int aps_step_info(aps_handle *h, int64_t *graph_steps, int64_t *single_steps);

/* A caller's flip_rate_fn (ref :59-62, applied elementwise to (sigma, m_field[pos]) at :261-262) on the device: the host
 * evaluates the callable on a uniform grid of m in [-1, 1] and hands over table[2][n + 1] -- row 0: sigma = +1, row 1: sigma = -1,
 * column i: m = -1 + 2 i / n -- which every rate evaluation of this handle then interpolates linearly instead of evaluating
 * exp(-beta sigma m) (error <= max |d2 rate / dm2| (2 / n)^2 / 8; exact for rates linear in m between grid points).  NULL
 * returns to the Curie-Weiss rate.  Rates must be finite and >= 0. */
int aps_set_flip_table(aps_handle *h, const double *table, int32_t n);

/* The field update as an exact convolution (csrc/ntt_conv.hpp): TILES handles (reflecting walls or torus) with one rank and a weight
 * table beyond LDS (BASELINE config 5) add a step's deposits to W, S by a number-theoretic transform of length
 * 2^log2_m >= L + 2 reach instead of gathering deposits x taps table entries -- exact integers, same bits as the sweep: mod
 * P0 = 15 * 2^27 + 1 for the 32-bit field (fp32), mod P0 and P1 = 27 * 2^26 + 1 with the Chinese remainder for the binary64 field.
 * on: whether this handle does (APS_NTT=0 keeps the sweep, APS_NTT=1 takes the convolution for tables that fit
 * LDS as well); prof_ms / prof_launches: summed duration and number of the convolution's launches in the last aps_step_profile. */
int aps_ntt_info(aps_handle *h, int32_t *on, int32_t *log2_m, double *prof_ms, int64_t *prof_launches);
/* Launches per convolution: 3 when the transform has at least two 128 x 128 slabs (log2_m >= 15: sweep along the slab index, ONE
 * launch for everything inside a slab -- two sweeps, the product with the table's spectrum, two sweeps back -- in LDS, sweep back;
 * APS_NTT_FUSED=0 keeps them apart), 5 otherwise (3 at log2_m = 14, a single slab); 0 when the handle does not take the convolution. */
int aps_ntt_launches(aps_handle *h);

/* Geometry of a TILES handle: sites per workgroup frame (64 RS), sites a tile owns, number of tiles, whether the weight table
 * sits in LDS (else: windows of it, tile_step's windowed sweep). */
int aps_tiles_info(aps_handle *h, int32_t *frame_sites, int32_t *owned_sites, int32_t *n_tiles, int32_t *table_in_lds);

/* Resident loop (TILES, one rank, weight table in LDS, local field, the whole grid of tiles resident on the device
 * at once -- BASELINE config 2): aps_step then runs its steps inside ONE launch; every tile keeps its state on chip and
 * exchanges only deposit lists and boundary cells with its neighbours between two steps.  Same bits as one launch per step
 * (replaces the loop of ParticleSystem.run, PARTICLE_solver_CLASS.py:511-516, like aps_step itself).  on = 0 keeps a handle
 * on one launch per step; default 1 (used when eligible).  aps_loop_info: steps of the last aps_step call taken inside the
 * loop; state 1 usable, 0 not eligible, -1 a call gave up (the steps were repeated the ordinary way; not tried again),
 * -2 not looked at yet; `why` receives the reason when the loop is not used. */
int aps_set_resident_loop(aps_handle *h, int32_t on);
int aps_loop_info(aps_handle *h, int64_t *loop_steps, int32_t *state, char *why, int32_t why_len);
/* Measurement: aps_step(nsteps) with HIP start/stop events attached to the resident loop's own dispatch; returns that
 * launch's duration and the steps it took (0 / 0.0 when the call did not use the loop). */
int aps_step_loop_timed(aps_handle *h, int64_t nsteps, double *kernel_ms, int64_t *loop_steps);

/* Measurement: bytes read + bytes written per second of a plain 16-byte-per-lane copy kernel over nbytes (>= 1 MiB; use
 * >= 1 GiB to get past the caches) on the handle's device -- the streaming ceiling of THIS box, quoted beside the spec. */
int aps_copy_bandwidth(aps_handle *h, int64_t nbytes, int32_t reps, double *gbytes_per_s);

/* What the instrument itself reads: mean elapsed time between two events recorded back to back (nothing in between)
 * on the handle's stream.  Event-bracketed durations of microsecond kernels carry this offset. */
int aps_event_overhead(aps_handle *h, int32_t reps, double *ms_per_pair);

/* The same for every kernel of the step: ms8[k] / launches8[k] summed over nsteps, k = pair_accumulate, propose,
 * claim, apply, plan_tiles, propose_lattice, field_update, tile_step. */
int aps_step_profile(aps_handle *h, int64_t nsteps, double *ms8, int64_t *launches8);

#ifdef __cplusplus
}
#endif
#endif /* APS_H */
