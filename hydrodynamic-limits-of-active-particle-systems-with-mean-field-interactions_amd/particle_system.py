"""`ParticleSystem` -- the reference's solver class (PARTICLE_solver_CLASS.py:13) re-hosted on the
MI355X stepper.  Same constructor keywords (ref :16-39), same attributes, same `run()` result
dictionary (ref :542-557); the time loop runs in HIP kernels behind the C ABI of include/aps.h.

Which dynamics `run()` integrates (`mode`):
  * default (no `mode`, no `dt` keyword -- an unchanged reference driver): `"gillespie_gpu"`, the reference's own exact
    one-event-per-iteration loop (ref :511-516, :358-362) resident on the GPU; statistically identical to the reference
    (fixture G4, no bias allowance).  With a custom `flip_rate_fn` the default is `"gillespie"` (below);
  * `mode="sync"` (also chosen when the caller passes `dt`): time advances in fixed steps `dt` (synchronous scheme, first
    order in `dt`: -2 % drift bias at dt = 0.0125; `dt` defaults to 0.1 / (largest possible total rate of one particle)) --
    the high-throughput stepper behind aps_step that BASELINE's particle-steps/s metric is quoted on;
  * `mode="gillespie"` (or calling `step_gillespie` yourself): the exact loop with the m-field and the rate vectors of every
    event computed on the GPU and the event drawn from `rng` in the reference's call order: reproduces the reference's
    seeded trajectories bit for bit, a custom `flip_rate_fn` (a Python callable) included.
Randomness of the device modes comes from Philox4x32-10 keyed by `seed` (default: drawn from `rng.random()` right after the
initial condition, so a seeded `rng` still makes the whole run reproducible).  `m_local_list[k]` is the field of the observed
state (the reference stores the field from before the last event, ref :525).
There is no CPU fallback: without libaps_hip.so or without a GPU, construction of the stepper raises.
"""
from __future__ import annotations

import math

import numpy as np

from . import capi


FLIP_TABLE_N = 1 << 16


def tabulate_flip_rate(fn, n=FLIP_TABLE_N):
    """A caller's flip_rate_fn(sigma, m) (reference :59-62: applied elementwise to the arrays sigma and m_field[pos], :261-262) on
    the uniform grid m_i = -1 + 2 i / n for sigma = +1 and -1: the [2][n + 1] table the device modes interpolate linearly.
    Error of the interpolated rate <= max |d2 rate / dm2| (2 / n)^2 / 8 (1.2e-10 x the second derivative at n = 2^16; a jump of
    the callable is smeared over one cell of width 3e-5).  Raises ValueError when the callable is not elementwise, or returns
    something that is not a finite rate >= 0 -- such a callable runs with mode="gillespie" (host draws) only."""
    m = np.linspace(-1.0, 1.0, n + 1)
    rows = []
    rng = np.random.default_rng(12345)
    pick = rng.choice(n + 1, size=257, replace=False)
    for sg in (1, -1):
        sig = np.full(n + 1, sg, dtype=np.int8)
        try:
            r = np.asarray(fn(sig, m), dtype=float)
            part = np.asarray(fn(sig[pick], m[pick]), dtype=float)
        except Exception as exc:                                # noqa: BLE001
            raise ValueError(f"flip_rate_fn cannot be tabulated for the device modes ({exc!r}); use mode='gillespie'") from exc
        if r.shape != m.shape or part.shape != (257,) or not np.array_equal(r[pick], part):
            raise ValueError("flip_rate_fn is not elementwise in (sigma, m): it cannot be tabulated for the device modes; use mode='gillespie'")
        if not np.all(np.isfinite(r)) or np.any(r < 0):
            raise ValueError("flip_rate_fn must return finite rates >= 0 on m in [-1, 1]")
        rows.append(r)
    return np.ascontiguousarray(np.stack(rows))


class ParticleSystem:
    def __init__(self, L, xlim, rate_diffusion, rate_active, beta, flip_rate_fn=None, init="fixed",
                 N=1000, rho0_plus=None, rho0_minus=None, rng=None, scale_rates=True,
                 local_kernel_sigma=0.005, periodic=False, minus_anchor=True,
                 immobilize_when_anchored=True, anchor_positions=None, anchor_radius=0.005,
                 site_capacity=1, crowding_suppresses_rates=False, k_on=0.1, k_off=0.01,
                 suppress_flip_when_bound=True, k_exit=0,
                 # extensions (all optional, after the reference's keywords)
                 dt=None, seed=None, device=0, sort_by_site=True, ensemble=0, mode=None, method="auto", fp32=False):
        self.L = int(L)
        self.xlim = xlim
        self.K = int(site_capacity)
        self.dx = self.xlim / self.L
        if scale_rates:                                            # ref :45-50
            self.rate_diffusion = rate_diffusion / (self.dx ** 2)
            self.rate_active = rate_active / self.dx
        else:
            self.rate_diffusion = float(rate_diffusion)
            self.rate_active = float(rate_active)
        self.beta = beta
        self.k_on, self.k_off, self.k_exit = k_on, k_off, k_exit
        self.suppress_flip_when_bound = suppress_flip_when_bound
        self.crowding_suppresses_rates = crowding_suppresses_rates
        # a custom flip rate (ref :59-62, applied at :261-262) is an arbitrary Python callable: it runs on the host, inside
        # the exact one-event-per-iteration loop (mode="gillespie"), where the other five rate channels still come from the GPU
        self.flip_rate_fn = flip_rate_fn
        assert init in ("fixed", "poisson")
        self.init_mode = init
        if init == "fixed":
            if float(N) != int(N):
                raise ValueError("N must be integral")
            self.N_fixed = int(N)
        else:
            self.rho0_plus = np.array([rho0_plus(i / self.L) for i in range(self.L)], dtype=float)
            self.rho0_minus = np.array([rho0_minus(i / self.L) for i in range(self.L)], dtype=float)
        self.rng = np.random.default_rng() if rng is None else rng
        self.local_kernel_sigma = local_kernel_sigma
        self.periodic = periodic
        self.immobilize_when_anchored = immobilize_when_anchored
        self.minus_anchor = minus_anchor
        self._sigma_grid = self.local_kernel_sigma / self.dx
        self.anchor_radius = anchor_radius
        self.anchor_positions = anchor_positions
        self.is_anchor_site = np.zeros(self.L, dtype=bool)
        if anchor_positions is None:
            self.anchor_idxs = np.array([], dtype=int)
            self.anchor_idx_array = np.array([], dtype=int)
        else:                                                      # ref :93-104
            centres = np.unique(np.round((np.asarray(anchor_positions, dtype=float) / self.xlim)
                                         * (self.L - 1)).astype(int))
            self.anchor_idxs = centres
            reach = int(np.ceil(anchor_radius / self.dx))
            for c in centres:
                self.is_anchor_site[max(0, c - reach):min(self.L - 1, c + reach) + 1] = True
            self.anchor_idx_array = np.flatnonzero(self.is_anchor_site)
        # stepper extensions
        self.dt = None if dt is None else float(dt)
        self.seed = seed
        self.device = int(device)
        self.sort_by_site = bool(sort_by_site)
        self.ensemble = int(ensemble)          # Philox counter word 3 (independent streams under one seed)
        if mode is None:                       # an unchanged driver gets the reference's dynamics (ref :511-516, :358-362)
            mode = "sync" if dt is not None else ("gillespie" if flip_rate_fn is not None else "gillespie_gpu")
        if mode not in ("sync", "gillespie", "gillespie_gpu"):
            raise ValueError("mode must be 'sync' (fixed-dt stepper), 'gillespie' (one exact event per iteration, drawn "
                             "from rng on the host) or 'gillespie_gpu' (the exact event loop resident on the GPU)")
        self.mode = mode
        self._flip_table = None
        if flip_rate_fn is not None and mode != "gillespie":
            # the device modes cannot call back into Python: the callable is tabulated here over m in [-1, 1] for sigma = +-1 and
            # the kernels interpolate the table linearly (aps_set_flip_table); mode="gillespie" applies the callable itself
            self._flip_table = tabulate_flip_rate(flip_rate_fn)
        if method not in capi.METHODS:
            raise ValueError("method must be 'auto', 'tiles' (one kernel per step over site tiles), 'lattice' (incremental lattice "
                             "field, three kernels) or 'pairs' (all-pairs kernel)")
        self.method = method
        self.fp32 = bool(fp32)                 # 32-bit integer field (aps_params.fp32): float32-class accuracy, still order-independent
        self._handle = None
        self._util = None                      # lazily created handle for compute_local_m_field / step_gillespie

    # ------------------------------------------------------------------ initial conditions (host)
    def _init_fixed(self):
        N, L, K = self.N_fixed, self.L, self.K
        if K == 1:                                                 # ref :144-147
            sites = self.rng.choice(L, size=N, replace=False)
        else:                                                      # ref :149-156, one draw per particle
            sites = np.empty(N, dtype=np.int64)
            fill = np.zeros(L, dtype=int)
            for k in range(N):
                sites[k] = self.rng.choice(np.flatnonzero(fill < K))
                fill[sites[k]] += 1
        spins = self.rng.choice([1, -1], size=N)
        return sites.astype(np.int64), spins.astype(np.int8)

    def _init_poisson(self):
        plus = self.rng.poisson(self.rho0_plus)                    # ref :161-162
        minus = self.rng.poisson(self.rho0_minus)
        sites, spins = [], []
        for x in np.flatnonzero(plus + minus):
            kinds = np.array([1] * int(plus[x]) + [-1] * int(minus[x]), dtype=int)
            if len(kinds) > self.K:                                # ref :174-176
                kinds = kinds[self.rng.choice(len(kinds), size=self.K, replace=False)]
            sites += [x] * len(kinds)
            spins += kinds.tolist()
        return np.array(sites, dtype=np.int64), np.array(spins, dtype=np.int8)

    def init_particles(self):
        return self._init_fixed() if self.init_mode == "fixed" else self._init_poisson()

    @staticmethod
    def empirical_densities_from_particles(pos, sigma, L, dx, total_norm=None):
        cp = np.bincount(pos[sigma == 1], minlength=L)
        cm = np.bincount(pos[sigma == -1], minlength=L)
        denom = float(max(1, pos.size) if total_norm is None else total_norm) * dx
        return (cp / denom).astype(float), (cm / denom).astype(float)

    # ------------------------------------------------------------------ stepper plumbing
    def default_dt(self):
        """0.1 / (upper bound of one particle's total rate): max_i r_i * dt <= 0.1."""
        flip_max = math.exp(abs(self.beta)) if self._flip_table is None else float(self._flip_table.max())
        rmax = (2.0 * self.rate_diffusion + self.rate_active + flip_max
                + max(self.k_on, self.k_off) + self.k_exit)
        return 0.1 / rmax

    def _make_handle(self, n_particles, seed, betas=None, ensemble_base=None):
        if self.dt is None:
            self.dt = self.default_dt()
        return capi.Handle(
            L=self.L, K=self.K, periodic=self.periodic, sigma_grid=self._sigma_grid,
            rate_diffusion=self.rate_diffusion, rate_active=self.rate_active,
            beta=[float(self.beta)] if betas is None else betas,
            ensemble_base=self.ensemble if ensemble_base is None else ensemble_base,
            dt=self.dt, seed=seed, n_particles=n_particles, minus_anchor=self.minus_anchor,
            immobilize=self.immobilize_when_anchored, suppress_flip=self.suppress_flip_when_bound,
            crowding=self.crowding_suppresses_rates, k_on=self.k_on, k_off=self.k_off, k_exit=self.k_exit,
            anchor_mask=self.is_anchor_site, device=self.device, sort_by_site=self.sort_by_site, method=self.method, fp32=self.fp32)

    def _stepper_handle(self, *args, **kw):
        """_make_handle plus the tabulated flip rate, for the fixed-dt stepper"""
        h = self._make_handle(*args, **kw)
        if self._flip_table is not None:
            h.set_flip_table(self._flip_table)
        return h

    def flip_table(self):
        """[2][n + 1] table of a custom flip_rate_fn for the device modes (None: Curie-Weiss rate, or mode="gillespie")."""
        return self._flip_table

    def _utility_handle(self):
        if self._util is None:
            self._util = self._make_handle(1, 0)
        return self._util

    def close(self):
        if self._util is not None:
            self._util.close()
            self._util = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compute_local_m_field(self, counts_p, counts_m):
        """m-field on all L sites for given site histograms (ref :216-246), evaluated on the GPU."""
        return self._utility_handle().field_from_counts(counts_p, counts_m)

    def step_gillespie(self, pos, sigma, bound, m_field, counts_p, counts_m, init_bin, exit_times, exit_positions,
                       exit_init_bin, t):
        """One exact Gillespie event (ref :254-448), same arguments, in-place updates and return tuple.
        The per-particle rate vectors are evaluated on the GPU (aps_rates_from_field); the waiting time, the
        particle and the event are drawn from `self.rng` with the reference's call sequence
        (exponential, choice(p=rates/R), random[, random])."""
        n = sigma.size
        if n == 0:
            return pos, sigma, bound, np.inf, counts_p, counts_m
        rt = self._utility_handle().rates_from_field(pos, sigma, bound, m_field, counts_p, counts_m)
        if self.flip_rate_fn is not None:                          # ref :261-267, then the sum of :351 in its order
            cvec = np.asarray(self.flip_rate_fn(sigma, m_field[pos]), dtype=float).copy()
            if self.suppress_flip_when_bound:
                cvec[np.asarray(bound, dtype=bool)] = 0.0
            rt["flip"] = cvec
            rt["total"] = ((((rt["diff"] + rt["act"]) + cvec) + rt["bind"]) + rt["unbind"]) + rt["exit"]
        rates = rt["total"]
        R = float(rates.sum())
        if R <= 0:
            return pos, sigma, bound, np.inf, counts_p, counts_m
        tau = self.rng.exponential(1.0 / R)
        i = self.rng.choice(n, p=rates / R)
        v = self.rng.random() * rates[i]
        edge_diff = rt["diff"][i]
        edge_act = edge_diff + rt["act"][i]
        edge_bind = edge_act + rt["bind"][i]
        edge_unbind = edge_bind + rt["unbind"][i]
        edge_exit = edge_unbind + rt["exit"][i]
        here = pos[i]
        lane = counts_p if sigma[i] == 1 else counts_m

        def hop(to):
            to = to % self.L if self.periodic else min(max(to, 0), self.L - 1)
            pos[i] = to
            lane[here] -= 1
            lane[to] += 1

        if v < edge_diff:
            a, b = rt["left"][i], rt["right"][i]
            if a + b <= 0:
                return pos, sigma, bound, tau, counts_p, counts_m
            hop(here - 1 if self.rng.random() < a / (a + b) else here + 1)
        elif v < edge_act:
            hop(here + (1 if sigma[i] == 1 else 0))
        elif v < edge_bind:
            bound[i] = True
        elif v < edge_unbind:
            bound[i] = False
        elif v < edge_exit:
            exit_times.append(t)
            exit_positions.append(pos[i])
            exit_init_bin.append(int(init_bin[i]))
            lane[here] -= 1
            pos, sigma, bound = np.delete(pos, i), np.delete(sigma, i), np.delete(bound, i)
        else:
            other = counts_m if sigma[i] == 1 else counts_p
            lane[here] -= 1
            other[here] += 1
            sigma[i] = -sigma[i]
        return pos, sigma, bound, tau, counts_p, counts_m, exit_times, exit_positions, exit_init_bin

    # ------------------------------------------------------------------ run (ref :450-558)
    def run(self, T=10.0, obs_dt=0.01, record_fft=False, record_var=False):
        if self.mode == "gillespie":
            return self._run_gillespie(T, obs_dt, record_fft, record_var)
        if self.mode == "gillespie_gpu":
            from .gillespie import run_batched_exact
            return run_batched_exact([self], T=T, obs_dt=obs_dt, record_fft=record_fft, record_var=record_var)[0]
        return run_batched([self], T=T, obs_dt=obs_dt, record_fft=record_fft, record_var=record_var)[0]

    def _run_gillespie(self, T, obs_dt, record_fft, record_var):
        """The reference's event loop as written (ref :450-558): one event per iteration, observation after the
        event that crossed the observation time, m_local_list = field from before that event.  The field and
        the rate vectors of every event come from the GPU."""
        L, dx = self.L, self.dx
        pos, sigma = self.init_particles()
        bound = np.zeros_like(sigma, dtype=bool)
        times_obs = np.arange(0.0, T, obs_dt)
        M = len(times_obs)
        pos_list, particle_count_list, bound_list = [None] * M, [None] * M, [None] * M
        rho_p_list, rho_m_list, total_list = np.zeros((M, L)), np.zeros((M, L)), np.zeros((M, L))
        m_local_list, m_global = np.zeros((M, L)), np.zeros(M)
        rho_hat_complex = np.zeros((M, L), dtype=complex) if record_fft else None
        fft_amp_list = np.zeros((M, L)) if record_fft else None
        var_list = np.zeros(M) if record_var else None
        exit_times, exit_positions, exit_init_bin = [], [], []
        init_bin = pos.copy()
        counts_p = np.bincount(pos[sigma == 1], minlength=L)
        counts_m = np.bincount(pos[sigma == -1], minlength=L)

        def snapshot(k, field):
            pos_list[k] = pos.copy()
            rho_p, rho_m = self.empirical_densities_from_particles(pos, sigma, L, dx)
            rho_p_list[k], rho_m_list[k], total_list[k] = rho_p, rho_m, rho_p + rho_m
            particle_count_list[k] = pos.size
            bound_list[k] = bound.copy()
            m_local_list[k] = field
            m_global[k] = np.mean(sigma)
            if record_fft:
                u = total_list[k]
                spec = np.fft.fft(u)
                rho_hat_complex[k], fft_amp_list[k] = spec, np.abs(spec)
                if record_var:
                    var_list[k] = float(np.var(u))

        field = self.compute_local_m_field(counts_p, counts_m)
        snapshot(0, field)
        k, t, self.n_events = 1, 0.0, 0
        while t < T:
            field = self.compute_local_m_field(counts_p, counts_m)
            ret = self.step_gillespie(pos, sigma, bound, field, counts_p, counts_m, init_bin, exit_times, exit_positions,
                                      exit_init_bin, t)
            pos, sigma, bound, tau = ret[0], ret[1], ret[2], ret[3]
            self.n_events += 1
            t += tau
            if t > T:
                break
            while k < M and times_obs[k] <= t:
                snapshot(k, field)
                k += 1
            if k >= M:
                break
        return {
            "times_obs": times_obs, "pos_list": pos_list, "rho_p_list": rho_p_list, "rho_m_list": rho_m_list,
            "total_list": total_list, "particle_count_list": particle_count_list, "bound_list": bound_list,
            "m_local_list": m_local_list, "m_global": m_global, "rho_hat_complex": rho_hat_complex,
            "fft_amp_list": fft_amp_list, "var_list": var_list, "exit_times": exit_times,
            "exit_positions": exit_positions,
        }

    # ------------------------------------------------------------------ presentation-only methods of the reference
    def visualize_all(self, *args, **kwargs):
        raise NotImplementedError("visualize_all (matplotlib figures, reference :561-661) is presentation code "
                                  "outside the accelerated path; plot the returned `out` dictionary yourself")

    def animate_profiles(self, *args, **kwargs):
        raise NotImplementedError("animate_profiles (vispy animation, reference :980-1093) is presentation code "
                                  "outside the accelerated path")

    # ------------------------------------------------------------------ the one plotting method a driver uses
    def plot_individuals(self, out, show_k_max=6, cmap_name="viridis", xlim=1, fig_size=(10, 6)):
        """Returns mean_v_eff like the reference (ref :901-906, :978); the PNG output of the
        reference's method is presentation code and is not reproduced."""
        total = out["total_list"]
        grid = np.linspace(0, 1.0, self.L)
        com = (total * grid).sum(axis=1) / (total.sum(axis=1) + 1e-12)
        v_eff = np.gradient(com, out["times_obs"])
        return np.mean(v_eff[int(len(v_eff) * 0.6):])


_SHAPE_ATTRS = ("L", "xlim", "K", "rate_diffusion", "rate_active", "k_on", "k_off", "k_exit", "periodic",
                "local_kernel_sigma", "minus_anchor", "immobilize_when_anchored", "suppress_flip_when_bound",
                "crowding_suppresses_rates")


def run_batched(systems, T=10.0, obs_dt=0.01, record_fft=False, record_var=False):
    """`run()` of several ParticleSystem objects at once: they may differ in beta, rng / initial condition and
    particle number but share every other parameter, and are stepped together as independent ensembles of ONE
    GPU handle (BASELINE config 4; the reference loops over them serially, ..._sweep_beta.py:75-95).
    Returns the list of result dictionaries (reference :542-557), one per system, in order.

    Ensemble e uses Philox counter word 3 = systems[0].ensemble + e under the common key `seed` of the first
    system (drawn from its rng after the initial conditions unless given), so a batched run equals separate
    runs with `ParticleSystem(..., seed=seed, ensemble=e)`."""
    first = systems[0]
    for ps in systems[1:]:
        for k in _SHAPE_ATTRS:
            if getattr(ps, k) != getattr(first, k):
                raise ValueError(f"run_batched: systems differ in {k}")
        if not np.array_equal(ps.is_anchor_site, first.is_anchor_site):
            raise ValueError("run_batched: systems differ in anchor sites")
        if (ps.flip_table() is None) != (first.flip_table() is None) or (ps.flip_table() is not None and not np.array_equal(ps.flip_table(), first.flip_table())):
            raise ValueError("run_batched: systems differ in flip_rate_fn")
    L, dx = first.L, first.dx
    inits = [ps.init_particles() for ps in systems]
    # the reference only requires choice / poisson / exponential / random of an rng object (ref :75-78)
    seed = first.seed if first.seed is not None else int(first.rng.random() * 2.0 ** 53)
    if first.dt is None:
        first.dt = min(ps.default_dt() for ps in systems)
    for ps in systems:
        ps.dt, ps.seed_used = first.dt, seed
    dt = first.dt
    cap = max(1, max(len(p) for p, _ in inits))
    h = first._stepper_handle(cap, seed, betas=[float(ps.beta) for ps in systems], ensemble_base=first.ensemble)
    E = len(systems)
    try:
        for e, (pos0, sigma0) in enumerate(inits):
            h.set_state(pos0, sigma0, ensemble=e)
        times_obs = np.arange(0.0, T, obs_dt)
        M = len(times_obs)
        recs = [dict(pos=[None] * M, count=[None] * M, bound=[None] * M, rho_p=np.zeros((M, L)), rho_m=np.zeros((M, L)),
                     total=np.zeros((M, L)), m_loc=np.zeros((M, L)), m_glob=np.zeros(M),
                     hat=np.zeros((M, L), dtype=complex) if record_fft else None,
                     amp=np.zeros((M, L)) if record_fft else None, var=np.zeros(M) if record_var else None)
                for _ in range(E)]
        done_steps = 0
        for k in range(M):
            # smallest step count whose time reaches the observation time (ref :517)
            want = int(math.ceil(times_obs[k] / dt - 1e-9))
            if want > done_steps:
                h.step(want - done_steps)
                done_steps = want
            for e, rec in enumerate(recs):
                pos, sigma, bound, alive = h.get_state(ensemble=e)
                live = alive.astype(bool)
                p, sg = pos[live].astype(np.int64), sigma[live]
                rec["pos"][k] = p
                rho_p, rho_m = ParticleSystem.empirical_densities_from_particles(p, sg, L, dx)
                rec["rho_p"][k], rec["rho_m"][k], rec["total"][k] = rho_p, rho_m, rho_p + rho_m
                rec["count"][k] = p.size
                rec["bound"][k] = bound[live].astype(bool)
                rec["m_loc"][k] = h.observe(ensemble=e, want_field=True)[2]
                rec["m_glob"][k] = np.mean(sg) if sg.size else np.nan
                if record_fft:
                    u = rec["total"][k]
                    spec = np.fft.fft(u)
                    rec["hat"][k], rec["amp"][k] = spec, np.abs(spec)
                    if record_var:                      # kept only together with the FFT (ref :499-507)
                        rec["var"][k] = float(np.var(u))
            if h.method == "pairs":                     # tile culling of the all-pairs kernel likes site-sorted slots;
                h.resort()                              # the lattice formulation does not care about the slot order
        exits = [h.exits(ensemble=e) for e in range(E)]
        for ps in systems:
            ps.steps_done = done_steps
    finally:
        h.close()
    return [{
        "times_obs": times_obs.copy() if E > 1 else times_obs, "pos_list": rec["pos"], "rho_p_list": rec["rho_p"],
        "rho_m_list": rec["rho_m"], "total_list": rec["total"], "particle_count_list": rec["count"],
        "bound_list": rec["bound"], "m_local_list": rec["m_loc"], "m_global": rec["m_glob"],
        "rho_hat_complex": rec["hat"], "fft_amp_list": rec["amp"], "var_list": rec["var"],
        "exit_times": [float(t) for t in ex[:, 0]], "exit_positions": [int(x) for x in ex[:, 1]],
    } for rec, ex in zip(recs, exits)]


def run_batched_statistics(systems, T=10.0, obs_dt=0.01):
    """The sweep drivers' per-run observables (v_eff, D_eff, mean magnetisation, front density, blocking
    probability; ..._sweep_beta.py:85-95) for several systems stepped together, WITHOUT materialising the M x L
    arrays of `run()`: at every observation time the device returns a dozen integer sums per ensemble
    (aps_observe_scalars).  Only valid while no particle exits (k_exit = 0, as in every reference sweep); returns a
    list of dicts like observables.run_observables."""
    from . import observables
    first = systems[0]
    for ps in systems[1:]:
        for k in _SHAPE_ATTRS:
            if getattr(ps, k) != getattr(first, k):
                raise ValueError(f"run_batched_statistics: systems differ in {k}")
    if first.k_exit:
        raise ValueError("run_batched_statistics needs k_exit = 0 (use run_batched and observables.run_observables)")
    inits = [ps.init_particles() for ps in systems]
    seed = first.seed if first.seed is not None else int(first.rng.random() * 2.0 ** 53)
    if first.dt is None:
        first.dt = min(ps.default_dt() for ps in systems)
    for ps in systems:
        ps.dt, ps.seed_used = first.dt, seed
    dt = first.dt
    cap = max(1, max(len(p) for p, _ in inits))
    h = first._stepper_handle(cap, seed, betas=[float(ps.beta) for ps in systems], ensemble_base=first.ensemble)
    times_obs = np.arange(0.0, T, obs_dt)
    accs = [observables.DeviceObservables(times_obs, first.L, first.dx, first.K) for _ in systems]
    try:
        for e, (pos0, sigma0) in enumerate(inits):
            h.set_state(pos0, sigma0, ensemble=e)
        done = 0
        for k, t_obs in enumerate(times_obs):
            want = int(math.ceil(t_obs / dt - 1e-9))
            if want > done:
                h.step(want - done)
                done = want
            if k == accs[0].start:
                for e in range(len(accs)):
                    h.mark_reference(ensemble=e)
            # all ensembles in one pass; the blocking threshold is shared when the particle numbers allow it
            tables = [acc.block_table(len(inits[e][0])) for e, acc in enumerate(accs)]   # no exits: particle numbers are constant
            if all(np.array_equal(tb, tables[0]) for tb in tables[1:]):
                rows = h.observe_scalars_all(x_wall=accs[0].x_wall, block_table=tables[0])
                fronts = [None] * len(accs)
                if k >= accs[0].start:
                    ranges = [accs[e].front_range(r["max_pos"]) if r["max_pos"] >= 0 else (0, -1) for e, r in enumerate(rows)]
                    again = h.observe_scalars_all(x_wall=accs[0].x_wall, ranges=ranges, block_table=tables[0])
                    fronts = [again[e]["n_range"] if rows[e]["max_pos"] >= 0 else None for e in range(len(accs))]
                for e, acc in enumerate(accs):
                    acc.add(k, rows[e], fronts[e])
            else:
                for e, acc in enumerate(accs):
                    sums = h.observe_scalars(ensemble=e, x_wall=acc.x_wall, block_table=tables[e])
                    n_front = None
                    if k >= acc.start and sums["max_pos"] >= 0:
                        lo, hi = acc.front_range(sums["max_pos"])
                        n_front = h.observe_scalars(ensemble=e, x_wall=acc.x_wall, range_lo=lo, range_hi=hi)["n_range"]
                    acc.add(k, sums, n_front)
        for ps in systems:
            ps.steps_done = done
    finally:
        h.close()
    return [acc.result() for acc in accs]


def run_batched_structure(systems, T=10.0, obs_dt=0.01, start_fraction=0.5, k_max=None):
    """The structure observables of PARTICLE_solver_BIOLOGY_local_structure.py:55-103 (time mean / spread of var(total) and of
    |fft(total)|, dominant mode, low-k power, variance of the local magnetisation, low-k variance) for several systems stepped
    together as ensembles of one handle, accumulated from per-observation sums taken on the GPU (aps_observe_structure):
    the reference materialises three M x L arrays per run for them.  `k_max=None` means all L modes, as in the reference.
    Returns a list of dicts with the reference's eight keys (observables.DeviceStructure.result)."""
    from . import observables
    first = systems[0]
    for ps in systems[1:]:
        for k in _SHAPE_ATTRS:
            if getattr(ps, k) != getattr(first, k):
                raise ValueError(f"run_batched_structure: systems differ in {k}")
    inits = [ps.init_particles() for ps in systems]
    seed = first.seed if first.seed is not None else int(first.rng.random() * 2.0 ** 53)
    if first.dt is None:
        first.dt = min(ps.default_dt() for ps in systems)
    for ps in systems:
        ps.dt, ps.seed_used = first.dt, seed
    dt = first.dt
    cap = max(1, max(len(p) for p, _ in inits))
    h = first._stepper_handle(cap, seed, betas=[float(ps.beta) for ps in systems], ensemble_base=first.ensemble)
    times_obs = np.arange(0.0, T, obs_dt)
    kk = first.L if k_max is None else min(int(k_max), first.L)
    accs = [observables.DeviceStructure(len(times_obs), first.L, first.dx, start_fraction, kk) for _ in systems]
    try:
        for e, (pos0, sigma0) in enumerate(inits):
            h.set_state(pos0, sigma0, ensemble=e)
        done = 0
        for k, t_obs in enumerate(times_obs):
            want = int(math.ceil(t_obs / dt - 1e-9))
            if want > done:
                h.step(want - done)
                done = want
            if k >= accs[0].start:
                for e, acc in enumerate(accs):
                    acc.add(k, *h.observe_structure(ensemble=e, k_max=kk))
        for ps in systems:
            ps.steps_done = done
    finally:
        h.close()
    return [acc.result() for acc in accs]
