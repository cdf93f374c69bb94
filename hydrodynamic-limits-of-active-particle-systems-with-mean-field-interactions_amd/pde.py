"""`IMEXPDE` -- the reference's hydrodynamic-limit solver class (IMEX_PDE_solver_class.py:11) re-hosted on the MI355X.

Same constructor keywords (ref :13-28), `initialize()` (ref :96-131, host NumPy with the reference's legacy
`np.random` call sequence, so a seeded initial condition is the reference's), `solve()`, `get_output()` keys
(ref :293-306).  The time loop (`solve` + `step`, the per-step observables and the Euler-Maruyama tracers) runs in
one persistent HIP kernel per system behind the C ABI of include/pde.h; `solve_batch` runs many beta values at once
(the reference's sweep drivers loop over them serially, IMEX_PDE_solver_run_sweep.py:17-48).

Differences that are part of the design: tracer noise comes from Philox4x32-10 keyed by `seed` on the device (the
reference draws from NumPy's global MT19937 inside the loop); the magnetisation kernel is applied by direct circular
convolution instead of rfft products; no output directory is created; plotting methods are not reproduced.
There is no CPU fallback: without libaps_hip.so or without a GPU `solve()` raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class PdeParams(C.Structure):
    """struct pde_params of include/pde.h, field for field."""
    _fields_ = [("L", C.c_int32), ("nsteps", C.c_int32), ("periodic", C.c_int32), ("anchored_minus", C.c_int32),
                ("kernel_mode", C.c_int32), ("snapshot_interval", C.c_int32), ("n_tracers", C.c_int32),
                ("window", C.c_int32), ("n_fft_modes", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32 * 2),
                ("xlim", C.c_double), ("dt", C.c_double), ("gamma", C.c_double), ("lam", C.c_double),
                ("kernel_sigma", C.c_double), ("seed", C.c_uint64)]


PDE_MAX_L = 1 << 22          # fields in LDS up to L ~ 3000 (PDE_LDS_L of include/pde.h), in global memory beyond


def _lib():
    lib = capi.load()
    if not getattr(lib, "_pde_ready", False):
        vp = C.c_void_p
        lib.pde_last_error.restype, lib.pde_last_error.argtypes = C.c_char_p, []
        lib.pde_solve_batch.restype = C.c_int
        lib.pde_solve_batch.argtypes = [C.POINTER(PdeParams), C.c_int32] + [vp] * 19 + [C.POINTER(C.c_double)]
        lib._pde_ready = True
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def solve_batch_raw(*, L, xlim, dt, nsteps, gamma, lam, betas, bc, active_model, gaussian_kernel, kernel_sigma,
                    snapshot_interval, rho_p0, rho_m0, tracer_x0=None, tracer_s0=None, rand_u=None, rand_n=None,
                    n_fft_modes=0, seed=0, device=0, want_snapshots=True):
    """All systems of `betas` from their initial states to step nsteps on the GPU; dict of arrays with a leading
    system axis.  rand_u / rand_n [n_systems, nsteps+1, n_tracers] replace the device's Philox draws (tests)."""
    lib = _lib()
    betas = np.ascontiguousarray(np.atleast_1d(betas), dtype=np.float64)
    S = len(betas)
    rho_p0 = np.ascontiguousarray(np.broadcast_to(rho_p0, (S, L)), dtype=np.float64)
    rho_m0 = np.ascontiguousarray(np.broadcast_to(rho_m0, (S, L)), dtype=np.float64)
    ntr = 0 if tracer_x0 is None else np.shape(tracer_x0)[-1]
    if ntr:
        tracer_x0 = np.ascontiguousarray(np.broadcast_to(tracer_x0, (S, ntr)), dtype=np.float64)
        tracer_s0 = np.ascontiguousarray(np.broadcast_to(tracer_s0, (S, ntr)), dtype=np.int8)
    if rand_u is not None:
        rand_u = np.ascontiguousarray(np.broadcast_to(rand_u, (S, nsteps + 1, ntr)), dtype=np.float64)
        rand_n = np.ascontiguousarray(np.broadcast_to(rand_n, (S, nsteps + 1, ntr)), dtype=np.float64)
    if not gaussian_kernel:
        mode = 0
    elif kernel_sigma > 100000:                                    # ref :161
        mode = 2
    else:
        mode = 1
    window = int(0.05 / dt)                                        # ref :238-239
    par = PdeParams(L=L, nsteps=nsteps, periodic=int(bc == "periodic"), anchored_minus=int(active_model != "bidirectional"),
                    kernel_mode=mode, snapshot_interval=snapshot_interval, n_tracers=ntr, window=max(window, 1),
                    n_fft_modes=n_fft_modes, device=device, xlim=xlim, dt=dt, gamma=gamma, lam=lam,
                    kernel_sigma=kernel_sigma, seed=int(seed) & (2 ** 64 - 1))
    if bc not in ("periodic", "neumann"):
        raise ValueError("bc must be 'periodic' or 'neumann'")
    n_snap = nsteps // snapshot_interval + 1
    out = dict(rho_p=np.zeros((S, L)), rho_m=np.zeros((S, L)), m_series=np.zeros((S, nsteps + 1)),
               var_series=np.zeros((S, nsteps + 1)), v_eff_series=np.full((S, nsteps + 1), np.nan),
               D_eff_series=np.full((S, nsteps + 1), np.nan))
    snaps = np.zeros((S, n_snap, L)) if want_snapshots else None
    msnaps = np.zeros((S, n_snap, L)) if want_snapshots else None
    fre = np.zeros((S, nsteps + 1, n_fft_modes)) if n_fft_modes else None
    fim = np.zeros((S, nsteps + 1, n_fft_modes)) if n_fft_modes else None
    tx = np.zeros((S, ntr)) if ntr else None
    ts = np.zeros((S, ntr), np.int8) if ntr else None
    ms = C.c_double()
    rc = lib.pde_solve_batch(C.byref(par), S, _p(betas), _p(rho_p0), _p(rho_m0), _p(tracer_x0) if ntr else None,
                             _p(tracer_s0) if ntr else None, _p(rand_u), _p(rand_n), _p(out["rho_p"]), _p(out["rho_m"]),
                             _p(out["m_series"]), _p(out["var_series"]), _p(out["v_eff_series"]) if ntr else None,
                             _p(out["D_eff_series"]) if ntr else None, _p(snaps), _p(msnaps), _p(fre), _p(fim), _p(tx), _p(ts),
                             C.byref(ms))
    if rc != 0:
        raise capi.ApsError(rc, lib.pde_last_error().decode())
    out.update(snapshots=snaps, m_snapshots=msnaps, fft_re=fre, fft_im=fim, tracers_unwrapped=tx, tracer_state=ts,
               times=np.arange(n_snap) * snapshot_interval * dt, kernel_ms=ms.value)
    return out


class IMEXPDE:
    def __init__(self, L=1000, xlim=1.0, T=10.0, dt=5e-4, gamma=2.33e-4, lam=0.6, beta=2.0, bc="periodic",
                 active_model="bidirectional", gaussian_kernel=False, kernel_sigma=0.02, snapshot_interval=50,
                 outdir="IMEX_output", seed=None,
                 # extensions (optional, after the reference's keywords)
                 device=0, record_fft=True):
        self.L, self.xlim, self.dx = L, xlim, xlim / L
        self.x = np.linspace(0, xlim, L, endpoint=False)
        self.T, self.dt, self.nsteps = T, dt, int(T / dt)
        self.gamma, self.lam, self.beta = gamma, lam, beta
        self.bc, self.active_model = bc, active_model
        self.gaussian_kernel, self.kernel_sigma = gaussian_kernel, kernel_sigma
        self.snapshot_interval, self.seed = snapshot_interval, seed
        self.outdir = outdir                                       # kept as an attribute; nothing is written
        self.device, self.record_fft = int(device), bool(record_fft)
        if L > PDE_MAX_L:
            raise ValueError(f"L <= {PDE_MAX_L}")
        if seed is not None:
            np.random.seed(seed)                                   # ref :55-56
        self.rho_mean = 1.0 / self.xlim
        self._out = None

    def cw_rate(self, sigma, m):                                   # ref :64-66
        return np.clip(np.exp(-self.beta * sigma * m), 1e-8, 1e8)

    def initialize(self, mode="poisson", rho0=1.0, noise=0.2, n_tracers=1000):   # ref :96-131, host side
        L = self.L
        if mode == "homogeneous":
            rho_p = rho0 + noise * np.random.randn(L)
            rho_m = rho0 + noise * np.random.randn(L)
        elif mode == "poisson":
            rho_p = np.exp(-np.abs(self.x - 0.5) / 0.05)
            rho_m = np.exp(-np.abs(self.x - 0.5) / 0.05)
            rho_p += noise * np.random.randn(L)
            rho_m += noise * np.random.randn(L)
        else:
            raise ValueError("Unknown init mode.")
        rho_p, rho_m = np.clip(rho_p, 0, None), np.clip(rho_m, 0, None)
        tot = (rho_p + rho_m).sum()
        self.rho_p, self.rho_m = rho_p / tot, rho_m / tot
        self.n_tracers = n_tracers
        self.tracers = np.random.choice(L, size=n_tracers) * self.dx
        self.tracers_unwrapped = self.tracers.copy()
        self.tracer_state = np.random.choice([-1, 1], size=n_tracers)
        self._out = None

    def _run(self, betas, rho_p0, rho_m0, tx0, ts0, rand_u=None, rand_n=None, want_snapshots=True):
        seed = self.seed if self.seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        return solve_batch_raw(L=self.L, xlim=self.xlim, dt=self.dt, nsteps=self.nsteps, gamma=self.gamma, lam=self.lam,
                               betas=betas, bc=self.bc, active_model=self.active_model,
                               gaussian_kernel=self.gaussian_kernel, kernel_sigma=self.kernel_sigma,
                               snapshot_interval=self.snapshot_interval, rho_p0=rho_p0, rho_m0=rho_m0, tracer_x0=tx0,
                               tracer_s0=ts0, rand_u=rand_u, rand_n=rand_n,
                               n_fft_modes=self.L // 2 + 1 if self.record_fft else 0, seed=seed, device=self.device,
                               want_snapshots=want_snapshots)

    def solve(self, rand_u=None, rand_n=None):                     # ref :236-290, on the GPU
        r = self._run([self.beta], self.rho_p, self.rho_m, self.tracers_unwrapped if self.n_tracers else None,
                      self.tracer_state if self.n_tracers else None, rand_u, rand_n)
        self._adopt(r, 0)
        return self

    def _adopt(self, r, s):
        self.rho_p, self.rho_m = r["rho_p"][s], r["rho_m"][s]
        self.m_series, self.var_series = r["m_series"][s], r["var_series"][s]
        self.v_eff_series, self.D_eff_series = r["v_eff_series"][s], r["D_eff_series"][s]
        self.snapshots = list(r["snapshots"][s]) if r["snapshots"] is not None else []
        self.m_snapshots = list(r["m_snapshots"][s]) if r["m_snapshots"] is not None else []
        self.times = list(r["times"])
        if r["fft_re"] is not None:
            self.fft_phase = r["fft_re"][s] + 1j * r["fft_im"][s]
            self.fft_amp = np.abs(self.fft_phase)
        else:
            self.fft_phase = self.fft_amp = None
        if r["tracers_unwrapped"] is not None:
            self.tracers_unwrapped = r["tracers_unwrapped"][s]
            self.tracers = self.tracers_unwrapped % self.xlim
            self.tracer_state = r["tracer_state"][s].astype(int)
        self.kernel_ms = r["kernel_ms"]

    def solve_batch(self, betas, want_snapshots=False):
        """The same initial condition evolved for every beta of `betas` in ONE launch (one workgroup per beta;
        independent tracer noise per system).  Returns the raw dict of arrays with a leading system axis."""
        return self._run(betas, self.rho_p, self.rho_m, self.tracers_unwrapped if self.n_tracers else None,
                         self.tracer_state if self.n_tracers else None, want_snapshots=want_snapshots)

    def get_output(self):                                          # ref :293-306
        return dict(rho_p=self.rho_p, rho_m=self.rho_m, m_series=self.m_series, var_series=self.var_series,
                    fft_amp=self.fft_amp, fft_phase=self.fft_phase, snapshots=np.array(self.snapshots),
                    m_snapshots=np.array(self.m_snapshots), times=np.array(self.times),
                    v_eff_series=self.v_eff_series, D_eff_series=self.D_eff_series)

    def plot_all(self, *args, **kwargs):
        raise NotImplementedError("plot_all (matplotlib figures, reference :309-346) is presentation code outside the "
                                  "accelerated path; plot get_output() yourself")

    def plot_individual(self, *args, **kwargs):
        raise NotImplementedError("plot_individual (matplotlib figures, reference :348-461) is presentation code "
                                  "outside the accelerated path")


def sweep_over_betas(beta_values, n_runs=3, t_min=20.0, t_max=40.0, seeds=None, init_kwargs=None, **ctor_kwargs):
    """The reference's PDE tracer sweep (IMEX_PDE_solver_run_sweep.py:7-75) as ONE launch: every (beta, run) pair is a
    system with its own seeded initial condition.  Returns (v_mean, v_err, D_mean, D_err) per beta exactly as the
    driver forms them: v = |nanmean(v_eff_series[t_min <= t <= t_max])|, D = nanmean(D_eff_series[...]), mean over runs,
    err = std(ddof=1) / sqrt(n_runs)."""
    init_kwargs = dict(init_kwargs or {})
    betas, rp, rm, tx, ts = [], [], [], [], []
    proto = None
    for bi, beta in enumerate(beta_values):
        for run in range(n_runs):
            seed = run if seeds is None else seeds[bi][run]         # the reference seeds each run with its run index
            s = IMEXPDE(beta=beta, seed=seed, record_fft=False, **ctor_kwargs)
            s.initialize(**init_kwargs)
            proto = proto or s
            betas.append(float(beta)); rp.append(s.rho_p); rm.append(s.rho_m); tx.append(s.tracers_unwrapped); ts.append(s.tracer_state)
    r = solve_batch_raw(L=proto.L, xlim=proto.xlim, dt=proto.dt, nsteps=proto.nsteps, gamma=proto.gamma, lam=proto.lam, betas=betas,
                        bc=proto.bc, active_model=proto.active_model, gaussian_kernel=proto.gaussian_kernel,
                        kernel_sigma=proto.kernel_sigma, snapshot_interval=proto.snapshot_interval, rho_p0=np.array(rp),
                        rho_m0=np.array(rm), tracer_x0=np.array(tx), tracer_s0=np.array(ts), seed=proto.seed or 0,
                        device=proto.device, want_snapshots=False)
    t = np.linspace(0, proto.T, proto.nsteps + 1)
    mask = (t >= t_min) & (t <= t_max)
    v = np.abs(np.nanmean(r["v_eff_series"][:, mask], axis=1)).reshape(len(beta_values), n_runs)
    D = np.nanmean(r["D_eff_series"][:, mask], axis=1).reshape(len(beta_values), n_runs)
    root = np.sqrt(n_runs)
    return (v.mean(axis=1), v.std(axis=1, ddof=1) / root, D.mean(axis=1), D.std(axis=1, ddof=1) / root, r["kernel_ms"])
