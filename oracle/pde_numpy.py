"""ORACLE (test infrastructure; never shipped, never the thing measured as the product).

NumPy/SciPy restatement of the reference's hydrodynamic-limit solver `IMEXPDE`
(IMEX_PDE_solver_class.py:11-307): implicit diffusion (sparse solve), upwind advection, Curie-Weiss reaction,
clip + mass renormalisation, and the Euler-Maruyama tracer particles advanced inside `solve`.  It issues the
same legacy `np.random` calls in the same order as the reference (seed -> randn, randn, choice, choice; per
step rand, randn), so seeded runs are pinned BIT FOR BIT by fixture tests/golden/g6_pde.npz
(tests/test_oracle_pde.py).  Third-party arithmetic on the path, as in the reference: scipy.sparse.linalg.spsolve,
numpy.fft.rfft/irfft, numpy.random (legacy MT19937 global state).

Every function cites the reference lines it follows.  Only tests/, __graft_entry__.smoke() and bench.py's CPU
legs may import this module.
"""
from __future__ import annotations

import numpy as np
from numpy.fft import irfft, rfft
from scipy.sparse import diags
from scipy.sparse.linalg import spsolve


class PdeOracle:
    def __init__(self, L=1000, xlim=1.0, T=10.0, dt=5e-4, gamma=2.33e-4, lam=0.6, beta=2.0, bc="periodic",
                 active_model="bidirectional", gaussian_kernel=False, kernel_sigma=0.02, snapshot_interval=50, seed=None):
        # ref :13-61 (no output directory is created here)
        self.L, self.xlim, self.dx = L, xlim, xlim / L
        self.x = np.linspace(0, xlim, L, endpoint=False)
        self.T, self.dt, self.nsteps = T, dt, int(T / dt)
        self.gamma, self.lam, self.beta = gamma, lam, beta
        self.bc, self.active_model = bc, active_model
        self.gaussian_kernel, self.kernel_sigma = gaussian_kernel, kernel_sigma
        self.snapshot_interval = snapshot_interval
        if seed is not None:
            np.random.seed(seed)                                   # ref :55-56 (legacy global generator)
        lap = diags([1, -2, 1], [-1, 0, 1], shape=(L, L)).tolil()  # ref :68-82
        if bc == "periodic":
            lap[0, -1] = lap[-1, 0] = 1
        elif bc == "neumann":
            lap[0, 1] = 2
            lap[-1, -2] = 2
        self.A_diff = (diags(np.ones(L), 0) - gamma * dt * lap / self.dx ** 2).tocsr()
        if not gaussian_kernel:                                    # ref :84-93
            self.kernel_hat = None
        else:
            i = np.arange(L)
            dist = np.minimum(i, L - i) * self.dx
            kernel = np.exp(-0.5 * (dist / kernel_sigma) ** 2)
            kernel /= kernel.sum()
            self.kernel_hat = rfft(kernel)

    def cw_rate(self, sigma, m):                                   # ref :64-66
        return np.clip(np.exp(-self.beta * sigma * m), 1e-8, 1e8)

    def initialize(self, mode="poisson", rho0=1.0, noise=0.2, n_tracers=1000):   # ref :96-131
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
        n = self.nsteps + 1
        self.m_series, self.var_series = np.zeros(n), np.zeros(n)
        self.fft_amp = np.zeros((n, L // 2 + 1))
        self.fft_phase = np.zeros((n, L // 2 + 1), dtype=complex)
        self.snapshots, self.m_snapshots, self.times = [], [], []
        self.v_eff_series, self.D_eff_series = np.full(n, np.nan), np.full(n, np.nan)
        self.n_tracers = n_tracers
        self.tracers = np.random.choice(L, size=n_tracers) * self.dx
        self.tracers_unwrapped = self.tracers.copy()
        self.tracer_history = []
        self.tracer_state = np.random.choice([-1, 1], size=n_tracers)

    def magnetization(self):                                       # ref :156-168
        if self.kernel_hat is None:
            return (self.rho_p - self.rho_m) / (self.rho_p + self.rho_m + 1e-12)
        if self.kernel_sigma > 100000:
            return np.sum(self.rho_p - self.rho_m) / (np.sum(self.rho_p + self.rho_m) + 1e-12)
        num = irfft(rfft(self.rho_p - self.rho_m) * self.kernel_hat, n=self.L)
        den = irfft(rfft(self.rho_p + self.rho_m) * self.kernel_hat, n=self.L)
        return num / (den + 1e-12)

    def upwind(self, rho, direction):                              # ref :170-188
        d = np.zeros_like(rho)
        if direction > 0:
            d[1:] = (rho[1:] - rho[:-1]) / self.dx
            d[0] = 0.0 if self.bc == "neumann" else (rho[0] - rho[-1]) / self.dx
        else:
            d[:-1] = (rho[1:] - rho[:-1]) / self.dx
            d[-1] = 0.0 if self.bc == "neumann" else (rho[0] - rho[-1]) / self.dx
        return d

    def step(self):                                                # ref :190-233
        rho_p = spsolve(self.A_diff, self.rho_p)
        rho_m = spsolve(self.A_diff, self.rho_m)
        m = self.magnetization()                                   # of the state before the diffusion solve
        R_p = self.cw_rate(-1, m) * rho_m - self.cw_rate(+1, m) * rho_p
        if self.active_model == "bidirectional":
            adv_p = -self.lam * self.upwind(rho_p, +1)
            adv_m = +self.lam * self.upwind(rho_m, -1)
            self.rho_p = np.clip(rho_p + self.dt * (adv_p + R_p), 0, None)
            self.rho_m = np.clip(rho_m + self.dt * (adv_m - R_p), 0, None)
        else:                                                      # the reference solves the diffusion twice here; same result
            star_p = np.clip(rho_p + self.dt * R_p, 0, None)
            star_m = np.clip(rho_m + self.dt * (-R_p), 0, None)
            adv_p = -self.lam * self.upwind(star_p, +1)
            self.rho_p = np.clip(star_p + self.dt * adv_p, 0, None)
            self.rho_m = star_m
        M0 = (rho_p + rho_m).sum()
        M1 = (self.rho_p + self.rho_m).sum()
        self.rho_p *= M0 / M1
        self.rho_m *= M0 / M1

    def solve(self, record_randoms=False):                         # ref :236-290
        window = int(0.05 / self.dt)
        self.rand_u, self.rand_n = [], []
        for n in range(self.nsteps + 1):
            total = self.rho_p + self.rho_m
            self.m_series[n] = np.mean(self.magnetization())
            self.var_series[n] = np.var(total)
            spec = rfft(total) / self.L
            self.fft_amp[n], self.fft_phase[n] = np.abs(spec), spec
            if n % self.snapshot_interval == 0:
                self.snapshots.append(total.copy())
                self.m_snapshots.append(self.rho_p - self.rho_m)
                self.times.append(n * self.dt)
            m_field = self.magnetization()
            idx = (self.tracers / self.dx).astype(int) % self.L
            m_loc = m_field[idx] if np.ndim(m_field) else np.full(self.n_tracers, m_field)
            rate = np.where(self.tracer_state == +1, self.cw_rate(+1, m_loc), self.cw_rate(-1, m_loc))
            u = np.random.rand(self.n_tracers)
            self.tracer_state[u < rate * self.dt] *= -1
            g = np.random.randn(self.n_tracers)
            if record_randoms:
                self.rand_u.append(u)
                self.rand_n.append(g)
            noise = np.sqrt(2 * self.gamma * self.dt) * g
            self.tracers_unwrapped += self.lam * self.tracer_state * self.dt + noise
            self.tracers = self.tracers_unwrapped % self.xlim
            self.tracer_history.append(self.tracers_unwrapped.copy())
            if len(self.tracer_history) > window:
                dr = self.tracers_unwrapped - self.tracer_history[-window]
                mean_dr = np.mean(dr)
                self.v_eff_series[n] = mean_dr / (window * self.dt)
                self.D_eff_series[n] = np.mean((dr - mean_dr) ** 2) / (2 * window * self.dt)
            if n < self.nsteps:
                self.step()

    def get_output(self):                                          # ref :293-306
        return dict(rho_p=self.rho_p, rho_m=self.rho_m, m_series=self.m_series, var_series=self.var_series,
                    fft_amp=self.fft_amp, fft_phase=self.fft_phase, snapshots=np.array(self.snapshots),
                    m_snapshots=np.array(self.m_snapshots), times=np.array(self.times),
                    v_eff_series=self.v_eff_series, D_eff_series=self.D_eff_series)
