"""Ensemble observables of the reference's sweep drivers, restated (host side, NumPy).

These are the quantities the reference's statistics are judged on; they consume the `out`
dictionary of `ParticleSystem.run`.  Reference map (file = PARTICLE_solver_BIOLOGY_EXCLUSION_sweep_beta.py):
    velocity_and_window      <- compute_v_eff_and_window      :123-162
    front_density            <- compute_rho_eff               :165-194
    blocking_probability     <- compute_blocking_probability  :197-229   (vectorised; the reference loops in Python)
    mean_magnetisation       <- compute_mean_magnetizatoin    :316-319
    active_diffusivity       <- compute_D_eff_active          :500-525
    ensemble_statistics      <- the reduction at the end of sweep_beta_ensemble :97-117
    structure_observables    <- extract_structure_observables_from_out of PARTICLE_solver_BIOLOGY_local_structure.py:55-103
Pinned by fixtures tests/golden/g7_observables.npz and g10_structure.npz (generated from the reference's own functions).
"""
from __future__ import annotations

import numpy as np


def velocity_and_window(out, L, boundary_xmin=0.99, max_boundary_fraction=0.06, min_window_fraction=0.10):
    """Centre-of-mass velocity and the averaging window [start, end).

    The window logic reproduces the reference as written: frames with too much mass at the right wall are
    collected as INDICES, and the reference then bit-inverts a slice of that index array, which is non-zero
    for every element -- so if any flagged index sits at position >= start of that array the window closes at
    `start` and is re-opened to the minimum length; otherwise it runs to the end."""
    times = out["times_obs"]
    total = out["total_list"]
    M = total.shape[0]
    grid = np.linspace(0.0, 1.0, total.shape[1])
    dx = grid[1] - grid[0]
    at_wall = total[:, grid >= boundary_xmin].sum(axis=1) * dx
    mass = total.sum(axis=1) * dx
    frac_boundary = at_wall / (mass + 1e-12)
    flagged = np.flatnonzero(frac_boundary >= max_boundary_fraction)
    start = int(0.65 * M)
    if flagged.size == 0:
        end = M
    else:
        end = M if flagged[start:].size == 0 else start
        shortest = max(3, int(min_window_fraction * M))
        if end - start < shortest:
            end = min(M, start + shortest)
    grid = np.linspace(0.0, 1.0, L)
    com = (total * grid).sum(axis=1) / (total.sum(axis=1) + 1e-12)
    v_eff = np.gradient(com, times)
    return float(np.mean(v_eff[start:end])), v_eff, times, start, end, frac_boundary


def front_density(out, start, end, window_fraction=0.05):
    total = np.asarray(out["total_list"])
    grid = np.linspace(0.0, 1.0, total.shape[1])
    dx = grid[1] - grid[0]
    vals = []
    for t in range(start, end):
        occupied = np.flatnonzero(total[t] > 0)
        if occupied.size == 0:
            continue
        x_front = grid[occupied[-1]]
        window = (grid >= x_front - window_fraction) & (grid <= x_front)
        if window.any():
            vals.append(total[t][window].sum() * dx / window_fraction)
    return float(np.mean(vals))


def blocking_probability(out, start, end):
    """Share of the + density whose right neighbour site carries total density >= 1 (reference units)."""
    total = np.asarray(out["total_list"])[start:end]
    plus = np.asarray(out["rho_p_list"])[start:end]
    movers = np.where(plus[:, :-1] > 0, plus[:, :-1], 0.0)
    attempts = movers.sum()
    if attempts == 0:
        return 0.0
    return float((movers * (total[:, 1:] >= 1.0)).sum() / attempts)


def mean_magnetisation(out, start, end):
    return float(np.mean(np.asarray(out["m_global"], dtype=float)[start:end]))


def active_diffusivity(out, dx, start, end):
    times, frames = out["times_obs"], out["pos_list"]
    ref, t_ref = frames[start] * dx, times[start]
    spread, lag = [], []
    for k in range(start + 1, end):
        cur = frames[k] * dx
        n = min(len(ref), len(cur))
        if n < 2:
            continue
        disp = cur[:n] - ref[:n]
        spread.append(np.sum((disp - np.mean(disp)) ** 2) / (n - 1))
        lag.append(times[k] - t_ref)
    return np.polyfit(lag, spread, 1)[0]


def run_observables(out, L, dx):
    """All five per-run observables with the reference's default window parameters."""
    v, _, _, start, end, _ = velocity_and_window(out, L)
    return dict(v=v, D=active_diffusivity(out, dx, start, end), m=mean_magnetisation(out, start, end),
                rho=front_density(out, start, end), block=blocking_probability(out, start, end), window=(start, end))


def ensemble_statistics(rows):
    """Mean / sample standard deviation / standard error over runs, keyed like the reference's savez arrays
    (..._sweep_beta.py:952-968: means, stds, ses, D_means, D_ses, m_means, m_stds, m_ses, rho_means, rho_ses,
    block_means, block_ses)."""
    n = len(rows)
    col = {k: np.array([r[k] for r in rows], dtype=float) for k in ("v", "D", "m", "rho", "block")}
    sd = {k: (float(a.std(ddof=1)) if n > 1 else 0.0) for k, a in col.items()}
    root = np.sqrt(max(1, n))
    return dict(mean=float(col["v"].mean()), std=sd["v"], se=sd["v"] / root, v_array=col["v"],
                D_mean=float(col["D"].mean()), D_se=sd["D"] / root,
                m_mean=float(col["m"].mean()), m_std=sd["m"], m_se=sd["m"] / root,
                rho_mean=float(col["rho"].mean()), rho_se=sd["rho"] / root,
                block_mean=float(col["block"].mean()), block_se=sd["block"] / root)


# ---------------------------------------------------------------------------------------------------------------
# The same five observables from the device-side integer sums (aps_observe_scalars): no M x L arrays leave the GPU.
class DeviceObservables:
    """Per-run accumulator.  `plan(k)` tells the stepping loop what to ask the device for at observation k,
    `add(k, sums, front)` stores the answer, `result()` evaluates the reference's formulas
    (PARTICLE_solver_BIOLOGY_EXCLUSION_sweep_beta.py:123-229, :316-319, :500-525) on the stored sums.
    Exact for the integer parts; the float parts differ from the array formulas only by summation order."""

    def __init__(self, times, L, dx, K, boundary_xmin=0.99, max_boundary_fraction=0.06, min_window_fraction=0.10,
                 front_window_fraction=0.05):
        self.times, self.L, self.dx, self.K = np.asarray(times, dtype=float), int(L), float(dx), int(K)
        self.M = len(self.times)
        self.grid = np.linspace(0.0, 1.0, self.L)
        self.gdx = self.grid[1] - self.grid[0]
        self.x_wall = int(np.searchsorted(self.grid, boundary_xmin, side="left"))
        self.max_boundary_fraction, self.min_window_fraction = max_boundary_fraction, min_window_fraction
        self.wf = front_window_fraction
        self.start = int(0.65 * self.M)
        self.rows = [None] * self.M
        self.front = [None] * self.M

    def block_table(self, n_live):
        """Is a right neighbour with (plus, minus) particles 'blocking'?  total density >= 1 in the reference's units
        rho = count / (N_now * dx), evaluated with the reference's float operations."""
        denom = float(max(1, n_live)) * self.dx
        cp, cm = np.meshgrid(np.arange(self.K + 1), np.arange(self.K + 1), indexing="ij")
        return ((cp / denom).astype(float) + (cm / denom).astype(float) >= 1.0).astype(np.uint8)

    def front_range(self, max_pos):
        """Site range [lo, hi] of the reference's front window below the right-most occupied site."""
        x_front = self.grid[max_pos]
        lo = int(np.searchsorted(self.grid, x_front - self.wf, side="left"))
        return lo, int(max_pos)

    def add(self, k, sums, n_front=None):
        self.rows[k] = dict(sums)
        self.front[k] = n_front

    def result(self):
        M, dx, gdx = self.M, self.dx, self.gdx
        n = np.array([r["n"] for r in self.rows], dtype=float)
        dens = 1.0 / (np.maximum(n, 1.0) * dx)                     # one particle in the reference's density units
        mass = n * dens * gdx
        frac_boundary = np.array([r["n_wall"] for r in self.rows]) * dens * gdx / (mass + 1e-12)
        flagged = np.flatnonzero(frac_boundary >= self.max_boundary_fraction)
        start = self.start
        if flagged.size == 0:
            end = M
        else:
            end = M if flagged[start:].size == 0 else start
            shortest = max(3, int(self.min_window_fraction * M))
            if end - start < shortest:
                end = min(M, start + shortest)
        com = (np.array([r["sum_pos"] for r in self.rows]) / (self.L - 1.0)) * dens / (n * dens + 1e-12)
        v_ts = np.gradient(com, self.times)
        m_glob = np.array([r["sum_sigma"] / r["n"] if r["n"] else np.nan for r in self.rows])
        fronts = [self.front[k] * dens[k] * gdx / self.wf for k in range(start, end) if self.front[k] is not None]
        att = sum(self.rows[k]["attempts"] * dens[k] for k in range(start, end))
        blk = sum(self.rows[k]["blocked"] * dens[k] for k in range(start, end))
        spread, lag = [], []
        for k in range(start + 1, end):
            r = self.rows[k]
            nd = r["n_d"]
            if nd < 2:
                continue
            spread.append(dx * dx * (r["sum_d2"] - r["sum_d"] ** 2 / nd) / (nd - 1))
            lag.append(self.times[k] - self.times[start])
        D = np.polyfit(lag, spread, 1)[0] if len(lag) >= 2 else float("nan")
        return dict(v=float(np.mean(v_ts[start:end])), D=float(D), m=float(np.mean(m_glob[start:end])),
                    rho=float(np.mean(fronts)) if fronts else float("nan"), block=float(blk / att) if att else 0.0,
                    window=(start, end), v_ts=v_ts, frac_boundary=frac_boundary, m_global=m_glob)


# ------------------------------------------------------------------------------------------------ structure observables
def structure_observables(out, start_fraction=0.5, k_max=None):
    """Pattern / clustering observables of a run made with record_fft=True, record_var=True
    (PARTICLE_solver_BIOLOGY_local_structure.py:55-103): time mean and spread of var(total) and of the Fourier
    amplitudes |fft(total)| over the steady-state window [start_fraction * M, M), the dominant non-zero mode, the summed
    low-k amplitudes (k = 1..24), the variance of the local magnetisation over window x lattice and the mean low-k power."""
    M = len(out["times_obs"])
    start = int(start_fraction * M)
    var_ts = np.asarray(out["var_list"], dtype=float)[start:]
    amp = np.asarray(out["fft_amp_list"], dtype=float)
    if k_max is not None:
        amp = amp[:, :k_max]
    ss = amp[start:]
    fft_mean, fft_std = ss.mean(axis=0), ss.std(axis=0, ddof=1)
    cut = min(25, amp.shape[1])
    m_ss = np.asarray(out["m_local_list"], dtype=float)[start:]
    return {"var_mean": var_ts.mean(), "var_std": var_ts.std(ddof=1), "fft_mean": fft_mean, "fft_std": fft_std,
            "dominant_k": int(np.argmax(fft_mean[1:]) + 1), "low_k_power": float(np.sum(fft_mean[1:cut])),
            "m_local_var": float(np.var(m_ss)), "lowk_variance": float(np.mean(np.sum(ss[:, 1:cut] ** 2, axis=1)))}


extract_structure_observables_from_out = structure_observables      # the reference's name


class DeviceStructure:
    """The same eight observables accumulated from what the GPU returns per observation (aps_observe_structure): the live
    particle number n, sum over sites of count^2, sum and sum of squares of the local magnetisation over the L sites, and the
    first k_max Fourier sums of the site histogram.  Nothing of size M x L ever leaves the device.

    total = count / (n dx) (ref :205-213), so var(total) = (sum c^2 / L - (n / L)^2) / (n dx)^2 and
    |fft(total)|_k = |sum_x c_x exp(-2 pi i k x / L)| / (n dx)."""

    def __init__(self, n_obs, L, dx, start_fraction=0.5, k_max=None):
        self.M, self.L, self.dx = int(n_obs), int(L), float(dx)
        self.start = int(start_fraction * self.M)
        self.k_max = self.L if k_max is None else min(int(k_max), self.L)
        self.var, self.amp, self.m1, self.m2 = [], [], 0.0, 0.0
        self.nm = 0

    def add(self, k, n_live, sum_c2, sum_m, sum_m2, re_im):
        if k < self.start:
            return
        L, nd = self.L, float(n_live) * self.dx
        self.var.append((sum_c2 / L - (n_live / L) ** 2) / (nd * nd) if n_live else np.nan)
        z = np.asarray(re_im, dtype=float).reshape(-1, 2)
        self.amp.append(np.hypot(z[:, 0], z[:, 1]) / nd if n_live else np.full(len(z), np.nan))
        self.m1 += sum_m
        self.m2 += sum_m2
        self.nm += L

    def result(self):
        var, amp = np.array(self.var), np.array(self.amp)
        fft_mean, fft_std = amp.mean(axis=0), amp.std(axis=0, ddof=1)
        cut = min(25, amp.shape[1])
        mean_m = self.m1 / self.nm
        return {"var_mean": var.mean(), "var_std": var.std(ddof=1), "fft_mean": fft_mean, "fft_std": fft_std,
                "dominant_k": int(np.argmax(fft_mean[1:]) + 1), "low_k_power": float(np.sum(fft_mean[1:cut])),
                "m_local_var": float(self.m2 / self.nm - mean_m * mean_m),
                "lowk_variance": float(np.mean(np.sum(amp[:, 1:cut] ** 2, axis=1)))}
