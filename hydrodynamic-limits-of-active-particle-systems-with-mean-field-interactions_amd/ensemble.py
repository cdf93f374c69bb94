"""Batched beta sweeps: the reference's `sweep_beta_ensemble` / `sweep_over_betas` loops
(PARTICLE_solver_BIOLOGY_EXCLUSION_sweep_beta.py:56-117, :828-1028) with every (beta, run) pair stepped
together as independent ensembles of one GPU handle, and the per-run observables of observables.py.
The two outer sweeps of the other drivers are here too: `sweep_over_sigmas` (interaction range,
..._sweep_beta_2.py:1030-1075) and `sweep_over_densities` (particle number x beta, ..._double_sweep.py:851-861) --
one batched handle per sigma / per particle number (the weight table and the state capacity differ), all (beta, run)
pairs inside it."""
from __future__ import annotations

import numpy as np

from . import observables
from .particle_system import ParticleSystem, run_batched, run_batched_statistics


def sweep_over_betas(beta_values, n_runs_per_beta=10, ps_kwargs=None, init_kwargs=None, run_kwargs=None,
                     rng_seeds=None, keep_outputs=False, on_device=False, dynamics=None):
    """Returns a dict with the keys the reference saves (`beta_values, means, stds, ses, D_means, D_ses, m_means,
    m_stds, m_ses, rho_means, rho_ses, block_means, block_ses`, ..._sweep_beta.py:952-968) plus `raw_by_beta`.
    `rng_seeds[b][r]` seeds the initial condition of run r at beta b (None: unseeded, like the reference).
    `on_device=True` evaluates the observables from integer sums taken on the GPU at each observation time
    (run_batched_statistics; needs k_exit = 0) instead of from the M x L arrays of `run()`; `run_kwargs` may then
    only hold T and obs_dt.  `dynamics="exact"` runs the reference's event-by-event dynamics resident on the GPU
    (gillespie.run_batched_exact / run_batched_exact_statistics), `"sync"` the fixed-dt scheme; the default follows
    ParticleSystem's: exact unless `ps_kwargs` asks for the stepper (`dt` or `mode="sync"`)."""
    ps_kwargs, init_kwargs, run_kwargs = dict(ps_kwargs or {}), dict(init_kwargs or {}), dict(run_kwargs or {})
    if dynamics is None:
        dynamics = "sync" if (ps_kwargs.get("dt") is not None or ps_kwargs.get("mode") == "sync") else "exact"
    systems, owner = [], []
    for bi, beta in enumerate(beta_values):
        for r in range(n_runs_per_beta):
            rng = None if rng_seeds is None else np.random.default_rng(int(rng_seeds[bi][r]))
            systems.append(ParticleSystem(beta=beta, rng=rng, **ps_kwargs, **init_kwargs))
            owner.append(bi)
    if dynamics not in ("sync", "exact"):
        raise ValueError("dynamics must be 'sync' or 'exact'")
    if on_device:
        if keep_outputs:
            raise ValueError("on_device=True keeps no per-run outputs")
        outs = None
        slim = {k: v for k, v in run_kwargs.items() if k in ("T", "obs_dt")}
        if dynamics == "exact":
            from .gillespie import run_batched_exact_statistics
            rows = run_batched_exact_statistics(systems, **slim)
        else:
            rows = run_batched_statistics(systems, **slim)
    elif dynamics == "exact":
        from .gillespie import run_batched_exact
        outs = run_batched_exact(systems, want_m_local=False, **run_kwargs)
        rows = [observables.run_observables(out, ps.L, ps.dx) for ps, out in zip(systems, outs)]
    else:
        outs = run_batched(systems, **run_kwargs)
        rows = [observables.run_observables(out, ps.L, ps.dx) for ps, out in zip(systems, outs)]
    res = {k: [] for k in ("means", "stds", "ses", "D_means", "D_ses", "m_means", "m_stds", "m_ses", "rho_means", "rho_ses",
                           "block_means", "block_ses", "raw_by_beta")}
    for bi in range(len(beta_values)):
        st = observables.ensemble_statistics([row for row, o in zip(rows, owner) if o == bi])
        for src, dst in (("mean", "means"), ("std", "stds"), ("se", "ses"), ("D_mean", "D_means"), ("D_se", "D_ses"),
                         ("m_mean", "m_means"), ("m_std", "m_stds"), ("m_se", "m_ses"), ("rho_mean", "rho_means"),
                         ("rho_se", "rho_ses"), ("block_mean", "block_means"), ("block_se", "block_ses"), ("v_array", "raw_by_beta")):
            res[dst].append(st[src])
    out = {k: (np.array(v) if k != "raw_by_beta" else v) for k, v in res.items()}
    out["beta_values"] = np.asarray(beta_values, dtype=float)
    if keep_outputs:
        out["outs"] = outs
    return out


def sweep_beta_ensemble(beta, n_runs=10, ps_kwargs=None, init_kwargs=None, run_kwargs=None, rng_seeds=None):
    """One beta, n_runs batched runs; same return tuple as the reference function (:117)."""
    r = sweep_over_betas([beta], n_runs, ps_kwargs, init_kwargs, run_kwargs,
                         None if rng_seeds is None else [rng_seeds], keep_outputs=True)
    return (float(r["means"][0]), float(r["stds"][0]), float(r["ses"][0]), r["raw_by_beta"][0], r["outs"],
            float(r["m_means"][0]), float(r["m_stds"][0]), float(r["m_ses"][0]), float(r["rho_means"][0]),
            float(r["rho_ses"][0]), float(r["block_means"][0]), float(r["block_ses"][0]), float(r["D_means"][0]),
            float(r["D_ses"][0]))


def sweep_over_sigmas(sigma_values, beta_values, n_runs_per_beta=5, ps_kwargs=None, init_kwargs=None, run_kwargs=None,
                      rng_seeds=None, on_device=False, dynamics=None):
    """The sigma sweep of PARTICLE_solver_BIOLOGY_EXCLUSION_sweep_beta_2.py:1030-1075: for every interaction range
    `local_kernel_sigma` a whole beta sweep (one GPU handle per sigma: the weight table changes; sigma = 0 selects the global
    mean field, sigma wider than the box the folded table).  Returns {sigma: {"beta", "v_mean", "v_se", "D_mean", "D_se",
    "ps_kwargs"}} like the reference (which also writes one .npz per sigma; saving is left to the caller)."""
    results = {}
    for sigma in sigma_values:
        kw = dict(ps_kwargs or {}, local_kernel_sigma=float(sigma))
        r = sweep_over_betas(beta_values, n_runs_per_beta, kw, init_kwargs, run_kwargs, rng_seeds, on_device=on_device, dynamics=dynamics)
        results[sigma] = {"beta": np.asarray(beta_values, dtype=float), "v_mean": r["means"], "v_se": r["ses"], "D_mean": r["D_means"],
                          "D_se": r["D_ses"], "m_mean": r["m_means"], "block_mean": r["block_means"], "ps_kwargs": kw}
    return results


def sweep_over_densities(n_part_values, beta_values, n_runs_per_beta=4, ps_kwargs=None, init_kwargs=None, run_kwargs=None,
                         rng_seeds=None, on_device=False, dynamics=None):
    """The density x beta double sweep of PARTICLE_solver_BIOLOGY_EXCLUSION_double_sweep.py:851-861
    (`list_N_part = np.linspace(50, 950, 19)`: N arrives as a float there and is used as an integer): one batched beta sweep
    per particle number.  Returns a list of the per-N sweep dictionaries (keys of `sweep_over_betas`) with "N_part" added;
    the reference's fit of the blocking coefficients f, g to them (`rho_model`, :290-317) is closed-form SciPy on these
    numbers and stays with the caller."""
    out = []
    for n_part in n_part_values:
        if float(n_part) != int(n_part):
            raise ValueError("particle numbers must be integral")
        ik = dict(init_kwargs or {}, N=int(n_part))
        r = sweep_over_betas(beta_values, n_runs_per_beta, ps_kwargs, ik, run_kwargs, rng_seeds, on_device=on_device, dynamics=dynamics)
        r["N_part"] = int(n_part)
        out.append(r)
    return out
