"""Batched beta sweeps: the reference's `sweep_beta_ensemble` / `sweep_over_betas` loops
(PARTICLE_solver_BIOLOGY_EXCLUSION_sweep_beta.py:56-117, :828-1028) with every (beta, run) pair stepped
together as independent ensembles of one GPU handle, and the per-run observables of observables.py."""
from __future__ import annotations

import numpy as np

from . import observables
from .particle_system import ParticleSystem, run_batched, run_batched_statistics


def sweep_over_betas(beta_values, n_runs_per_beta=10, ps_kwargs=None, init_kwargs=None, run_kwargs=None,
                     rng_seeds=None, keep_outputs=False, on_device=False, dynamics="sync"):
    """Returns a dict with the keys the reference saves (`beta_values, means, stds, ses, D_means, D_ses, m_means,
    m_stds, m_ses, rho_means, rho_ses, block_means, block_ses`, ..._sweep_beta.py:952-968) plus `raw_by_beta`.
    `rng_seeds[b][r]` seeds the initial condition of run r at beta b (None: unseeded, like the reference).
    `on_device=True` evaluates the observables from integer sums taken on the GPU at each observation time
    (run_batched_statistics; needs k_exit = 0) instead of from the M x L arrays of `run()`; `run_kwargs` may then
    only hold T and obs_dt.  `dynamics="exact"` runs the reference's event-by-event dynamics resident on the GPU
    (gillespie.run_batched_exact / run_batched_exact_statistics) instead of the fixed-dt scheme."""
    ps_kwargs, init_kwargs, run_kwargs = dict(ps_kwargs or {}), dict(init_kwargs or {}), dict(run_kwargs or {})
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
