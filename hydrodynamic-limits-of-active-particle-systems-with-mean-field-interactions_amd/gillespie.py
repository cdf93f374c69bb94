"""Device-resident exact Gillespie loop for batches of systems (include/gillespie.h): the reference's
`ParticleSystem.run` as written (one event per iteration, PARTICLE_solver_CLASS.py:450-558), one persistent workgroup
per system.  `run_batched_exact` returns the reference's result dictionaries; `sweep` statistics can be taken from the
scalar sums without the M x L arrays (`scalars_only=True`).

Differences to the reference: randomness is Philox4x32-10 keyed by `seed` (the reference consumes a NumPy Generator), so
trajectories agree in distribution, not draw for draw; `m_local_list[k]` is the field of the observed state (the
reference stores the field from before the last event).  There is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

GIL_MAX_L, GIL_MAX_N, NSCALARS = 4096, 2048, 12
SCALARS = ("n", "sum_sigma", "sum_pos", "n_wall", "max_pos", "n_front", "attempts", "blocked", "sum_d", "sum_d2", "n_d", "events")


class GilParams(C.Structure):
    """struct gil_params of include/gillespie.h, field for field."""
    _fields_ = [("L", C.c_int32), ("K", C.c_int32), ("periodic", C.c_int32), ("minus_anchor", C.c_int32),
                ("immobilize", C.c_int32), ("suppress_flip", C.c_int32), ("crowding", C.c_int32), ("n_systems", C.c_int32),
                ("n_cap", C.c_int32), ("n_obs", C.c_int32), ("device", C.c_int32), ("x_wall", C.c_int32),
                ("ref_obs", C.c_int32), ("flip_n", C.c_int32), ("sigma_grid", C.c_double), ("rate_diffusion", C.c_double),
                ("rate_active", C.c_double), ("k_on", C.c_double), ("k_off", C.c_double), ("k_exit", C.c_double),
                ("T", C.c_double), ("seed", C.c_uint64), ("max_events", C.c_int64), ("beta", C.c_void_p),
                ("anchor_mask", C.c_void_p), ("times_obs", C.c_void_p), ("front_lo", C.c_void_p), ("block_table", C.c_void_p),
                ("flip_table", C.c_void_p)]


def _lib():
    lib = capi.load()
    if not getattr(lib, "_gil_ready", False):
        lib.gil_last_error.restype, lib.gil_last_error.argtypes = C.c_char_p, []
        lib.gil_run_batch.restype = C.c_int
        lib.gil_run_batch.argtypes = [C.POINTER(GilParams)] + [C.c_void_p] * 14 + [C.POINTER(C.c_double)]
        lib.gil_large_last_error.restype, lib.gil_large_last_error.argtypes = C.c_char_p, []
        lib.gil_run_large.restype = C.c_int
        lib.gil_run_large.argtypes = [C.POINTER(GilParams), C.c_int32] + [C.c_void_p] * 12 + [C.POINTER(C.c_double)]
        lib._gil_ready = True
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run_raw(*, L, K, periodic, sigma_grid, rate_diffusion, rate_active, betas, states, times_obs, T, seed=0,
            minus_anchor=True, immobilize=True, suppress_flip=True, crowding=False, k_on=0.0, k_off=0.0, k_exit=0.0,
            anchor_mask=None, uniforms=None, max_events=None, want_states=True, x_wall=0, ref_obs=-1, front_lo=None,
            block_table=None, device=0, flip_table=None):
    """`states` = list of (pos, sigma[, bound]) per system.  Returns a dict of arrays with a leading system axis."""
    lib = _lib()
    S = len(states)
    betas = np.ascontiguousarray(np.broadcast_to(np.asarray(betas, dtype=np.float64), (S,)))
    ncap = max(1, max(len(st[0]) for st in states))
    n0 = np.array([len(st[0]) for st in states], np.int32)
    pos0, sg0, bd0 = np.zeros((S, ncap), np.int32), np.ones((S, ncap), np.int8), np.zeros((S, ncap), np.uint8)
    for s, st in enumerate(states):
        pos0[s, :n0[s]], sg0[s, :n0[s]] = st[0], st[1]
        if len(st) > 2 and st[2] is not None:
            bd0[s, :n0[s]] = st[2]
    times = np.ascontiguousarray(times_obs, dtype=np.float64)
    M = len(times)
    if uniforms is not None:
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        assert uniforms.shape[0] == S and uniforms.shape[2] == 4
        max_events = uniforms.shape[1]
    elif max_events is None:
        max_events = 2 ** 40
    mask = None if anchor_mask is None or not np.any(anchor_mask) else np.ascontiguousarray(anchor_mask, dtype=np.uint8)
    flo = None if front_lo is None else np.ascontiguousarray(front_lo, dtype=np.int32)
    btab = None if block_table is None else np.ascontiguousarray(block_table, dtype=np.uint8)
    ftab = None if flip_table is None else np.ascontiguousarray(flip_table, dtype=np.float64)      # [2][n + 1] (aps_set_flip_table)
    par = GilParams(flip_n=0 if ftab is None else ftab.shape[1] - 1, flip_table=None if ftab is None else _p(ftab).value, L=L, K=K, periodic=int(bool(periodic)), minus_anchor=int(bool(minus_anchor)), immobilize=int(bool(immobilize)),
                    suppress_flip=int(bool(suppress_flip)), crowding=int(bool(crowding)), n_systems=S, n_cap=ncap, n_obs=M,
                    device=device, x_wall=int(x_wall), ref_obs=int(ref_obs), sigma_grid=float(sigma_grid),
                    rate_diffusion=float(rate_diffusion), rate_active=float(rate_active), k_on=float(k_on), k_off=float(k_off),
                    k_exit=float(k_exit), T=float(T), seed=int(seed) & (2 ** 64 - 1), max_events=int(max_events),
                    beta=_p(betas).value, anchor_mask=None if mask is None else _p(mask).value, times_obs=_p(times).value,
                    front_lo=None if flo is None else _p(flo).value, block_table=None if btab is None else _p(btab).value)
    pos_obs = np.zeros((S, M, ncap), np.int32) if want_states else None
    sg_obs = np.zeros((S, M, ncap), np.int8) if want_states else None
    fl_obs = np.zeros((S, M, ncap), np.uint8) if want_states else None
    scal = np.zeros((S, M, NSCALARS), np.int64)
    n_rec, n_ev, t_fin = np.zeros(S, np.int32), np.zeros(S, np.int64), np.zeros(S)
    exits, n_exit = np.zeros((S, ncap, 3)), np.zeros(S, np.int32)
    ms = C.c_double()
    rc = lib.gil_run_batch(C.byref(par), _p(n0), _p(pos0), _p(sg0), _p(bd0), _p(uniforms), _p(pos_obs), _p(sg_obs), _p(fl_obs),
                           _p(scal), _p(n_rec), _p(n_ev), _p(t_fin), _p(exits), _p(n_exit), C.byref(ms))
    if rc != 0:
        raise capi.ApsError(rc, lib.gil_last_error().decode())
    return dict(pos=pos_obs, sigma=sg_obs, flags=fl_obs, scalars=scal, n_recorded=n_rec, n_events=n_ev, t_final=t_fin,
                exits=exits, n_exits=n_exit, n0=n0, kernel_ms=ms.value)


def run_batched_exact(systems, T=10.0, obs_dt=0.01, record_fft=False, record_var=False, uniforms=None, want_m_local=True):
    """`run()` of several ParticleSystem objects with the reference's exact event-by-event dynamics, all systems at
    once on the GPU.  They may differ in beta, rng / initial condition and particle number only.  Returns the list of
    result dictionaries (reference :542-557).  Observations the loop never reached (t passed T first, ref :515-516)
    keep the reference's pre-allocated zeros / None.  want_m_local=False leaves m_local_list zero (saves one field
    evaluation per observation and system)."""
    from .particle_system import ParticleSystem, _SHAPE_ATTRS
    first = systems[0]
    for ps in systems[1:]:
        for k in _SHAPE_ATTRS:
            if getattr(ps, k) != getattr(first, k):
                raise ValueError(f"run_batched_exact: systems differ in {k}")
    L, dx = first.L, first.dx
    inits = [ps.init_particles() for ps in systems]
    seed = first.seed if first.seed is not None else int(first.rng.random() * 2.0 ** 53)
    times_obs = np.arange(0.0, T, obs_dt)
    M = len(times_obs)
    if L > GIL_MAX_L or max(len(p) for p, _ in inits) > GIL_MAX_N:
        # beyond one workgroup's LDS: the large-system kernel (one system per launch, state in global memory)
        if uniforms is not None:
            raise ValueError("run_batched_exact: caller-supplied uniforms with large systems go through run_large_raw")
        parts = []
        for s, (ps, st) in enumerate(zip(systems, inits)):
            one = run_large_raw(L=L, K=first.K, periodic=first.periodic, sigma_grid=first._sigma_grid, rate_diffusion=first.rate_diffusion,
                                rate_active=first.rate_active, beta=float(ps.beta), state=st, times_obs=times_obs, T=T, seed=seed + s,
                                minus_anchor=first.minus_anchor, immobilize=first.immobilize_when_anchored,
                                suppress_flip=first.suppress_flip_when_bound, crowding=first.crowding_suppresses_rates, k_on=first.k_on,
                                k_off=first.k_off, k_exit=first.k_exit, anchor_mask=first.is_anchor_site, device=first.device,
                                flip_table=first.flip_table())
            parts.append(one)
        ncap = max(p["pos"].shape[1] for p in parts)

        def stack(key, dtype):
            out = np.zeros((len(parts), M, ncap), dtype)
            for s, p in enumerate(parts):
                out[s, :, :p[key].shape[1]] = p[key]
            return out
        exits = np.zeros((len(parts), ncap, 3))
        for s, p in enumerate(parts):
            exits[s, :p["exits"].shape[0]] = p["exits"]
        r = dict(pos=stack("pos", np.int32), sigma=stack("sigma", np.int8), flags=stack("flags", np.uint8),
                 n_recorded=np.array([p["n_recorded"] for p in parts]), n_events=np.array([p["n_events"] for p in parts]),
                 t_final=np.array([p["t_final"] for p in parts]), exits=exits, n_exits=np.array([p["n_exits"] for p in parts]),
                 n0=np.array([p["n0"] for p in parts]), kernel_ms=sum(p["kernel_ms"] for p in parts))
    else:
        r = run_raw(L=L, K=first.K, periodic=first.periodic, sigma_grid=first._sigma_grid, rate_diffusion=first.rate_diffusion,
                rate_active=first.rate_active, betas=[float(ps.beta) for ps in systems], states=inits, times_obs=times_obs, T=T,
                seed=seed, minus_anchor=first.minus_anchor, immobilize=first.immobilize_when_anchored,
                suppress_flip=first.suppress_flip_when_bound, crowding=first.crowding_suppresses_rates, k_on=first.k_on,
                k_off=first.k_off, k_exit=first.k_exit, anchor_mask=first.is_anchor_site, uniforms=uniforms, device=first.device,
                flip_table=first.flip_table())
    outs = []
    for s, ps in enumerate(systems):
        n0 = int(r["n0"][s])
        pos_list, count, bound_list = [None] * M, [None] * M, [None] * M
        rho_p, rho_m, total, m_loc, m_glob = np.zeros((M, L)), np.zeros((M, L)), np.zeros((M, L)), np.zeros((M, L)), np.zeros(M)
        hat = np.zeros((M, L), dtype=complex) if record_fft else None
        amp = np.zeros((M, L)) if record_fft else None
        var = np.zeros(M) if record_var else None
        for k in range(int(r["n_recorded"][s])):
            fl = r["flags"][s, k, :n0]
            live = (fl & 2) != 0
            p, sg = r["pos"][s, k, :n0][live].astype(np.int64), r["sigma"][s, k, :n0][live]
            pos_list[k], count[k], bound_list[k] = p, p.size, (fl[live] & 1).astype(bool)
            a, b = ParticleSystem.empirical_densities_from_particles(p, sg, L, dx)
            rho_p[k], rho_m[k], total[k] = a, b, a + b
            if want_m_local:                                       # field of the observed state, on the GPU (aps_field_from_counts)
                m_loc[k] = ps.compute_local_m_field(np.bincount(p[sg == 1], minlength=L), np.bincount(p[sg == -1], minlength=L))
            m_glob[k] = np.mean(sg) if sg.size else np.nan
            if record_fft:
                spec = np.fft.fft(total[k])
                hat[k], amp[k] = spec, np.abs(spec)
                if record_var:
                    var[k] = float(np.var(total[k]))
        ex = r["exits"][s, :int(r["n_exits"][s])]
        ps.n_events = int(r["n_events"][s])
        outs.append({"times_obs": times_obs.copy(), "pos_list": pos_list, "rho_p_list": rho_p, "rho_m_list": rho_m,
                     "total_list": total, "particle_count_list": count, "bound_list": bound_list, "m_local_list": m_loc,
                     "m_global": m_glob, "rho_hat_complex": hat, "fft_amp_list": amp, "var_list": var,
                     "exit_times": [float(t) for t in ex[:, 0]], "exit_positions": [int(x) for x in ex[:, 1]]})
    first.kernel_ms = r["kernel_ms"]
    return outs


def run_batched_exact_statistics(systems, T=10.0, obs_dt=0.01):
    """The sweep drivers' per-run observables (observables.DeviceObservables: v_eff, D_eff, mean magnetisation, front
    density, blocking probability) for many systems under the exact dynamics, from the integer sums the event-loop
    kernel records at every observation -- no state arrays leave the GPU.  Needs k_exit = 0 like every reference sweep."""
    from . import observables
    from .particle_system import _SHAPE_ATTRS
    first = systems[0]
    for ps in systems[1:]:
        for k in _SHAPE_ATTRS:
            if getattr(ps, k) != getattr(first, k):
                raise ValueError(f"run_batched_exact_statistics: systems differ in {k}")
    if first.k_exit:
        raise ValueError("run_batched_exact_statistics needs k_exit = 0")
    inits = [ps.init_particles() for ps in systems]
    seed = first.seed if first.seed is not None else int(first.rng.random() * 2.0 ** 53)
    times_obs = np.arange(0.0, T, obs_dt)
    acc0 = observables.DeviceObservables(times_obs, first.L, first.dx, first.K)
    tables = [acc0.block_table(len(p)) for p, _ in inits]      # the blocking threshold depends on the particle number
    if any(not np.array_equal(t, tables[0]) for t in tables[1:]):
        raise ValueError("run_batched_exact_statistics: the systems' particle numbers give different blocking thresholds; "
                         "run them in separate batches")
    front_lo = np.array([acc0.front_range(s)[0] for s in range(first.L)], np.int32)
    r = run_raw(L=first.L, K=first.K, periodic=first.periodic, sigma_grid=first._sigma_grid, rate_diffusion=first.rate_diffusion,
                rate_active=first.rate_active, betas=[float(ps.beta) for ps in systems], states=inits, times_obs=times_obs, T=T,
                seed=seed, minus_anchor=first.minus_anchor, immobilize=first.immobilize_when_anchored,
                suppress_flip=first.suppress_flip_when_bound, crowding=first.crowding_suppresses_rates, k_on=first.k_on,
                k_off=first.k_off, k_exit=0.0, anchor_mask=first.is_anchor_site, want_states=False, x_wall=acc0.x_wall,
                ref_obs=acc0.start, front_lo=front_lo, block_table=tables[0], device=first.device, flip_table=first.flip_table())
    rows = []
    for s, ps in enumerate(systems):
        if int(r["n_recorded"][s]) < len(times_obs):
            raise RuntimeError("a system passed T before its last observation time (choose T beyond the last observation)")
        acc = observables.DeviceObservables(times_obs, first.L, first.dx, first.K)
        for k in range(len(times_obs)):
            sums = dict(zip(SCALARS, (int(v) for v in r["scalars"][s, k])))
            acc.add(k, sums, sums["n_front"] if k >= acc.start and sums["max_pos"] >= 0 else None)
        rows.append(acc.result())
        ps.n_events = int(r["n_events"][s])
    first.kernel_ms = r["kernel_ms"]
    return rows


def run_large_raw(*, L, K, periodic, sigma_grid, rate_diffusion, rate_active, beta, state, times_obs, T, seed=0,
                  minus_anchor=True, immobilize=True, suppress_flip=True, crowding=False, k_on=0.0, k_off=0.0, k_exit=0.0,
                  anchor_mask=None, uniforms=None, max_events=None, want_states=True, device=0, flip_table=None):
    """One large system (gil_run_large): state = (pos, sigma[, bound]).  Returns a dict like run_raw's, without a system axis."""
    lib = _lib()
    pos0 = np.ascontiguousarray(state[0], dtype=np.int32)
    sg0 = np.ascontiguousarray(state[1], dtype=np.int8)
    bd0 = None if len(state) < 3 or state[2] is None else np.ascontiguousarray(state[2], dtype=np.uint8)
    n0 = len(pos0)
    times = np.ascontiguousarray(times_obs, dtype=np.float64)
    M = len(times)
    if uniforms is not None:
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        max_events = uniforms.shape[0]
    elif max_events is None:
        max_events = 2 ** 40
    betas = np.array([float(beta)])
    mask = None if anchor_mask is None or not np.any(anchor_mask) else np.ascontiguousarray(anchor_mask, dtype=np.uint8)
    ftab = None if flip_table is None else np.ascontiguousarray(flip_table, dtype=np.float64)
    par = GilParams(flip_n=0 if ftab is None else ftab.shape[1] - 1, flip_table=None if ftab is None else _p(ftab).value,
                    L=L, K=K, periodic=int(bool(periodic)), minus_anchor=int(bool(minus_anchor)), immobilize=int(bool(immobilize)),
                    suppress_flip=int(bool(suppress_flip)), crowding=int(bool(crowding)), n_systems=1, n_cap=max(n0, 1), n_obs=M,
                    device=device, x_wall=0, ref_obs=-1, sigma_grid=float(sigma_grid), rate_diffusion=float(rate_diffusion),
                    rate_active=float(rate_active), k_on=float(k_on), k_off=float(k_off), k_exit=float(k_exit), T=float(T),
                    seed=int(seed) & (2 ** 64 - 1), max_events=int(max_events), beta=_p(betas).value,
                    anchor_mask=None if mask is None else _p(mask).value, times_obs=_p(times).value, front_lo=None, block_table=None)
    ncap = max(n0, 1)
    pos_obs = np.zeros((M, ncap), np.int32) if want_states else None
    sg_obs = np.zeros((M, ncap), np.int8) if want_states else None
    fl_obs = np.zeros((M, ncap), np.uint8) if want_states else None
    n_rec, n_ev, t_fin, n_exit = np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1), np.zeros(1, np.int32)
    exits = np.zeros((ncap, 3))
    ms = C.c_double()
    rc = lib.gil_run_large(C.byref(par), n0, _p(pos0), _p(sg0), _p(bd0), _p(uniforms), _p(pos_obs), _p(sg_obs), _p(fl_obs), _p(n_rec),
                           _p(n_ev), _p(t_fin), _p(exits), _p(n_exit), C.byref(ms))
    if rc != 0:
        raise capi.ApsError(rc, lib.gil_large_last_error().decode())
    return dict(pos=pos_obs, sigma=sg_obs, flags=fl_obs, n_recorded=int(n_rec[0]), n_events=int(n_ev[0]), t_final=float(t_fin[0]),
                exits=exits, n_exits=int(n_exit[0]), n0=n0, kernel_ms=ms.value)
