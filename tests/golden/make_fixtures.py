#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference in this container.

This script is the only place where the reference is imported.  It refuses to run when
/root/reference is absent (e.g. on the GPU box) and never copies reference source anywhere:
the outputs are data only (inputs + expected outputs), stored as compressed .npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py [g1 g2 g3 g5 g4 g6]

Fixture families:
  g1  m-field known answers          ParticleSystem.compute_local_m_field   PARTICLE_solver_CLASS.py:216-246
  g2  one-event known answers        ParticleSystem.step_gillespie          PARTICLE_solver_CLASS.py:254-448
      (driven through a recording proxy rng: exponential(scale) reveals R, choice(n, p) reveals rates/R)
  g3  seeded full trajectories       ParticleSystem.run                     PARTICLE_solver_CLASS.py:450-558
  g5  seeded initial conditions      _init_fixed / _init_poisson            PARTICLE_solver_CLASS.py:141-189
  g4  ensemble statistics            run + sweep-driver observables         ..._sweep_beta.py:123-229, :316-319, :500-525
  g6  hydrodynamic PDE               IMEXPDE.solve                          IMEX_PDE_solver_class.py:236-290
  g9  custom flip_rate_fn runs       ParticleSystem.run with a callable     PARTICLE_solver_CLASS.py:59-62, :261-262
  g10 structure observables          extract_structure_observables_from_out PARTICLE_solver_BIOLOGY_local_structure.py:55-103
"""
import ast
import json
import os
import sys
import types

REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("make_fixtures.py: /root/reference is absent; fixtures can only be regenerated "
             "in the build container")

sys.dont_write_bytecode = True           # never write __pycache__ into the read-only tree
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np   # noqa: E402
import scipy         # noqa: E402

# vispy is not installed here; the reference imports it at module top for an animation helper
# that the fixtures never call.  An empty stand-in module satisfies the import statement.
_v = types.ModuleType("vispy")
for _name in ("app", "scene", "io"):
    setattr(_v, _name, types.ModuleType("vispy." + _name))
    sys.modules["vispy." + _name] = getattr(_v, _name)
sys.modules["vispy"] = _v
sys.path.insert(0, REF)

import PARTICLE_solver_CLASS as ref_particle   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
VERSIONS = {"numpy": np.__version__, "scipy": scipy.__version__,
            "python": sys.version.split()[0]}


def _save(name, meta, arrays):
    meta = dict(meta)
    meta["versions"] = VERSIONS
    path = os.path.join(HERE, name)
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB, {len(arrays)} arrays")


def _random_state(rng, L, N, K):
    """Random capacity-respecting configuration (own helper, not reference code)."""
    slots = np.repeat(np.arange(L), K)
    pos = np.sort(rng.choice(slots, size=N, replace=False)).astype(np.int64)
    rng.shuffle(pos)
    sigma = rng.choice(np.array([1, -1], dtype=np.int8), size=N)
    return pos, sigma


def _counts(pos, sigma, L):
    cp = np.bincount(pos[sigma == 1], minlength=L)
    cm = np.bincount(pos[sigma == -1], minlength=L)
    return cp, cm


BASE_KW = dict(xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7, scale_rates=False)


# --------------------------------------------------------------------------------------- g1
def make_g1():
    rng = np.random.default_rng(1001)
    cases, arrays = [], {}
    spec = []
    for L in (64, 400, 1000):
        for K in (1, 3):
            spec += [
                (L, K, 0.0, False, "global"),
                (L, K, 0.005, True, "periodic_narrow"),
                (L, K, 0.05, True, "periodic_wide"),
                (L, K, 0.005, False, "reflect_narrow"),
                (L, K, 0.02, False, "reflect_mid"),
                (L, K, 0.3, False, "reflect_lw_ge_L"),
            ]
    # an almost-empty lattice (tot_conv == 0 far away from the particles) and a wall cluster
    spec += [(400, 1, 0.005, False, "reflect_sparse"), (400, 3, 0.01, False, "reflect_wall")]
    for idx, (L, K, sig, periodic, tag) in enumerate(spec):
        if tag == "reflect_sparse":
            N = 3
        elif tag == "reflect_wall":
            N = 30
        else:
            N = int(0.45 * L * K)
        pos, sigma = _random_state(rng, L, N, K)
        if tag == "reflect_wall":
            pos = np.concatenate([np.repeat(np.arange(5), 3), L - 1 - np.repeat(np.arange(5), 3)])
            sigma = rng.choice(np.array([1, -1], dtype=np.int8), size=pos.size)
        ps = ref_particle.ParticleSystem(L=L, N=N, site_capacity=K, local_kernel_sigma=sig,
                                         periodic=periodic, rng=np.random.default_rng(0), **BASE_KW)
        cp, cm = _counts(pos, sigma, L)
        m = ps.compute_local_m_field(cp, cm)
        cases.append(dict(L=L, K=K, sigma=sig, periodic=periodic, tag=tag, N=int(pos.size)))
        arrays[f"c{idx}_pos"] = pos.astype(np.int32)
        arrays[f"c{idx}_sigma"] = sigma.astype(np.int8)
        arrays[f"c{idx}_m"] = m
    _save("g1_mfield.npz", dict(cases=cases, base_kw=BASE_KW), arrays)


# --------------------------------------------------------------------------------------- g2
class RecordingRng:
    """Stands in for numpy's Generator inside ONE step_gillespie call (the class accepts any rng
    object, PARTICLE_solver_CLASS.py:75-78).  Records what the reference asks for and returns
    forced answers so every branch can be reached deterministically."""

    def __init__(self, forced_i, forced_uniforms, forced_tau=0.125):
        self.forced_i = forced_i
        self.uniforms = list(forced_uniforms)
        self.forced_tau = forced_tau
        self.scale = None
        self.p = None
        self.n_random = 0

    def exponential(self, scale):
        self.scale = float(scale)
        return self.forced_tau

    def choice(self, n, p=None):
        self.p = np.array(p, dtype=float)
        return self.forced_i

    def random(self):
        self.n_random += 1
        return self.uniforms.pop(0)


def make_g2():
    rng = np.random.default_rng(2002)
    scen = [
        dict(tag="k1_reflect", L=40, N=18, K=1, periodic=False),
        dict(tag="k1_periodic", L=40, N=18, K=1, periodic=True),
        dict(tag="k3_reflect", L=30, N=50, K=3, periodic=False),
        dict(tag="k3_crowding", L=30, N=50, K=3, periodic=False, crowding_suppresses_rates=True),
        dict(tag="anchors_bind", L=40, N=45, K=2, periodic=False, anchor_positions=[0.25, 0.7],
             anchor_radius=0.06, k_on=0.8, k_off=0.4, k_exit=0.6),
        dict(tag="anchors_free_minus", L=40, N=45, K=2, periodic=True, anchor_positions=[0.5],
             anchor_radius=0.1, k_on=0.8, k_off=0.4, k_exit=0.6, minus_anchor=False,
             immobilize_when_anchored=False, suppress_flip_when_bound=False),
        dict(tag="scaled_rates", L=25, N=10, K=1, periodic=False, scale_rates=True),
        dict(tag="walls_full", L=12, N=12, K=1, periodic=False),
    ]
    arrays, cases = {}, []
    for s_idx, sc in enumerate(scen):
        sc = dict(sc)
        tag, L, N, K = sc.pop("tag"), sc.pop("L"), sc.pop("N"), sc.pop("K")
        kw = dict(BASE_KW)
        kw.update(rate_diffusion=0.35, rate_active=2.5, beta=1.3)
        kw.update(sc)
        ps = ref_particle.ParticleSystem(L=L, N=N, site_capacity=K, local_kernel_sigma=0.02,
                                         rng=np.random.default_rng(0), **kw)
        pos0, sigma0 = _random_state(rng, L, N, K)
        bound0 = np.zeros(N, dtype=bool)
        if "anchor_positions" in sc:
            # bind a random half of the minus particles sitting on anchor sites
            elig = np.where((sigma0 == -1) & ps.is_anchor_site[pos0])[0]
            bound0[elig[::2]] = True
            # and a couple of bound particles off-anchor / plus (legal inputs for the rate code)
            bound0[rng.choice(N, size=3, replace=False)] = True
        m_field = np.clip(rng.normal(0.0, 0.6, size=L), -1, 1)
        events = []
        stk = {k: [] for k in ('p', 'n1', 'pos', 'sigma', 'bound', 'cp', 'cm')}
        n_ev = 36
        for e in range(n_ev):
            i = int(rng.integers(N))
            u_v = float(rng.random()) if e % 6 else float([0.0, 0.999999][(e // 6) % 2])
            u_lr = float(rng.random())
            pos, sigma, bound = pos0.copy(), sigma0.copy(), bound0.copy()
            cp, cm = _counts(pos, sigma, L)
            cp, cm = cp.copy(), cm.copy()
            init_bin = pos.copy()
            r = RecordingRng(i, [u_v, u_lr])
            ps.rng = r
            ret = ps.step_gillespie(pos, sigma, bound, m_field, cp, cm, init_bin, [], [], [], 1.5)
            assert len(ret) == 9, "early return hit in fixture scenario"
            pos1, sigma1, bound1, tau, cp1, cm1, ex_t, ex_p, ex_b = ret
            events.append(dict(i=i, u_v=u_v, u_lr=u_lr, n_random=r.n_random, scale=r.scale,
                               exit_t=list(map(float, ex_t)), exit_p=list(map(int, ex_p)),
                               exit_b=list(map(int, ex_b))))
            n1 = len(pos1)
            stk["p"].append(r.p)
            stk["n1"].append(n1)
            for key, arr, fill in (("pos", pos1, -1), ("sigma", sigma1, 0), ("bound", bound1, 0)):
                padded = np.full(N, fill, dtype=np.int32)
                padded[:n1] = np.asarray(arr).astype(np.int32)
                stk[key].append(padded)
            stk["cp"].append(np.asarray(cp1, dtype=np.int32))
            stk["cm"].append(np.asarray(cm1, dtype=np.int32))
        arrays[f"s{s_idx}_p"] = np.stack(stk["p"])
        arrays[f"s{s_idx}_n1"] = np.array(stk["n1"], dtype=np.int32)
        arrays[f"s{s_idx}_pos1"] = np.stack(stk["pos"])
        arrays[f"s{s_idx}_sigma1"] = np.stack(stk["sigma"]).astype(np.int8)
        arrays[f"s{s_idx}_bound1"] = np.stack(stk["bound"]).astype(np.int8)
        arrays[f"s{s_idx}_cp1"] = np.stack(stk["cp"])
        arrays[f"s{s_idx}_cm1"] = np.stack(stk["cm"])
        arrays[f"s{s_idx}_pos0"] = pos0.astype(np.int32)
        arrays[f"s{s_idx}_sigma0"] = sigma0
        arrays[f"s{s_idx}_bound0"] = bound0
        arrays[f"s{s_idx}_m_field"] = m_field
        arrays[f"s{s_idx}_is_anchor"] = ps.is_anchor_site.copy()
        ctor = dict(L=L, N=N, site_capacity=K, local_kernel_sigma=0.02, **kw)
        cases.append(dict(tag=tag, ctor=ctor, events=events,
                          rate_diffusion_eff=ps.rate_diffusion, rate_active_eff=ps.rate_active))
    _save("g2_events.npz", dict(cases=cases), arrays)


# --------------------------------------------------------------------------------------- g3
def _exp_profile(L, amp, ell):
    xs = np.arange(L) / float(L)
    return amp * np.exp(-xs / ell)


def _table_callable(arr):
    L = len(arr)
    return lambda x: float(arr[int(np.clip(np.round(x * L), 0, L - 1))])


def _pack_out(prefix, out, arrays):
    M = len(out["times_obs"])
    lens = np.array([len(p) for p in out["pos_list"]], dtype=np.int64)
    arrays[prefix + "times_obs"] = out["times_obs"]
    arrays[prefix + "pos_len"] = lens
    arrays[prefix + "pos_cat"] = np.concatenate(out["pos_list"]).astype(np.int64)
    arrays[prefix + "bound_cat"] = np.concatenate(out["bound_list"]).astype(bool)
    arrays[prefix + "particle_count"] = np.array(out["particle_count_list"], dtype=np.int64)
    for k in ("rho_p_list", "rho_m_list", "total_list", "m_local_list", "m_global"):
        arrays[prefix + k] = out[k]
    for k in ("rho_hat_complex", "fft_amp_list", "var_list"):
        if out[k] is not None:
            arrays[prefix + k] = out[k]
    arrays[prefix + "exit_times"] = np.array(out["exit_times"], dtype=float)
    arrays[prefix + "exit_positions"] = np.array(out["exit_positions"], dtype=np.int64)
    assert M == len(lens)


def make_g3():
    arrays, cases = {}, []
    L = 96
    rp = _exp_profile(L, 0.9, 0.3)
    rm = np.full(L, 0.25)
    specs = [
        dict(tag="fixed_k1_reflect", seed=11, ctor=dict(L=100, N=50, site_capacity=1, init="fixed",
             local_kernel_sigma=0.02, periodic=False, rate_diffusion=0.5, rate_active=3.0, beta=1.2,
             xlim=1.0, scale_rates=False), run=dict(T=1.5, obs_dt=0.25, record_fft=True, record_var=True)),
        dict(tag="poisson_k2_periodic", seed=12, ctor=dict(L=L, site_capacity=2, init="poisson",
             local_kernel_sigma=0.03, periodic=True, rate_diffusion=0.4, rate_active=2.0, beta=0.8,
             xlim=1.0, scale_rates=False), run=dict(T=1.0, obs_dt=0.2, record_fft=True, record_var=False),
             poisson=True),
        dict(tag="fixed_k3_global_anchors", seed=13, ctor=dict(L=80, N=90, site_capacity=3, init="fixed",
             local_kernel_sigma=0.0, periodic=False, rate_diffusion=0.3, rate_active=2.0, beta=1.5,
             xlim=1.0, scale_rates=False, anchor_positions=[0.3, 0.75], anchor_radius=0.05,
             k_on=3.0, k_off=1.0, k_exit=2.0), run=dict(T=1.2, obs_dt=0.3, record_fft=False, record_var=False)),
        dict(tag="fixed_k1_reflect_wide_scaled", seed=14, ctor=dict(L=60, N=30, site_capacity=1, init="fixed",
             local_kernel_sigma=0.3, periodic=False, rate_diffusion=0.0005, rate_active=0.05, beta=2.0,
             xlim=1.0, scale_rates=True), run=dict(T=0.8, obs_dt=0.2, record_fft=False, record_var=True)),
    ]
    for c_idx, sp in enumerate(specs):
        ctor = dict(sp["ctor"])
        kw = dict(ctor)
        if sp.get("poisson"):
            kw["rho0_plus"] = _table_callable(rp)
            kw["rho0_minus"] = _table_callable(rm)
            arrays[f"c{c_idx}_rho0_plus"] = rp
            arrays[f"c{c_idx}_rho0_minus"] = rm
        ps = ref_particle.ParticleSystem(rng=np.random.default_rng(sp["seed"]), **kw)
        out = ps.run(**sp["run"])
        _pack_out(f"c{c_idx}_", out, arrays)
        cases.append(dict(tag=sp["tag"], seed=sp["seed"], ctor=ctor, run=sp["run"],
                          poisson=bool(sp.get("poisson"))))
        print("  g3", sp["tag"], "events->final N", out["particle_count_list"][-1],
              "exits", len(out["exit_times"]))
    _save("g3_trajectories.npz", dict(cases=cases), arrays)


# --------------------------------------------------------------------------------------- g5
def make_g5():
    arrays, cases = {}, []
    L = 120
    rp = _exp_profile(L, 1.1, 0.25)
    rm = np.full(L, 0.4)
    specs = [
        dict(tag="fixed_k1", seed=51, ctor=dict(L=200, N=120, site_capacity=1, init="fixed")),
        dict(tag="fixed_k1_float_N", seed=52, ctor=dict(L=200, N=75, site_capacity=1, init="fixed")),
        dict(tag="fixed_k3", seed=53, ctor=dict(L=50, N=110, site_capacity=3, init="fixed")),
        dict(tag="poisson_k1", seed=54, ctor=dict(L=L, site_capacity=1, init="poisson"), poisson=True),
        dict(tag="poisson_k2", seed=55, ctor=dict(L=L, site_capacity=2, init="poisson"), poisson=True),
    ]
    for c_idx, sp in enumerate(specs):
        kw = dict(sp["ctor"], **BASE_KW)
        if sp.get("poisson"):
            kw["rho0_plus"] = _table_callable(rp)
            kw["rho0_minus"] = _table_callable(rm)
            arrays[f"c{c_idx}_rho0_plus"] = rp
            arrays[f"c{c_idx}_rho0_minus"] = rm
        ps = ref_particle.ParticleSystem(rng=np.random.default_rng(sp["seed"]), **kw)
        pos, sigma = ps.init_particles()
        arrays[f"c{c_idx}_pos"] = pos
        arrays[f"c{c_idx}_sigma"] = sigma
        cases.append(dict(tag=sp["tag"], seed=sp["seed"], ctor=sp["ctor"], poisson=bool(sp.get("poisson"))))
    _save("g5_init.npz", dict(cases=cases, base_kw=BASE_KW), arrays)


# --------------------------------------------------------------------------------------- g4
_OBS_FUNCS = ("compute_v_eff_and_window", "compute_rho_eff", "compute_blocking_probability",
              "compute_mean_magnetizatoin", "compute_D_eff_active")


def _load_driver_observables():
    """The sweep driver runs a whole sweep at import time (module-level code,
    ..._sweep_beta.py:1030-1034), so it cannot be imported.  Pull only the five observable
    functions out of its syntax tree and compile them here, in memory."""
    path = os.path.join(REF, "PARTICLE_solver_BIOLOGY_EXCLUSION_sweep_beta.py")
    with open(path) as fh:
        tree = ast.parse(fh.read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in _OBS_FUNCS]
    assert len(keep) == len(_OBS_FUNCS)
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


G4_CTOR = dict(L=1000, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, init="fixed", N=500,
               scale_rates=False, local_kernel_sigma=0.005, periodic=False, site_capacity=1,
               k_on=0.0, k_off=0.0, k_exit=0.0)
G4_RUN = dict(T=20.0, obs_dt=0.1, record_fft=False, record_var=False)
G4_BETAS = (0.0, 0.9, 1.5, 3.0)
G4_RUNS = 32


def _g4_one(args):
    beta, seed = args
    obs = _load_driver_observables()
    ps = ref_particle.ParticleSystem(beta=beta, rng=np.random.default_rng(seed), **G4_CTOR)
    out = ps.run(**G4_RUN)
    mean_v, v_ts, times, si, ei, frac_b = obs["compute_v_eff_and_window"](
        out, ps, boundary_xmin=0.99, max_buondary_fraction=0.06, min_window_fraction=0.10)
    D = obs["compute_D_eff_active"](out, ps, start_idx=si, end_idx=ei)
    m = obs["compute_mean_magnetizatoin"](out, si, ei)
    rho = obs["compute_rho_eff"](out, si, ei)
    blk = obs["compute_blocking_probability"](out, si, ei)
    prof = out["total_list"][-1].reshape(50, -1).mean(axis=1)
    m_ts = out["m_global"]
    # early-time observables too (less boundary pile-up): COM displacement over t in [0, 5]
    k5 = int(round(5.0 / G4_RUN["obs_dt"]))
    com = np.array([p.mean() for p in out["pos_list"]]) * ps.dx
    return dict(beta=beta, seed=seed, v=mean_v, D=float(D), m=m, rho=rho, blk=blk, si=si, ei=ei,
                prof=prof, m_ts=m_ts[::10].copy(), com=com[::10].copy(), com5=float(com[k5] - com[0]))


def make_g4():
    import multiprocessing as mp
    jobs = [(b, 4000 + 100 * bi + r) for bi, b in enumerate(G4_BETAS) for r in range(G4_RUNS)]
    with mp.Pool(8) as pool:
        res = pool.map(_g4_one, jobs, chunksize=2)
    arrays, cases = {}, []
    for bi, b in enumerate(G4_BETAS):
        rr = [r for r in res if r["beta"] == b]
        for k in ("v", "D", "m", "rho", "blk", "com5"):
            arrays[f"b{bi}_{k}"] = np.array([r[k] for r in rr], dtype=float)
        arrays[f"b{bi}_prof"] = np.stack([r["prof"] for r in rr])
        arrays[f"b{bi}_m_ts"] = np.stack([r["m_ts"] for r in rr])
        arrays[f"b{bi}_com"] = np.stack([r["com"] for r in rr])
        arrays[f"b{bi}_window"] = np.array([[r["si"], r["ei"]] for r in rr], dtype=np.int64)
        arrays[f"b{bi}_seeds"] = np.array([r["seed"] for r in rr], dtype=np.int64)
        cases.append(dict(beta=b, n_runs=len(rr)))
        print(f"  g4 beta={b}: v={np.mean([r['v'] for r in rr]):.4f} m={np.mean([r['m'] for r in rr]):.4f} "
              f"D={np.mean([r['D'] for r in rr]):.3e} blk={np.mean([r['blk'] for r in rr]):.4f}")
    _save("g4_ensemble_stats.npz", dict(cases=cases, ctor=G4_CTOR, run=G4_RUN, stride=10), arrays)


# --------------------------------------------------------------------------------------- g7
def make_g7():
    """Inputs (small reference runs) and outputs of the sweep driver's observable functions
    (..._sweep_beta.py:123-229, :316-319, :500-525) -- pins package file observables.py."""
    obs = _load_driver_observables()
    arrays, cases = {}, []
    specs = [
        dict(tag="k1_drift", seed=71, ctor=dict(L=120, xlim=1.0, rate_diffusion=0.05, rate_active=2.0, beta=0.6, init="fixed",
             N=50, scale_rates=False, local_kernel_sigma=0.02, site_capacity=1, k_on=0.0, k_off=0.0, k_exit=0.0),
             run=dict(T=6.0, obs_dt=0.1)),
        dict(tag="k2_ordered", seed=72, ctor=dict(L=100, xlim=1.0, rate_diffusion=0.2, rate_active=1.0, beta=2.5, init="fixed",
             N=90, scale_rates=False, local_kernel_sigma=0.05, site_capacity=2, k_on=0.0, k_off=0.0, k_exit=0.0),
             run=dict(T=5.0, obs_dt=0.125)),
        dict(tag="k1_wall_pileup", seed=73, ctor=dict(L=80, xlim=1.0, rate_diffusion=0.0, rate_active=6.0, beta=0.2, init="fixed",
             N=40, scale_rates=False, local_kernel_sigma=0.0, site_capacity=1, k_on=0.0, k_off=0.0, k_exit=0.0),
             run=dict(T=8.0, obs_dt=0.1)),
    ]
    for c_idx, sp in enumerate(specs):
        ps = ref_particle.ParticleSystem(rng=np.random.default_rng(sp["seed"]), **sp["ctor"])
        out = ps.run(**sp["run"])
        mean_v, v_ts, times, si, ei, frac_b = obs["compute_v_eff_and_window"](
            out, ps, boundary_xmin=0.99, max_buondary_fraction=0.06, min_window_fraction=0.10)
        res = dict(mean_v=mean_v, si=int(si), ei=int(ei),
                   D=float(obs["compute_D_eff_active"](out, ps, start_idx=si, end_idx=ei)),
                   m=float(obs["compute_mean_magnetizatoin"](out, si, ei)),
                   rho=float(obs["compute_rho_eff"](out, si, ei)),
                   blk=float(obs["compute_blocking_probability"](out, si, ei)))
        pre = f"c{c_idx}_"
        arrays[pre + "times_obs"] = out["times_obs"]
        arrays[pre + "total_list"] = out["total_list"]
        arrays[pre + "rho_p_list"] = out["rho_p_list"]
        arrays[pre + "m_global"] = out["m_global"]
        arrays[pre + "pos_cat"] = np.concatenate(out["pos_list"]).astype(np.int64)
        arrays[pre + "pos_len"] = np.array([len(p) for p in out["pos_list"]], dtype=np.int64)
        arrays[pre + "v_ts"] = v_ts
        arrays[pre + "frac_boundary"] = frac_b
        cases.append(dict(tag=sp["tag"], L=sp["ctor"]["L"], dx=ps.dx, **res))
        print("  g7", sp["tag"], res)
    _save("g7_observables.npz", dict(cases=cases), arrays)


# --------------------------------------------------------------------------------------- g6
G6_CASES = [
    dict(tag="periodic_bidirectional_local", bc="periodic", active_model="bidirectional", gaussian_kernel=False, init="poisson"),
    dict(tag="neumann_anchored_kernel", bc="neumann", active_model="anchored_minus", gaussian_kernel=True, init="poisson"),
    dict(tag="periodic_anchored_kernel", bc="periodic", active_model="anchored_minus", gaussian_kernel=True, init="homogeneous"),
    dict(tag="neumann_bidirectional_local", bc="neumann", active_model="bidirectional", gaussian_kernel=False, init="homogeneous"),
]
G6_KW = dict(L=200, xlim=1.0, T=0.15, dt=5e-4, gamma=2.33e-4, lam=0.6, beta=2.0, kernel_sigma=0.02, snapshot_interval=100)


def make_g6():
    """Hydrodynamic-limit PDE (IMEX_PDE_solver_class.py:11-307): seeded short runs of all four
    boundary-condition / active-model combinations, tracers included."""
    import tempfile
    import IMEX_PDE_solver_class as ref_pde
    arrays, cases = {}, []
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:       # the constructor creates an output directory in the cwd
        os.chdir(tmp)
        try:
            for c_idx, c in enumerate(G6_CASES):
                pde = ref_pde.IMEXPDE(bc=c["bc"], active_model=c["active_model"], gaussian_kernel=c["gaussian_kernel"],
                                      seed=600 + c_idx, outdir="out", **G6_KW)
                pde.initialize(mode=c["init"], rho0=1.0, noise=0.2, n_tracers=300)
                arrays[f"c{c_idx}_rho_p0"], arrays[f"c{c_idx}_rho_m0"] = pde.rho_p.copy(), pde.rho_m.copy()
                pde.solve()
                o = pde.get_output()
                for k in ("rho_p", "rho_m", "m_series", "var_series", "v_eff_series", "D_eff_series", "snapshots", "times"):
                    arrays[f"c{c_idx}_{k}"] = np.asarray(o[k])
                arrays[f"c{c_idx}_fft_amp_last"] = o["fft_amp"][-1]
                arrays[f"c{c_idx}_tracers"] = pde.tracers_unwrapped.copy()
                arrays[f"c{c_idx}_tracer_state"] = pde.tracer_state.copy()
                cases.append(dict(c, seed=600 + c_idx))
                print("  g6", c["tag"], "m_end", float(o["m_series"][-1]), "mass", float((o["rho_p"] + o["rho_m"]).sum()))
        finally:
            os.chdir(cwd)
    _save("g6_pde.npz", dict(cases=cases, kw=G6_KW, init=dict(rho0=1.0, noise=0.2, n_tracers=300)), arrays)


# --------------------------------------------------------------------------------------- g9
FLIP_FNS = {   # custom flip_rate_fn callables (ref :21, :59-62, applied at :261-262), by name so that tests can rebuild them
    "glauber": lambda p: (lambda sigma, m: 0.5 * p["nu"] * (1.0 - sigma * np.tanh(p["b"] * m))),
    "threshold": lambda p: (lambda sigma, m: np.where(sigma * m > p["m0"], p["lo"], p["hi"]).astype(float)),
}


def make_g9():
    """Seeded trajectories of the reference with a CUSTOM flip_rate_fn (ParticleSystem.run, ref :450-558 with :261-262)."""
    arrays, cases = {}, []
    specs = [
        dict(tag="glauber_k1_reflect", seed=91, fn="glauber", fn_par=dict(nu=2.0, b=1.4),
             ctor=dict(L=90, N=45, site_capacity=1, init="fixed", local_kernel_sigma=0.03, periodic=False, rate_diffusion=0.4,
                       rate_active=2.5, beta=0.0, xlim=1.0, scale_rates=False), run=dict(T=1.5, obs_dt=0.25, record_fft=False, record_var=False)),
        dict(tag="threshold_k2_periodic_anchors", seed=92, fn="threshold", fn_par=dict(m0=0.1, lo=0.2, hi=1.7),
             ctor=dict(L=80, N=70, site_capacity=2, init="fixed", local_kernel_sigma=0.04, periodic=True, rate_diffusion=0.3,
                       rate_active=2.0, beta=1.0, xlim=1.0, scale_rates=False, anchor_positions=[0.4], anchor_radius=0.06,
                       k_on=2.0, k_off=1.0, k_exit=1.5), run=dict(T=1.2, obs_dt=0.2, record_fft=False, record_var=False)),
    ]
    for c_idx, sp in enumerate(specs):
        fn = FLIP_FNS[sp["fn"]](sp["fn_par"])
        ps = ref_particle.ParticleSystem(rng=np.random.default_rng(sp["seed"]), flip_rate_fn=fn, **sp["ctor"])
        out = ps.run(**sp["run"])
        _pack_out(f"c{c_idx}_", out, arrays)
        cases.append(dict(tag=sp["tag"], seed=sp["seed"], ctor=sp["ctor"], run=sp["run"], fn=sp["fn"], fn_par=sp["fn_par"]))
        print("  g9", sp["tag"], "final N", out["particle_count_list"][-1], "exits", len(out["exit_times"]))
    _save("g9_flip_rate_fn.npz", dict(cases=cases), arrays)


# --------------------------------------------------------------------------------------- g10
def _load_structure_observables():
    """extract_structure_observables_from_out of PARTICLE_solver_BIOLOGY_local_structure.py:55-103, pulled out of the
    file's syntax tree (the module imports plotting code at its top) and compiled in memory."""
    path = os.path.join(REF, "PARTICLE_solver_BIOLOGY_local_structure.py")
    with open(path) as fh:
        tree = ast.parse(fh.read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "extract_structure_observables_from_out"]
    assert len(keep) == 1
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["extract_structure_observables_from_out"]


def make_g10():
    """Inputs (var_list, fft_amp_list, m_local_list of small reference runs with record_fft/record_var) and outputs of
    the reference's structure observables -- pins observables.extract_structure_observables_from_out."""
    fn = _load_structure_observables()
    arrays, cases = {}, []
    specs = [
        dict(tag="k1_dense", seed=101, k_max=None, start_fraction=0.5,
             ctor=dict(L=100, xlim=1.0, rate_diffusion=0.1, rate_active=2.0, beta=1.8, init="fixed", N=90, scale_rates=False,
                       local_kernel_sigma=0.03, site_capacity=1, k_on=0.0, k_off=0.0, k_exit=0.0), run=dict(T=4.0, obs_dt=0.1)),
        dict(tag="k2_kmax12", seed=102, k_max=12, start_fraction=0.4,
             ctor=dict(L=128, xlim=1.0, rate_diffusion=0.3, rate_active=1.0, beta=2.5, init="fixed", N=140, scale_rates=False,
                       local_kernel_sigma=0.05, site_capacity=2, k_on=0.0, k_off=0.0, k_exit=0.0), run=dict(T=3.0, obs_dt=0.125)),
    ]
    for c_idx, sp in enumerate(specs):
        ps = ref_particle.ParticleSystem(rng=np.random.default_rng(sp["seed"]), **sp["ctor"])
        out = ps.run(record_fft=True, record_var=True, **sp["run"])
        res = fn(out, start_fraction=sp["start_fraction"], k_max=sp["k_max"])
        pre = f"c{c_idx}_"
        for k in ("times_obs", "var_list", "fft_amp_list", "m_local_list", "total_list"):
            arrays[pre + k] = np.asarray(out[k])
        arrays[pre + "fft_mean"] = res["fft_mean"]
        arrays[pre + "fft_std"] = res["fft_std"]
        cases.append(dict(tag=sp["tag"], k_max=sp["k_max"], start_fraction=sp["start_fraction"], L=sp["ctor"]["L"],
                          **{k: float(res[k]) for k in ("var_mean", "var_std", "low_k_power", "m_local_var", "lowk_variance")},
                          dominant_k=int(res["dominant_k"])))
        print("  g10", sp["tag"], {k: cases[-1][k] for k in ("var_mean", "dominant_k", "low_k_power", "m_local_var")})
    _save("g10_structure.npz", dict(cases=cases), arrays)


# --------------------------------------------------------------------------------------- g8
def make_g8():
    """The only simulation results the reference itself records: the number lists of plot_figs.py:6-9 (v_eff, D_eff and
    their standard errors over beta of the PDE tracer sweep).  Read as DATA from the file's text; nothing is executed."""
    import re
    with open(os.path.join(REF, "plot_figs.py")) as fh:
        text = fh.read()
    out = {"provenance": "Numbers the reference holds as pasted results in plot_figs.py:6-9 (series plotted as 'Particle Sim'; by their values -- "
                         "D -> gamma = 0.2, v -> lam = 0.6 -- they are the tracer sweep of IMEX_PDE_solver_run_sweep.py:7-75: L=1000, T=40, dt=5e-4, "
                         "gamma=0.2, lam=0.6, periodic, bidirectional, gaussian_kernel=True, kernel_sigma=1e5-10, homogeneous init rho0=1 noise=0.3, "
                         "1000 tracers, 3 runs per beta, window t in [20, 40], v = |nanmean v_eff|, D = nanmean D_eff, err = std(ddof=1)/sqrt(3)). "
                         "Data only (tests/golden/make_fixtures.py g8). Note: the D series corresponds to a longer averaging window than the 0.05 "
                         "of the shipped class (whose D_eff at beta=0 is 0.207, see tests/test_gpu_pde.py); only the v series is used as a parity target.",
           "beta_values": [0.3 * i for i in range(11)]}
    for key in ("v_mean", "v_err", "D_mean", "D_err"):
        m = re.search(r"^%s\s*=\s*\[(.*?)\]" % key, text, re.M | re.S)
        out[key] = [float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
        assert len(out[key]) == 11
    with open(os.path.join(HERE, "g8_pde_sweep_published.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote g8_pde_sweep_published.json")


if __name__ == "__main__":
    todo = sys.argv[1:] or ["g1", "g2", "g3", "g5"]
    for t in todo:
        print("==", t)
        globals()["make_" + t]()
    left = [f for f in os.listdir(REF) if f == "__pycache__"]
    assert not left, "bytecode was written into the reference tree"
