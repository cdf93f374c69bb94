"""CPU suite: the exact-Gillespie NumPy oracle against the fixtures generated from the reference.

Tolerances: initial conditions and seeded trajectories bit-exact; m-field <= 1e-15 abs;
event probabilities rates/R <= 1e-14 rel."""
import numpy as np
import pytest

from oracle.gillespie_numpy import GillespieOracle, gauss_reflect
from conftest import table_callable


def _counts(pos, sigma, L):
    return (np.bincount(pos[sigma == 1], minlength=L), np.bincount(pos[sigma == -1], minlength=L))


def test_g1_mean_field_matches_reference(golden):
    g = golden("g1_mfield.npz")
    base = g.meta["base_kw"]
    worst = 0.0
    for idx, c in enumerate(g.meta["cases"]):
        orc = GillespieOracle(L=c["L"], N=c["N"], site_capacity=c["K"], local_kernel_sigma=c["sigma"],
                              periodic=c["periodic"], rng=np.random.default_rng(0), **base)
        pos, sigma = g[f"c{idx}_pos"].astype(np.int64), g[f"c{idx}_sigma"]
        m = orc.mean_field(*_counts(pos, sigma, c["L"]))
        err = np.max(np.abs(m - g[f"c{idx}_m"]))
        worst = max(worst, err)
        assert err <= 1e-15, (c, err)
    print("worst |dm| =", worst)


def test_gauss_reflect_is_scipy_bit_for_bit():
    scipy_nd = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(3)
    for L, sg in ((50, 0.7), (64, 3.3), (400, 8.0), (100, 40.0)):     # last: radius 160 > L
        x = rng.integers(0, 4, size=L).astype(float)
        ours = gauss_reflect(x, sg)
        theirs = scipy_nd.gaussian_filter1d(x, sigma=sg, mode="reflect")
        assert np.array_equal(ours, theirs), (L, sg, np.max(np.abs(ours - theirs)))


def test_g5_initial_conditions_bit_exact(golden):
    g = golden("g5_init.npz")
    for idx, c in enumerate(g.meta["cases"]):
        kw = dict(c["ctor"], **g.meta["base_kw"])
        if c["poisson"]:
            kw["rho0_plus"] = table_callable(g[f"c{idx}_rho0_plus"])
            kw["rho0_minus"] = table_callable(g[f"c{idx}_rho0_minus"])
        orc = GillespieOracle(rng=np.random.default_rng(c["seed"]), **kw)
        pos, sigma = orc.init_particles()
        assert pos.dtype == np.int64 and sigma.dtype == np.int8
        assert np.array_equal(pos, g[f"c{idx}_pos"]), c["tag"]
        assert np.array_equal(sigma, g[f"c{idx}_sigma"]), c["tag"]


class ForcedRng:
    def __init__(self, i, uniforms):
        self.i, self.u, self.scale, self.p = i, list(uniforms), None, None

    def exponential(self, scale):
        self.scale = float(scale)
        return 0.125

    def choice(self, n, p=None):
        self.p = np.array(p)
        return self.i

    def random(self):
        return self.u.pop(0)


def test_g2_single_events_match_reference(golden):
    g = golden("g2_events.npz")
    seen = set()
    for s_idx, sc in enumerate(g.meta["cases"]):
        orc = GillespieOracle(rng=np.random.default_rng(0), **sc["ctor"])
        assert orc.par.rate_diffusion == sc["rate_diffusion_eff"]
        assert orc.par.rate_active == sc["rate_active_eff"]
        assert np.array_equal(orc.par.is_anchor_site, g[f"s{s_idx}_is_anchor"])
        L = sc["ctor"]["L"]
        pos0 = g[f"s{s_idx}_pos0"].astype(np.int64)
        sigma0, bound0 = g[f"s{s_idx}_sigma0"], g[f"s{s_idx}_bound0"]
        m_field = g[f"s{s_idx}_m_field"]
        for e, ev in enumerate(sc["events"]):
            pos, sigma, bound = pos0.copy(), sigma0.copy(), bound0.copy()
            cp, cm = _counts(pos, sigma, L)
            cp, cm = cp.copy(), cm.copy()
            orc.rng = ForcedRng(ev["i"], [ev["u_v"], ev["u_lr"]])
            exits = ([], [])
            pos, sigma, bound, tau = orc.fire_event(pos, sigma, bound, m_field, cp, cm, 1.5, exits)
            assert orc.rng.scale == ev["scale"]                       # 1/R, bit-exact
            np.testing.assert_allclose(orc.rng.p, g[f"s{s_idx}_p"][e], rtol=1e-14, atol=0)
            n1 = int(g[f"s{s_idx}_n1"][e])
            assert len(pos) == n1
            assert np.array_equal(pos, g[f"s{s_idx}_pos1"][e][:n1])
            assert np.array_equal(sigma, g[f"s{s_idx}_sigma1"][e][:n1])
            assert np.array_equal(bound, g[f"s{s_idx}_bound1"][e][:n1].astype(bool))
            assert np.array_equal(cp, g[f"s{s_idx}_cp1"][e])
            assert np.array_equal(cm, g[f"s{s_idx}_cm1"][e])
            assert exits[0] == ev["exit_t"] and [int(x) for x in exits[1]] == ev["exit_p"]
            assert 2 - len(orc.rng.u) == ev["n_random"]
            # classify what happened, to prove every branch was exercised
            if n1 < len(pos0):
                seen.add("exit")
            elif not np.array_equal(pos, pos0):
                seen.add("hop_diff" if ev["n_random"] == 2 else "hop_active")
            elif not np.array_equal(sigma, sigma0):
                seen.add("flip")
            elif not np.array_equal(bound, bound0):
                seen.add("bind" if bound.sum() > bound0.sum() else "unbind")
    assert seen == {"exit", "hop_diff", "hop_active", "flip", "bind", "unbind"}, seen


def test_g3_seeded_trajectories_bit_exact(golden):
    g = golden("g3_trajectories.npz")
    for idx, c in enumerate(g.meta["cases"]):
        kw = dict(c["ctor"])
        if c["poisson"]:
            kw["rho0_plus"] = table_callable(g[f"c{idx}_rho0_plus"])
            kw["rho0_minus"] = table_callable(g[f"c{idx}_rho0_minus"])
        orc = GillespieOracle(rng=np.random.default_rng(c["seed"]), **kw)
        out = orc.run(**c["run"])
        pre = f"c{idx}_"
        assert np.array_equal(out["times_obs"], g[pre + "times_obs"])
        assert np.array_equal(np.concatenate(out["pos_list"]), g[pre + "pos_cat"]), c["tag"]
        assert [len(p) for p in out["pos_list"]] == g[pre + "pos_len"].tolist()
        assert np.array_equal(np.concatenate(out["bound_list"]), g[pre + "bound_cat"])
        assert out["particle_count_list"] == g[pre + "particle_count"].tolist()
        for k in ("rho_p_list", "rho_m_list", "total_list", "m_global"):
            assert np.array_equal(out[k], g[pre + k]), (c["tag"], k)
        assert np.max(np.abs(out["m_local_list"] - g[pre + "m_local_list"])) <= 1e-15
        for k in ("rho_hat_complex", "fft_amp_list", "var_list"):
            if (pre + k) in g:
                assert np.array_equal(out[k], g[pre + k]), (c["tag"], k)
            else:
                assert out[k] is None
        assert np.array_equal(np.array(out["exit_times"], dtype=float), g[pre + "exit_times"])
        assert np.array_equal(np.array(out["exit_positions"], dtype=np.int64), g[pre + "exit_positions"])


FLIP_FNS = {   # the named callables of fixture G9 (tests/golden/make_fixtures.py)
    "glauber": lambda p: (lambda sigma, m: 0.5 * p["nu"] * (1.0 - sigma * np.tanh(p["b"] * m))),
    "threshold": lambda p: (lambda sigma, m: np.where(sigma * m > p["m0"], p["lo"], p["hi"]).astype(float)),
}


def test_g9_custom_flip_rate_fn_trajectories_bit_exact(golden):
    """A caller-supplied flip_rate_fn (ref :59-62, :261-262): the oracle's event loop reproduces the reference's seeded runs."""
    g = golden("g9_flip_rate_fn.npz")
    for idx, c in enumerate(g.meta["cases"]):
        orc = GillespieOracle(rng=np.random.default_rng(c["seed"]), flip_rate_fn=FLIP_FNS[c["fn"]](c["fn_par"]), **c["ctor"])
        out = orc.run(**c["run"])
        pre = f"c{idx}_"
        assert np.array_equal(np.concatenate(out["pos_list"]), g[pre + "pos_cat"]), c["tag"]
        assert np.array_equal(np.concatenate(out["bound_list"]), g[pre + "bound_cat"])
        for k in ("rho_p_list", "rho_m_list", "total_list", "m_global"):
            assert np.array_equal(out[k], g[pre + k]), (c["tag"], k)
        assert np.array_equal(np.array(out["exit_times"], dtype=float), g[pre + "exit_times"])
