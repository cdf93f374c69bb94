"""CPU suite: the C restatement of the synchronous stepper (oracle/sync_oracle.c) pinned against
published known answers (Philox), libm (exp), and the reference fixtures G1 (m-field), G2 (rates)."""
import numpy as np
import pytest

from oracle.gillespie_numpy import LatticeGasParams
from oracle import sync_oracle as so


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors: philox4x32 10
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = so.philox4x32_10(ctr, key)
        assert tuple(int(x) for x in got) == want


def test_deterministic_exp_within_2ulp_of_libm():
    xs = np.concatenate([np.linspace(-40, 40, 20001), np.linspace(-1e-3, 1e-3, 2001), [0.0, -700.0, 700.0]])
    got = so.det_exp(xs)
    want = np.exp(xs)
    ulp = np.abs(got - want) / np.spacing(want)
    assert ulp.max() <= 2.0, ulp.max()
    assert so.det_exp([0.0])[0] == 1.0


def _raw_table(sigma_g, L, periodic):
    """Unquantised image-folded Gaussian taps, straight from the definition (numpy exp)."""
    if periodic:
        t = np.arange(L // 2 + 1, dtype=float)
        return np.exp(-0.5 * t * t / sigma_g ** 2)
    lw = int(4.0 * sigma_g + 0.5)
    tmax = min(lw, L)
    out = np.zeros(tmax + 1)
    for t in range(tmax + 1):
        k = 0
        while True:
            d1, d2 = t + 2 * L * k, 2 * L * k - t
            hit = False
            if d1 <= lw:
                out[t] += np.exp(-0.5 * d1 * d1 / sigma_g ** 2); hit = True
            if k > 0 and d2 <= lw:
                out[t] += np.exp(-0.5 * d2 * d2 / sigma_g ** 2); hit = True
            if not hit:
                break
            k += 1
    return out


def _params(c, base):
    return LatticeGasParams.from_kwargs(L=c["L"], xlim=base["xlim"], rate_diffusion=base["rate_diffusion"],
                                        rate_active=base["rate_active"], beta=base["beta"],
                                        scale_rates=base["scale_rates"], local_kernel_sigma=c["sigma"],
                                        periodic=c["periodic"], site_capacity=c["K"])


def test_g1_field_image_formula_matches_reference(golden):
    """All-images formula with UNQUANTISED weights reproduces the reference m-field to rounding;
    with the 2^-q grid the difference is the documented weight rounding only."""
    g = golden("g1_mfield.npz")
    worst_raw = worst_q = 0.0
    for idx, c in enumerate(g.meta["cases"]):
        par = _params(c, g.meta["base_kw"])
        pos, sigma, want = g[f"c{idx}_pos"], g[f"c{idx}_sigma"], g[f"c{idx}_m"]
        raw = _raw_table(par.sigma_grid, c["L"], c["periodic"]) if c["sigma"] > 0 else None
        for quantised in (False, True):
            orc = so.SyncOracle(par, dt=0.01, seed=1, raw_table=None if quantised else raw)
            orc.set_state(pos, sigma)
            cp, cm, m = orc.field_sites()
            assert np.array_equal(cp, np.bincount(pos[sigma == 1], minlength=c["L"]))
            assert np.array_equal(cm, np.bincount(pos[sigma == -1], minlength=c["L"]))
            # The reference's periodic branch goes through an FFT (ref :223-227): where the true
            # Gaussian mass W is tiny its own round-off (~1e-16 absolute) dominates the ratio, so the
            # comparison is weighted by min(W, 1) (W >= 1 at every occupied site).
            wgt = np.minimum(orc.last_site_sums[1], 1.0) if c["periodic"] and c["sigma"] > 0 else 1.0
            err = np.max(np.abs(m - want) * wgt)
            if quantised:
                worst_q = max(worst_q, err)
                assert err <= 2e-11, (c, err, orc.q)
            else:
                worst_raw = max(worst_raw, err)
                assert err <= 5e-15, (c, err)
    print("worst raw", worst_raw, "worst quantised", worst_q)


def test_table_sums_are_exact_on_the_grid():
    for sg, L, K, per in ((5.0, 1000, 1, False), (1000.0, 200000, 1, False), (300.0, 1000, 3, False),
                          (20.0, 400, 2, True), (0.32, 64, 1, False)):
        tab, q = so.build_table(sg, L, K, per)
        scaled = tab * 2.0 ** q
        assert np.array_equal(scaled, np.round(scaled)), "weights are not on the 2^-q grid"
        # worst-case sum of every term one target can see stays below 2^52 grid units
        if per:
            worst = K * (2 * scaled[1:].sum() + scaled[0])
        elif len(tab) - 1 < L:
            worst = K * (2 * scaled[1:].sum() + scaled[0])
        else:
            worst = 2 * K * L * scaled.max()
        assert worst < 2.0 ** 52, (sg, L, K, per, q)
        assert tab[0] >= 1.0 and tab[-1] > 0.0


def test_g2_rates_match_reference(golden):
    g = golden("g2_events.npz")
    for s_idx, sc in enumerate(g.meta["cases"]):
        ct = dict(sc["ctor"])
        par = LatticeGasParams.from_kwargs(
            **{k: ct[k] for k in ct if k not in ("N", "site_capacity")}, site_capacity=ct["site_capacity"])
        orc = so.SyncOracle(par, dt=0.01, seed=1)
        orc.set_state(g[f"s{s_idx}_pos0"], g[f"s{s_idx}_sigma0"], g[f"s{s_idx}_bound0"])
        r = orc.rates_from_field(g[f"s{s_idx}_m_field"], use_libm=True)
        R = float(r["total"].sum())
        # every event of the scenario starts from the same state, so every recorded p is rates/R
        for e, ev in enumerate(sc["events"]):
            assert abs(1.0 / R - ev["scale"]) <= 4e-16 * ev["scale"]
            np.testing.assert_allclose(r["total"] / R, g[f"s{s_idx}_p"][e], rtol=1e-14, atol=1e-18)
        # deterministic exp changes the flip channel by at most 2 ulp
        r2 = orc.rates_from_field(g[f"s{s_idx}_m_field"], use_libm=False)
        np.testing.assert_allclose(r2["total"], r["total"], rtol=5e-16)
        for k in ("diff", "act", "bind", "unbind", "exit", "left", "right"):
            assert np.array_equal(r2[k], r[k])


def _mk(L=200, N=120, K=2, sigma=0.02, periodic=False, **kw):
    base = dict(xlim=1.0, rate_diffusion=0.6, rate_active=4.0, beta=1.1, scale_rates=False)
    base.update(kw)
    par = LatticeGasParams.from_kwargs(L=L, local_kernel_sigma=sigma, periodic=periodic, site_capacity=K, **base)
    rng = np.random.default_rng(5)
    slots = np.repeat(np.arange(L), K)
    pos = rng.permutation(rng.choice(slots, size=N, replace=False)).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    return par, pos, spin


@pytest.mark.parametrize("periodic", [False, True])
def test_sync_step_invariants(periodic):
    par, pos, spin = _mk(periodic=periodic)
    orc = so.SyncOracle(par, dt=0.05, seed=77)
    orc.set_state(pos, spin)
    moved = 0
    for _ in range(200):
        before = orc.pos.copy()
        d = orc.step(want_detail=True)
        occ = np.bincount(orc.pos, minlength=par.L)
        assert occ.max() <= par.K                       # exclusion
        assert orc.alive.all()                          # k_exit = 0: nobody leaves
        hop = np.isin(d["prop"], (1, 2, 3))
        step = orc.pos.astype(int) - before
        if periodic:
            step = (step + par.L // 2) % par.L - par.L // 2
        assert np.all(np.abs(step) <= 1)
        assert np.all(step[~(hop & (d["accepted"] == 1))] == 0)
        moved += int((step != 0).sum())
    assert moved > 100
    # determinism: same seed -> same trajectory; different seed -> different
    a = so.SyncOracle(par, dt=0.05, seed=77); a.set_state(pos, spin); a.run(200)
    b = so.SyncOracle(par, dt=0.05, seed=78); b.set_state(pos, spin); b.run(200)
    assert np.array_equal(a.pos, orc.pos) and np.array_equal(a.spin, orc.spin)
    assert not np.array_equal(b.pos, orc.pos)


def test_sync_commit_is_index_ordered_and_capacity_bound():
    """Two particles wanting the same free site: the lower index gets it (K=1)."""
    par = LatticeGasParams.from_kwargs(L=3, xlim=1.0, rate_diffusion=50.0, rate_active=0.0, beta=0.0,
                                       scale_rates=False, local_kernel_sigma=0.0, site_capacity=1)
    wins = {0: 0, 1: 0}
    for seed in range(300):
        orc = so.SyncOracle(par, dt=1.0, seed=seed)
        orc.set_state(np.array([0, 2], np.int32), np.array([-1, -1], np.int8))
        d = orc.step(want_detail=True)
        assert np.bincount(orc.pos, minlength=3).max() <= 1
        if d["prop"][0] == 2 and d["prop"][1] == 1:       # both propose site 1
            assert d["accepted"][0] == 1 and d["accepted"][1] == 0
            assert orc.pos.tolist() == [1, 2]
            wins[0] += 1
    assert wins[0] > 20


def test_sync_exit_log_and_dead_particles_are_inert():
    par = LatticeGasParams.from_kwargs(L=50, xlim=1.0, rate_diffusion=0.2, rate_active=1.0, beta=0.5,
                                       scale_rates=False, local_kernel_sigma=0.05, site_capacity=2,
                                       anchor_positions=[0.5], anchor_radius=0.2, k_on=5.0, k_off=0.5, k_exit=4.0)
    rng = np.random.default_rng(0)
    pos = rng.choice(np.repeat(np.arange(50), 2), size=40, replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=40)
    orc = so.SyncOracle(par, dt=0.05, seed=3)
    orc.set_state(pos, spin)
    orc.run(400)
    ex = orc.exits()
    assert len(ex) == int((orc.alive == 0).sum()) > 0
    assert np.all(np.diff(ex[:, 0]) >= 0)
    dead = np.where(orc.alive == 0)[0]
    assert sorted(ex[:, 2].astype(int).tolist()) == sorted(dead.tolist())
    assert np.array_equal(orc.pos[dead], ex[np.argsort(ex[:, 2]), 1].astype(np.int32))
    # the field ignores dead particles
    S, W, occ4 = orc.pair_sums()
    live = orc.alive == 1
    ref = so.SyncOracle(par, dt=0.05, seed=3)
    ref.set_state(orc.pos[live], orc.spin[live], orc.bound[live])
    S2, W2, occ42 = ref.pair_sums()
    assert np.array_equal(S[live], S2) and np.array_equal(W[live], W2) and np.array_equal(occ4[live], occ42)
