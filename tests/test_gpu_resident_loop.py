"""GPU suite (-m gpu): the resident loop (csrc/tile_loop.hpp -- many steps of the tiles formulation inside ONE launch,
the tiles exchanging deposit lists and boundary cells through tagged granules) against the CPU oracle and against the
one-launch-per-step path, bit for bit.

Replaces the loop of ParticleSystem.run (PARTICLE_solver_CLASS.py:511-516) like aps_step itself; the bar is the one of
tests/test_gpu_parity.py: integer state and the lattice arrays {W, S, occupancy} identical to the oracle's."""
import importlib
import os

import numpy as np
import pytest

from oracle import sync_oracle as so
from test_gpu_parity import check_lattice, make_handle, params, random_state

pytestmark = pytest.mark.gpu
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"


@pytest.fixture(scope="module")
def capi():
    mod = importlib.import_module(PKG + ".capi")
    assert mod.device_count() >= 1, "no GPU visible"
    return mod


def same_state(h, orc, tag=""):
    p, sg, bd, al = h.get_state()
    assert np.array_equal(al, orc.alive), tag
    assert np.array_equal(p, orc.pos), tag
    assert np.array_equal(sg, orc.spin), tag
    assert np.array_equal(bd, orc.bound), tag


LOOP_CASES = [
    dict(tag="k1_reflect", L=3000, K=1, sigma=0.01, frac=0.5),
    dict(tag="k1_periodic", L=3000, K=1, sigma=0.012, periodic=True, frac=0.5),
    dict(tag="k1_reach_one_tile", L=2000, K=1, sigma=0.002, frac=0.6),
    dict(tag="k2_crowding", L=1250, K=2, sigma=0.01, frac=0.7, crowding_suppresses_rates=True),
    dict(tag="k3_wide_images", L=200, K=3, sigma=0.3, frac=0.6),
    dict(tag="k2_periodic_wide", L=400, K=2, sigma=0.4, periodic=True, frac=0.5),
    dict(tag="anchors_bind_unbind", L=900, K=2, sigma=0.02, frac=0.5, anchor_positions=[0.3, 0.7], anchor_radius=0.08,
         k_on=3.0, k_off=1.0, k_exit=0.0),
    # particles leave the system (ref :307-312, :427-446): exit log and dead marks written from inside the loop
    dict(tag="anchors_exits", L=900, K=2, sigma=0.02, frac=0.5, anchor_positions=[0.3, 0.7], anchor_radius=0.08,
         k_on=3.0, k_off=1.0, k_exit=1.5),
    dict(tag="k3_anchors_exits_many_tiles", L=2500, K=3, sigma=0.01, frac=0.5, anchor_positions=[0.2, 0.5, 0.8], anchor_radius=0.05,
         k_on=4.0, k_off=0.5, k_exit=2.0),
    dict(tag="dense_diffusive", L=1280, K=1, sigma=0.05, frac=0.9, rate_diffusion=6.0),
    dict(tag="ragged_last_tile", L=60 * 7 + 5, K=1, sigma=0.03, frac=0.5),
    dict(tag="last_tile_of_two_sites_avoided", L=60 * 6 + 2, K=1, sigma=0.03, frac=0.5),   # (the geometry shrinks the tiles by a site)
    dict(tag="single_tile_torus", L=50, K=2, sigma=0.1, periodic=True, frac=0.5),
    # more deposits of one class in reach than a pooled list holds (1040): the entries that find it full are swept one by one
    dict(tag="overflowing_lists_torus", L=1200, K=3, sigma=0.4, periodic=True, frac=0.95, rate_diffusion=6.0),
    dict(tag="overflowing_lists_small_box_images", L=1200, K=3, sigma=0.3, frac=0.95, rate_diffusion=6.0),
    dict(tag="two_workgroups_per_cu_geometry", L=70000, K=1, sigma=0.002, frac=0.4),   # 512 tiles of 137 sites (ts_choose_geometry)
]


@pytest.mark.parametrize("case", LOOP_CASES, ids=lambda c: c["tag"])
@pytest.mark.parametrize("fp32", [False, True], ids=["f64", "i32"])
def test_resident_loop_equals_oracle(capi, case, fp32):
    case = dict(case)
    tag, frac = case.pop("tag"), case.pop("frac")
    par = params(**case)
    rng = np.random.default_rng(11)
    N = max(1, int(frac * par.L * par.K))
    pos, spin = random_state(rng, par.L, N, par.K)
    dt, seed = 0.04, 20260202
    kw = dict(sum_bits=29) if fp32 else {}
    orc = so.SyncOracle(par, dt=dt, seed=seed, **kw)
    orc.set_state(pos, spin)
    h = make_handle(capi, par, N, dt=dt, seed=seed, method="tiles", fp32=fp32)
    os.environ["APS_LOOP_MIN"] = "3"                     # (a grid of more than one workgroup per CU starts at 10 steps by default)
    try:
        h.set_state(pos, spin)
        total = 0
        for n in (7, 4, 1, 33, 2, 50, 3):               # odd and even calls, calls too short for the loop, observations in between
            h.step(n)
            orc.run(n)
            total += n
            taken, state, why = h.loop_info()
            assert state == 1, (tag, why)
            assert taken == (0 if n < 3 else n), (tag, n, taken)
            same_state(h, orc, (tag, total))
            check_lattice(h, orc)
        t, k = h.time()
        assert k == total
        assert not np.array_equal(h.get_state()[0], pos)
        assert np.array_equal(h.exits(), orc.exits())
        if case.get("k_exit", 0.0) > 0.0:
            assert (orc.alive == 0).sum() > 10, tag
    finally:
        del os.environ["APS_LOOP_MIN"]
        h.close()


def test_resident_loop_equals_one_launch_per_step_at_config2(capi):
    """BASELINE config 2 at full size (N = 1e5, L = 2e5, K = 1, 4001-tap table): 512 tiles resident at once.  The loop
    against the per-step path over 401 steps (state, {W, S}, occupancy on all sites) and against the oracle over 5."""
    L, N = 200000, 100000
    par = params(L=L, K=1, sigma=0.005, rate_diffusion=0.02, rate_active=5.0, beta=0.7)
    rng = np.random.default_rng(3)
    pos = rng.choice(L, size=N, replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    dt, seed = 0.0125, 99
    orc = so.SyncOracle(par, dt=dt, seed=seed)
    orc.set_state(pos, spin)
    a = make_handle(capi, par, N, dt=dt, seed=seed, method="tiles")
    b = make_handle(capi, par, N, dt=dt, seed=seed, method="tiles")
    try:
        b.set_resident_loop(False)
        a.set_state(pos, spin)
        b.set_state(pos, spin)
        os.environ["APS_LOOP_MIN"] = "3"                 # a full device takes the loop from 10 steps on by default
        try:
            a.step(5)
        finally:
            del os.environ["APS_LOOP_MIN"]
        b.step(5)
        orc.run(5)
        assert a.loop_info()[:2] == (5, 1), a.loop_info()
        assert b.loop_info()[0] == 0
        same_state(a, orc)
        same_state(b, orc)
        check_lattice(a, orc)
        a.step(6)                                        # default: a full device takes the loop from 10 steps on
        b.step(6)
        assert a.loop_info()[0] == 0 and a.step_info()[0] == 6
        for n in (200, 190):
            a.step(n)
            b.step(n)
            assert a.loop_info()[0] == n
            for x, y in zip(a.get_state(), b.get_state()):
                assert np.array_equal(x, y)
            for x, y in zip(a.get_lattice(), b.get_lattice()):
                assert np.array_equal(x, y)
        assert a.time()[1] == 401
    finally:
        a.close()
        b.close()


def test_resident_loop_ensembles(capi):
    par = params(L=2400, K=1, sigma=0.01)
    betas = [0.3, 1.1, 2.0]
    rng = np.random.default_rng(5)
    N = 1100
    states = [random_state(rng, par.L, N, par.K) for _ in betas]
    dt, seed = 0.04, 77
    h = make_handle(capi, par, N, dt=dt, seed=seed, method="tiles", beta=betas)
    try:
        for e, (p, s) in enumerate(states):
            h.set_state(p, s, ensemble=e)
        h.step(41)
        assert h.loop_info()[:2] == (41, 1), h.loop_info()
        for e, (p, s) in enumerate(states):
            pe = params(L=2400, K=1, sigma=0.01, beta=betas[e])
            orc = so.SyncOracle(pe, dt=dt, seed=seed, ensemble=e)
            orc.set_state(p, s)
            orc.run(41)
            got = h.get_state(ensemble=e)
            assert np.array_equal(got[0], orc.pos) and np.array_equal(got[1], orc.spin), e
            check_lattice(h, orc, ensemble=e)
    finally:
        h.close()


def test_not_eligible_falls_back_silently(capi):
    """A global mean field: one launch per step, same results."""
    for case, reason in [(dict(L=600, K=1, sigma=0.0), "global")]:
        par = params(**case)
        rng = np.random.default_rng(2)
        N = par.L * par.K // 2
        pos, spin = random_state(rng, par.L, N, par.K)
        orc = so.SyncOracle(par, dt=0.04, seed=4)
        orc.set_state(pos, spin)
        h = make_handle(capi, par, N, dt=0.04, seed=4, method="tiles")
        try:
            h.set_state(pos, spin)
            h.step(25)
            orc.run(25)
            taken, state, why = h.loop_info()
            assert (taken, state) == (0, 0) and reason in why, (why, reason)
            same_state(h, orc)
        finally:
            h.close()


def test_a_call_that_gives_up_is_repeated_the_ordinary_way(capi):
    """APS_LOOP_TEST_ABORT raises the "a wait ran out" word before the launch: every workgroup leaves at once, the host
    finds the flag, repeats the steps with one launch per step from the untouched inputs and never tries the loop again."""
    par = params(L=3000, K=1, sigma=0.01)
    rng = np.random.default_rng(8)
    N = 1500
    pos, spin = random_state(rng, par.L, N, par.K)
    orc = so.SyncOracle(par, dt=0.04, seed=6)
    orc.set_state(pos, spin)
    h = make_handle(capi, par, N, dt=0.04, seed=6, method="tiles")
    try:
        h.set_state(pos, spin)
        h.step(9)
        orc.run(9)
        assert h.loop_info()[:2] == (9, 1)
        os.environ["APS_LOOP_TEST_ABORT"] = "1"
        try:
            h.step(10)
        finally:
            del os.environ["APS_LOOP_TEST_ABORT"]
        orc.run(10)
        taken, state, why = h.loop_info()
        assert (taken, state) == (0, -1) and "ran out" in why
        same_state(h, orc)
        check_lattice(h, orc)
        h.step(21)
        orc.run(21)
        assert h.loop_info()[:2] == (0, -1)
        same_state(h, orc)
    finally:
        h.close()


@pytest.mark.parametrize("case", [
    # config 2's geometry (two workgroups on every CU) at reduced size, two ensembles, an EVEN call: ensemble 1 and the far tiles of
    # ensemble 0 finish all n steps and write their final state and the step word before the call is given up
    dict(tag="k1_two_ensembles_even", L=70000, K=1, sigma=0.002, betas=[0.7, 1.9], frac=0.4, n=12, stall="200:4"),
    dict(tag="k2_odd", L=3000, K=2, sigma=0.01, betas=[1.1], frac=0.6, n=9, stall="3:2"),
    dict(tag="k1_tile_never_starts", L=3000, K=1, sigma=0.01, betas=[0.7], frac=0.5, n=10, stall="5:0"),
    # particles leave while the call is under way: the far tiles have logged exits of steps that are then repeated -- the exit
    # counts of before the call are put back, the repeated steps log the same rows again
    dict(tag="k2_exits_logged_before_giving_up", L=3000, K=2, sigma=0.01, betas=[1.1], frac=0.6, n=30, stall="3:20",
         extra=dict(anchor_positions=[0.3, 0.7], anchor_radius=0.08, k_on=4.0, k_off=0.5, k_exit=2.0)),
], ids=lambda c: c["tag"])
def test_a_wait_that_runs_out_mid_loop(capi, case):
    """APS_LOOP_TEST_STALL=<tile>:<iteration>: that tile leaves at the top of that iteration without its record and without
    raising the give-up word -- a workgroup that is not resident.  Its neighbours' waits run out (2 ms here) while other tiles
    are iterations ahead or already done; everyone leaves, aps_step repeats the call with one launch per step from the intact
    inputs: same bits as the oracle after the call and after 20 further steps (the device step words included: the random
    numbers of the repeated steps are those of the right step indices)."""
    extra = case.get("extra", {})
    par0 = params(L=case["L"], K=case["K"], sigma=case["sigma"], **extra)
    betas = case["betas"]
    rng = np.random.default_rng(21)
    N = max(1, int(case["frac"] * par0.L * par0.K))
    states = [random_state(rng, par0.L, N, par0.K) for _ in betas]
    dt, seed = 0.04, 31
    orcs = []
    for e, b in enumerate(betas):
        orc = so.SyncOracle(params(L=case["L"], K=case["K"], sigma=case["sigma"], beta=b, **extra), dt=dt, seed=seed, ensemble=e)
        orc.set_state(*states[e])
        orcs.append(orc)
    h = make_handle(capi, par0, N, dt=dt, seed=seed, method="tiles", beta=betas)
    os.environ["APS_LOOP_MIN"] = "3"
    try:
        for e, (p, s) in enumerate(states):
            h.set_state(p, s, ensemble=e)

        def same(tag):
            for e, orc in enumerate(orcs):
                got = h.get_state(ensemble=e)
                assert np.array_equal(got[0], orc.pos) and np.array_equal(got[1], orc.spin) and np.array_equal(got[2], orc.bound), (tag, e)
                assert np.array_equal(got[3], orc.alive) and np.array_equal(h.exits(ensemble=e), orc.exits()), (tag, e)
                check_lattice(h, orc, ensemble=e)

        h.step(5)                                            # the loop works on this handle
        for orc in orcs:
            orc.run(5)
        assert h.loop_info()[:2] == (5, 1), h.loop_info()
        same("before")
        os.environ["APS_LOOP_TEST_STALL"], os.environ["APS_LOOP_TIMEOUT_MS"] = case["stall"], "2"
        try:
            h.step(case["n"])
        finally:
            del os.environ["APS_LOOP_TEST_STALL"], os.environ["APS_LOOP_TIMEOUT_MS"]
        for orc in orcs:
            orc.run(case["n"])
        taken, state, why = h.loop_info()
        assert (taken, state) == (0, -1) and "ran out" in why, (taken, state, why)
        assert h.time()[1] == 5 + case["n"]
        same("after the repeated call")
        h.step(20)
        for orc in orcs:
            orc.run(20)
        assert h.loop_info()[:2] == (0, -1)
        same("20 steps later")
        if extra:
            assert sum(int((orc.alive == 0).sum()) for orc in orcs) > 10
    finally:
        del os.environ["APS_LOOP_MIN"]
        h.close()
