"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bars (DESIGN.md "Parity"):
  weight table, S, W, occ4, m-field .... bit-exact vs oracle/sync_oracle.c (sums are exact on the 2^-q grid)
  integer state (pos, sigma, bound, alive) after every step .... bit-exact vs the oracle
  m-field vs the REFERENCE fixture G1 .... <= 2e-11 (weight-grid rounding, same bound as the oracle's)
The oracle uses the lattice formulation recomputed from scratch every step (histogram + windowed stencil); the GPU
is run in BOTH of its formulations (`method`): the all-pairs kernel, and the lattice field maintained incrementally
(whose W, S and occupancy arrays are additionally compared with the oracle's recomputation after the runs)."""
import importlib

import os

import numpy as np
import pytest

from oracle.gillespie_numpy import LatticeGasParams
from oracle import sync_oracle as so

pytestmark = pytest.mark.gpu
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"


@pytest.fixture(scope="module")
def capi():
    mod = importlib.import_module(PKG + ".capi")
    assert mod.device_count() >= 1, "no GPU visible"
    return mod


@pytest.fixture(params=["pairs", "lattice", "tiles"])
def method(request):
    return request.param


def check_lattice(h, orc, ensemble=0):
    """The incrementally maintained lattice arrays equal the oracle's from-scratch recomputation, bit for bit."""
    if h.method not in ("lattice", "tiles"):
        return
    W, S, occ = h.get_lattice(ensemble)
    cp0, cm0, _ = orc.field_sites()
    assert np.array_equal(occ, cp0 + cm0)
    if orc.par.sigma_kernel > 0:
        S0, W0 = orc.last_site_sums
        assert np.array_equal(W, W0)
        assert np.array_equal(S, S0)
    Sp, Wp, occ4 = h.lattice_accumulate(ensemble)
    S1, W1, occ1 = orc.pair_sums()
    assert np.array_equal(occ4, occ1)
    if orc.par.sigma_kernel > 0:
        assert np.array_equal(Sp, S1) and np.array_equal(Wp, W1)


def make_handle(capi, par, n, dt=0.05, seed=1, beta=None, **kw):
    return capi.Handle(L=par.L, K=par.K, periodic=par.periodic, sigma_grid=par.sigma_grid if par.sigma_kernel > 0 else 0.0,
                       rate_diffusion=par.rate_diffusion, rate_active=par.rate_active,
                       beta=[par.beta] if beta is None else beta, dt=dt, seed=seed, n_particles=n,
                       minus_anchor=par.minus_anchor, immobilize=par.immobilize_when_anchored,
                       suppress_flip=par.suppress_flip_when_bound, crowding=par.crowding_suppresses_rates,
                       k_on=par.k_on, k_off=par.k_off, k_exit=par.k_exit, anchor_mask=par.is_anchor_site, **kw)


def random_state(rng, L, N, K):
    slots = np.repeat(np.arange(L), K)
    pos = rng.permutation(rng.choice(slots, size=N, replace=False)).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    return pos, spin


def params(L, K=1, sigma=0.02, periodic=False, **kw):
    base = dict(xlim=1.0, rate_diffusion=0.6, rate_active=4.0, beta=1.1, scale_rates=False)
    base.update(kw)
    return LatticeGasParams.from_kwargs(L=L, local_kernel_sigma=sigma, periodic=periodic, site_capacity=K, **base)


FIELD_CASES = [
    dict(L=64, K=1, sigma=0.005), dict(L=64, K=3, sigma=0.3), dict(L=400, K=1, sigma=0.02),
    dict(L=400, K=2, sigma=0.02, periodic=True), dict(L=1000, K=3, sigma=0.005, periodic=True),
    dict(L=1000, K=1, sigma=0.3), dict(L=1000, K=1, sigma=0.0), dict(L=997, K=2, sigma=0.05),
    dict(L=5000, K=1, sigma=0.005), dict(L=3000, K=1, sigma=0.2, periodic=True),
]


@pytest.mark.parametrize("case", FIELD_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
@pytest.mark.parametrize("sort", [True, False], ids=["sorted", "unsorted"])
def test_table_and_pair_sums_bit_exact(capi, case, sort, method):
    par = params(**case)
    rng = np.random.default_rng(42)
    N = int(0.55 * par.L * par.K)
    pos, spin = random_state(rng, par.L, N, par.K)
    alive = (rng.random(N) > 0.1).astype(np.uint8)            # some dead particles must be inert
    orc = so.SyncOracle(par, dt=0.05, seed=1)
    orc.set_state(pos, spin, alive=alive)
    h = make_handle(capi, par, N, sort_by_site=sort, method=method)
    try:
        assert h.method == method
        tab, q = h.table()
        if par.sigma_kernel > 0:
            assert q == orc.q and np.array_equal(tab, orc.table)
        h.set_state(pos, spin, alive=alive)
        check_lattice(h, orc)
        S, W, occ4 = h.pair_accumulate()
        S0, W0, occ0 = orc.pair_sums()
        assert np.array_equal(W, W0)
        assert np.array_equal(S, S0)
        assert np.array_equal(occ4, occ0)
        cp, cm, m = h.observe()
        cp0, cm0, m0 = orc.field_sites()
        assert np.array_equal(cp, cp0) and np.array_equal(cm, cm0)
        assert np.array_equal(m, m0)
        # caller-supplied histogram path (compute_local_m_field)
        assert np.array_equal(h.field_from_counts(cp0, cm0), m0)
    finally:
        h.close()


def test_m_field_vs_reference_fixture(capi, golden):
    g = golden("g1_mfield.npz")
    base = g.meta["base_kw"]
    worst = 0.0
    for idx, c in enumerate(g.meta["cases"]):
        par = LatticeGasParams.from_kwargs(L=c["L"], local_kernel_sigma=c["sigma"], periodic=c["periodic"],
                                           site_capacity=c["K"], **base)
        pos, sigma, want = g[f"c{idx}_pos"], g[f"c{idx}_sigma"], g[f"c{idx}_m"]
        h = make_handle(capi, par, len(pos))
        try:
            h.set_state(pos, sigma)
            m = h.observe()[2]
        finally:
            h.close()
        if c["periodic"] and c["sigma"] > 0:
            # the reference's FFT branch is round-off dominated where the Gaussian mass is tiny;
            # weight the comparison by min(W, 1) exactly as tests/test_oracle_sync.py does
            orc = so.SyncOracle(par, dt=0.05, seed=1)
            orc.set_state(pos, sigma)
            orc.field_sites()
            wgt = np.minimum(orc.last_site_sums[1], 1.0)
        else:
            wgt = 1.0
        err = np.max(np.abs(m - want) * wgt)
        worst = max(worst, err)
        assert err <= 2e-11, (c, err)
    print("worst |m - m_ref| =", worst)


STEP_CASES = [
    dict(tag="k1_reflect", L=300, K=1, sigma=0.02, frac=0.5),
    dict(tag="k1_periodic", L=300, K=1, sigma=0.03, periodic=True, frac=0.5),
    dict(tag="k3_reflect_wide", L=200, K=3, sigma=0.3, frac=0.6),
    dict(tag="k2_crowding", L=250, K=2, sigma=0.01, frac=0.7, crowding_suppresses_rates=True),
    dict(tag="global_field", L=400, K=1, sigma=0.0, frac=0.45, beta=1.6),
    dict(tag="anchors_exit", L=200, K=2, sigma=0.02, frac=0.5, anchor_positions=[0.3, 0.7], anchor_radius=0.08,
         k_on=3.0, k_off=1.0, k_exit=2.0),
    dict(tag="anchors_free_minus_periodic", L=200, K=2, sigma=0.02, frac=0.5, periodic=True, anchor_positions=[0.5],
         anchor_radius=0.1, k_on=2.0, k_off=1.0, k_exit=1.0, minus_anchor=False, immobilize_when_anchored=False,
         suppress_flip_when_bound=False),
    dict(tag="dense_diffusive", L=128, K=1, sigma=0.05, frac=0.9, rate_diffusion=6.0),
    dict(tag="tiny", L=2, K=1, sigma=0.0, frac=0.5, periodic=False),
]


@pytest.mark.parametrize("case", STEP_CASES, ids=lambda c: c["tag"])
@pytest.mark.parametrize("sort", [True, False], ids=["sorted", "unsorted"])
def test_trajectory_bit_exact_every_step(capi, case, sort, method):
    case = dict(case)
    tag, frac = case.pop("tag"), case.pop("frac")
    par = params(**case)
    rng = np.random.default_rng(7)
    N = max(1, int(frac * par.L * par.K))
    pos, spin = random_state(rng, par.L, N, par.K)
    dt, seed = 0.04, 20260101
    orc = so.SyncOracle(par, dt=dt, seed=seed)
    orc.set_state(pos, spin)
    h = make_handle(capi, par, N, dt=dt, seed=seed, sort_by_site=sort, method=method)
    try:
        h.set_state(pos, spin)
        nsteps = 120
        for s in range(nsteps):
            orc.step()
            h.step(1)
            p, sg, bd, al = h.get_state()
            assert np.array_equal(al, orc.alive), (tag, s)
            assert np.array_equal(p, orc.pos), (tag, s)
            assert np.array_equal(sg, orc.spin), (tag, s)
            assert np.array_equal(bd, orc.bound), (tag, s)
            if s == 60:
                h.resort()                      # re-sorting must not change anything observable
            if s % 40 == 39:
                check_lattice(h, orc)
        t, k = h.time()
        assert k == nsteps and t == nsteps * dt
        ex, ex0 = h.exits(), orc.exits()
        assert len(ex) == len(ex0)
        if len(ex0):
            order = np.lexsort((ex0[:, 2], ex0[:, 0]))
            assert np.array_equal(ex, ex0[order])
        assert (np.bincount(p[al == 1], minlength=par.L).max() if al.any() else 0) <= par.K
        assert not np.array_equal(p, pos) or N <= 1
    finally:
        h.close()


def test_propose_commit_halves_equal_step(capi, method):
    par = params(L=500, K=2, sigma=0.02)
    rng = np.random.default_rng(3)
    pos, spin = random_state(rng, 500, 600, 2)
    a = make_handle(capi, par, 600, seed=5, method=method)
    b = make_handle(capi, par, 600, seed=5, method=method)
    try:
        a.set_state(pos, spin)
        b.set_state(pos, spin)
        a.step(40)
        for _ in range(40):
            b.propose()
            b.commit()
        for x, y in zip(a.get_state(), b.get_state()):
            assert np.array_equal(x, y)
        ptr, total, off, mine = b.exchange_buffer()
        assert ptr and off == 0 and total == mine and mine >= 600
    finally:
        a.close()
        b.close()


def test_ensembles_are_independent_and_match_single_runs(capi, method):
    """BASELINE config 4 shape: E ensembles with their own beta in one handle == E separate handles."""
    par = params(L=400, K=1, sigma=0.02)
    rng = np.random.default_rng(9)
    betas = [0.0, 0.7, 1.5, 3.0]
    states = [random_state(rng, 400, 180, 1) for _ in betas]
    big = make_handle(capi, par, 180, seed=11, beta=betas, method=method)
    try:
        for e, (p, s) in enumerate(states):
            big.set_state(p, s, ensemble=e)
        big.step(60)
        for e, (p, s) in enumerate(states):
            one = make_handle(capi, par, 180, seed=11, beta=[betas[e]], ensemble_base=e, method=method)
            orc_par = params(L=400, K=1, sigma=0.02, beta=betas[e])
            orc = so.SyncOracle(orc_par, dt=0.05, seed=11, ensemble=e)
            try:
                one.set_state(p, s)
                one.step(60)
                orc.set_state(p, s)
                orc.run(60)
                got = big.get_state(ensemble=e)
                for x, y in zip(got, one.get_state()):
                    assert np.array_equal(x, y)
                assert np.array_equal(got[0], orc.pos) and np.array_equal(got[1], orc.spin)
                check_lattice(big, orc, ensemble=e)
            finally:
                one.close()
    finally:
        big.close()


def test_error_paths(capi):
    par = params(L=100, K=1)
    with pytest.raises(capi.ApsError):
        make_handle(capi, par, 1000)                       # more particles than K*L
    h = make_handle(capi, par, 10)
    try:
        with pytest.raises(capi.ApsError):
            h.step(1)                                      # no state uploaded
        with pytest.raises(capi.ApsError):
            h.set_state(np.array([5, 5], np.int32), np.array([1, -1], np.int8))   # capacity exceeded
        with pytest.raises(capi.ApsError):
            h.set_state(np.array([100], np.int32), np.array([1], np.int8))        # outside the lattice
        h.set_state(np.array([5, 6], np.int32), np.array([1, -1], np.int8))
        h.step(3)
    finally:
        h.close()


_FULL_SIZE_ORACLES = {}


def _full_size_oracles(par, pos, spin):
    """The oracle's trajectory of the BASELINE-size system (about 2 s per step on the CPU) is the same for every formulation:
    stepped once per session, snapshots at steps 0, 15 and 49 as oracles of their own."""
    if not _FULL_SIZE_ORACLES:
        orc = so.SyncOracle(par, dt=0.0125, seed=0)
        orc.set_state(pos, spin)
        done = 0
        for upto in (0, 15, 49):
            orc.run(upto - done)
            done = upto
            snap = so.SyncOracle(par, dt=0.0125, seed=0)
            snap.set_state(orc.pos, orc.spin, bound=orc.bound, alive=orc.alive)
            snap.step_index = orc.step_index
            _FULL_SIZE_ORACLES[upto] = snap
    return _FULL_SIZE_ORACLES


def test_full_size_properties(capi, method):
    """BASELINE config 2 shape (N=1e5, L=2e5, K=1, sigma_g=1000): properties that need no oracle run:
    exclusion, conservation, |dx| <= 1 per step, sorted == unsorted bit for bit, a spot check of
    S/W against the oracle on a subset of targets via a small window state is covered elsewhere."""
    L, N = 200_000, 100_000
    par = LatticeGasParams.from_kwargs(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7,
                                       scale_rates=False, local_kernel_sigma=0.005, site_capacity=1)
    rng = np.random.default_rng(0)
    pos = rng.choice(L, size=N, replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    a = make_handle(capi, par, N, dt=0.0125, seed=0, sort_by_site=True, method=method)
    b = make_handle(capi, par, N, dt=0.0125, seed=0, sort_by_site=False, method=method)
    try:
        a.set_state(pos, spin)
        b.set_state(pos, spin)
        # S and W at full size against the oracle (lattice formulation, ~1 s on the CPU)
        orcs = _full_size_oracles(par, pos, spin)
        orc = orcs[0]
        S0, W0, occ0 = orc.pair_sums()
        S, W, occ4 = a.pair_accumulate()
        assert np.array_equal(S, S0) and np.array_equal(W, W0) and np.array_equal(occ4, occ0)
        prev = pos.copy()
        for _ in range(3):
            a.step(5)
            b.step(1); b.step(4)
            pa, sa, ba, aa = a.get_state()
            pb, sb, bb, ab = b.get_state()
            assert np.array_equal(pa, pb) and np.array_equal(sa, sb)
            assert aa.all() and np.bincount(pa, minlength=L).max() <= 1
            assert np.abs(pa.astype(int) - prev).max() <= 5
            prev = pa.astype(int)
        orc = orcs[15]
        assert np.array_equal(pa, orc.pos) and np.array_equal(sa, orc.spin)
        check_lattice(a, orc)
        if method in ("lattice", "tiles"):      # through the graph-replay path (>= 33 steps per call), field still exact
            a.step(34)                          # (the oracle needs ~2 s per step at this size)
            b.step(34)
            orc = orcs[49]
            pa, sa, _, _ = a.get_state()
            assert np.array_equal(pa, orc.pos) and np.array_equal(sa, orc.spin)
            assert np.array_equal(pa, b.get_state()[0])
            check_lattice(a, orc)
            check_lattice(b, orc)
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("method", ["lattice", "tiles"])
def test_config2_ten_thousand_steps_field_stays_exact(capi, method):
    """BASELINE config 2 at its full length (N = 1e5, 10 000 steps, both incremental lattice formulations): after 10 000 incremental updates the
    smoothed histograms W, S on all 2e5 sites still equal a from-scratch recomputation from the final state bit for bit
    (no drift: every value sits on the weight grid), the run is independent of how it is cut into calls (one call = 312
    graph replays + a tail; 16 calls of 625 steps), exclusion holds and nobody is lost."""
    L, N = 200_000, 100_000
    par = LatticeGasParams.from_kwargs(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7,
                                       scale_rates=False, local_kernel_sigma=0.005, site_capacity=1)
    rng = np.random.default_rng(5)
    pos = rng.choice(L, size=N, replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    a = make_handle(capi, par, N, dt=0.0125, seed=3, method=method)
    b = make_handle(capi, par, N, dt=0.0125, seed=3, method=method)
    try:
        a.set_state(pos, spin)
        b.set_state(pos, spin)
        a.step(10_000)
        for _ in range(16):
            b.step(625)
        pa, sa, _, alive = a.get_state()
        pb, sb, _, _ = b.get_state()
        assert np.array_equal(pa, pb) and np.array_equal(sa, sb)
        assert alive.all() and np.bincount(pa, minlength=L).max() <= 1
        assert (pa != pos).mean() > 0.5                      # the system has moved
        orc = so.SyncOracle(par, dt=0.0125, seed=3)
        orc.set_state(pa, sa)
        cp0, cm0, _ = orc.field_sites()
        S0, W0 = orc.last_site_sums
        for h in (a, b):
            W, S, occ = h.get_lattice(0)
            assert np.array_equal(occ, cp0 + cm0) and np.array_equal(W, W0) and np.array_equal(S, S0)
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("method", ["lattice", "tiles"])
def test_config5_scale_incremental_field_equals_from_scratch(capi, method):
    """BASELINE config 5 scale (N = 1e6, L = 2e6, 40 001-entry table = 320 KB, beyond LDS): the field kept incrementally by
    `field_update` (table windows in LDS for interior tiles, global gathers next to the walls) over 300 steps equals the one
    a fresh handle builds from scratch from the final state with the `field_sites` kernel (itself pinned to the oracle at
    the sizes the oracle can do), bit for bit on all 2e6 sites.  Too large for the CPU oracle: 8e10 table terms per field."""
    L, N = 2_000_000, 1_000_000
    par = LatticeGasParams.from_kwargs(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7,
                                       scale_rates=False, local_kernel_sigma=0.005, site_capacity=1)
    rng = np.random.default_rng(11)
    pos = rng.choice(L, size=N, replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    a = make_handle(capi, par, N, dt=0.0125, seed=4, method=method)
    b = make_handle(capi, par, N, dt=0.0125, seed=4, method=method)
    try:
        assert len(a.table()[0]) == 40001
        a.set_state(pos, spin)
        a.step(300)
        pa, sa, _, alive = a.get_state()
        assert alive.all() and np.bincount(pa, minlength=L).max() <= 1 and (pa != pos).mean() > 0.3
        b.set_state(pa, sa)
        W, S, occ = a.get_lattice(0)
        W0, S0, occ0 = b.get_lattice(0)
        assert np.array_equal(occ, occ0) and np.array_equal(occ, np.bincount(pa, minlength=L))
        assert np.array_equal(W, W0) and np.array_equal(S, S0)
        assert W.min() > 0.0 and np.abs(S).max() <= W.max()
    finally:
        a.close()
        b.close()


def test_two_rank_shards_emulated_on_one_gpu(capi, method):
    """world=2 on ONE device: each handle evaluates its own particle shard, the proposal blocks are swapped
    by hand (what the all-gather does), both commit everything -> identical states, equal to world=1."""
    if method == "tiles":
        pytest.skip("particle-index shards are a protocol of the particle-indexed formulations")
    torch = pytest.importorskip("torch")
    par = params(L=2000, K=2, sigma=0.02, anchor_positions=[0.5], anchor_radius=0.05, k_on=2.0, k_off=1.0, k_exit=0.5)
    rng = np.random.default_rng(21)
    N = 1500
    pos, spin = random_state(rng, par.L, N, par.K)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ranks = []
    for r in range(2):
        h = make_handle(capi, par, N, dt=0.04, seed=8, rank=r, world=2, method=method)
        h.set_state(pos, spin)
        _, total, off, mine = h.exchange_buffer()
        buf = torch.zeros(total, dtype=torch.uint8, device=dev)
        h.set_stream(stream)
        h.bind_exchange_buffer(buf.data_ptr(), total)
        ranks.append((h, buf, off, mine))
    single = make_handle(capi, par, N, dt=0.04, seed=8, method=method)
    try:
        single.set_state(pos, spin)
        assert ranks[0][2] == 0 and ranks[1][2] == ranks[0][3]
        with pytest.raises(capi.ApsError):
            ranks[0][0].step(1)                           # aps_step refuses sharded handles
        for _ in range(60):
            for h, _, _, _ in ranks:
                h.propose()
            (h0, b0, o0, m0), (h1, b1, o1, m1) = ranks
            b0[o1:o1 + m1] = b1[o1:o1 + m1]
            b1[o0:o0 + m0] = b0[o0:o0 + m0]
            for h, _, _, _ in ranks:
                h.commit()
        torch.cuda.synchronize()
        single.step(60)
        want = single.get_state()
        for h, _, _, _ in ranks:
            for x, y in zip(h.get_state(), want):
                assert np.array_equal(x, y)
        assert np.array_equal(ranks[0][0].exits(), single.exits())
        if method == "lattice":
            for h, _, _, _ in ranks:
                for x, y in zip(h.get_lattice(), single.get_lattice()):
                    assert np.array_equal(x, y)
    finally:
        for h, _, _, _ in ranks:
            h.close()
        single.close()


def test_hip_engine_over_nccl_world1(capi):
    """The production multi-GPU path (ShardedStepper + HipEngine + torch.distributed 'nccl' = RCCL) at world
    size 1: exercises the torch-tensor <-> C-ABI pointer hand-over, the shared stream and the collective call."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    sharded = importlib.import_module(PKG + ".sharded")
    par = params(L=1000, K=1, sigma=0.02)
    rng = np.random.default_rng(4)
    pos, spin = random_state(rng, 1000, 450, 1)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1)
    a = make_handle(capi, par, 450, seed=12)
    b = make_handle(capi, par, 450, seed=12)
    try:
        a.set_state(pos, spin)
        b.set_state(pos, spin)
        stepper = sharded.ShardedStepper(sharded.HipEngine(a, torch.device("cuda", 0)))
        assert stepper.world == 1
        stepper.step(50)
        stepper.dist.all_gather_into_tensor(stepper.buf, stepper.mine)     # the collective itself, in place
        torch.cuda.synchronize()
        b.step(50)
        for x, y in zip(a.get_state(), b.get_state()):
            assert np.array_equal(x, y)
    finally:
        a.close()
        b.close()
        dist.destroy_process_group()


def test_inlibrary_rccl_allgather_world1(capi, method):
    """aps_comm_init + aps_step with the library's own ncclAllGather (world size 1: the collective runs in place
    on one rank) must equal the plain single-GPU stepping."""
    if method == "tiles":
        pytest.skip("proposal all-gather belongs to the particle-indexed formulations")
    pytest.importorskip("torch")                          # as in production: torch's librccl is the process's RCCL
    par = params(L=1500, K=2, sigma=0.02)
    rng = np.random.default_rng(6)
    pos, spin = random_state(rng, 1500, 900, 2)
    a = make_handle(capi, par, 900, seed=3, method=method)
    b = make_handle(capi, par, 900, seed=3, method=method)
    try:
        a.set_state(pos, spin)
        b.set_state(pos, spin)
        ident = capi.comm_unique_id()
        assert len(ident) == 128
        a.comm_init(ident)
        with pytest.raises(capi.ApsError):
            a.comm_init(ident)                            # second initialisation is refused
        a.step(40)
        b.step(40)
        for x, y in zip(a.get_state(), b.get_state()):
            assert np.array_equal(x, y)
    finally:
        a.close()
        b.close()


def test_table_too_large_for_lds_uses_global_table(capi, method):
    """sigma_g = 6000 -> 24001-entry table (188 KB) does not fit LDS: the kernels gather from the table in
    global memory instead.  Same bit-exact bars."""
    par = params(L=60000, K=1, sigma=0.1)
    rng = np.random.default_rng(31)
    N = 3000
    pos, spin = random_state(rng, par.L, N, par.K)
    orc = so.SyncOracle(par, dt=0.05, seed=9)
    orc.set_state(pos, spin)
    h = make_handle(capi, par, N, seed=9, method=method)
    try:
        tab, q = h.table()
        assert len(tab) == 24001 and q == orc.q and np.array_equal(tab, orc.table)
        h.set_state(pos, spin)
        S, W, occ4 = h.pair_accumulate()
        S0, W0, occ0 = orc.pair_sums()
        assert np.array_equal(S, S0) and np.array_equal(W, W0) and np.array_equal(occ4, occ0)
        h.step(10)
        orc.run(10)
        p, sg, bd, al = h.get_state()
        assert np.array_equal(p, orc.pos) and np.array_equal(sg, orc.spin)
        m = h.observe()[2]
        assert np.array_equal(m, orc.field_sites()[2])
        check_lattice(h, orc)
    finally:
        h.close()


@pytest.mark.parametrize("periodic", [False, True], ids=["walls", "torus"])
def test_windowed_table_dense_buckets(capi, periodic):
    """Table beyond LDS (20 001 entries) with K = 3 at density 1.5 and a large dt: about 75 deposits per 256-site bucket
    and step, so the list loads of `field_update` take several rounds per bucket, the wave segments are flushed inside a
    group, and about 39 interior tiles go through the double-buffered table windows (walls; the torus, whose table has
    L/2 + 1 entries, gathers from the table in global memory).  Same bit-exact bars."""
    par = params(L=60000, K=3, sigma=5000.0 / 60000, periodic=periodic)
    rng = np.random.default_rng(77)
    N = 90000
    pos, spin = random_state(rng, par.L, N, par.K)
    orc = so.SyncOracle(par, dt=0.05, seed=21)
    orc.set_state(pos, spin)
    h = make_handle(capi, par, N, seed=21, method="lattice")
    try:
        assert len(h.table()[0]) == (30001 if periodic else 20001)
        h.set_state(pos, spin)
        for _ in range(2):
            h.step(1)
            orc.run(1)
            p, sg, bd, al = h.get_state()
            assert np.array_equal(p, orc.pos) and np.array_equal(sg, orc.spin)
            W, S, occ = h.get_lattice(0)
            cp0, cm0, _ = orc.field_sites()
            S0, W0 = orc.last_site_sums
            assert np.array_equal(occ, cp0 + cm0) and np.array_equal(W, W0) and np.array_equal(S, S0)
    finally:
        h.close()


def _random_case(rng):
    L = int(rng.choice([2, 3, 5, 17, 64, 129, 300, 777]))
    K = int(rng.integers(1, 6))
    sigma = float(rng.choice([0.0, 0.3 / L, 2.0 / L, 0.02, 0.11, 0.4]))
    kw = dict(L=L, K=K, sigma=sigma, periodic=bool(rng.integers(2)),
              rate_diffusion=float(rng.choice([0.0, 0.05, 1.5, 9.0])), rate_active=float(rng.choice([0.0, 0.7, 6.0])),
              beta=float(rng.choice([0.0, 0.4, 1.7, 4.0])), minus_anchor=bool(rng.integers(2)),
              immobilize_when_anchored=bool(rng.integers(2)), suppress_flip_when_bound=bool(rng.integers(2)),
              crowding_suppresses_rates=bool(rng.integers(2)))
    if rng.integers(2):
        kw.update(anchor_positions=list(rng.random(int(rng.integers(1, 4)))), anchor_radius=float(rng.choice([0.0, 0.02, 0.2])),
                  k_on=float(rng.choice([0.0, 0.5, 4.0])), k_off=float(rng.choice([0.0, 0.3, 2.0])),
                  k_exit=float(rng.choice([0.0, 0.4, 3.0])))
    else:
        kw.update(k_on=0.0, k_off=0.0, k_exit=0.0)
    return kw


def test_randomised_parameter_sweep_bit_exact(capi, method):
    """32 random parameter sets (tiny to medium lattices, K up to 5, every field mode, anchors, crowding, zero
    rates, wrap-around, particles dead from the start) -- integer state bit-exact against the oracle along 60 steps,
    S/W/occupancy and the m-field bit-exact at the start."""
    master = np.random.default_rng(20261004)
    for case_no in range(32):
        rng = np.random.default_rng(master.integers(2 ** 32))
        kw = _random_case(rng)
        par = params(**kw)
        N = int(max(1, rng.integers(1, par.L * par.K + 1) * rng.choice([0.2, 0.6, 1.0])))
        pos, spin = random_state(rng, par.L, N, par.K)
        alive = (rng.random(N) > 0.15).astype(np.uint8) if case_no % 3 == 0 else None
        bound = (rng.random(N) > 0.6).astype(np.uint8) if case_no % 2 == 0 else None
        dt, seed = float(rng.choice([0.002, 0.03, 0.2])), int(rng.integers(2 ** 62))
        orc = so.SyncOracle(par, dt=dt, seed=seed)
        orc.set_state(pos, spin, bound=bound, alive=alive)
        h = make_handle(capi, par, N, dt=dt, seed=seed, sort_by_site=bool(case_no % 2), method=method)
        try:
            h.set_state(pos, spin, bound=bound, alive=alive)
            S, W, occ4 = h.pair_accumulate()
            S0, W0, occ0 = orc.pair_sums()
            assert np.array_equal(S, S0) and np.array_equal(W, W0) and np.array_equal(occ4, occ0), (case_no, kw)
            assert np.array_equal(h.observe()[2], orc.field_sites()[2]), (case_no, kw)
            for chunk in range(6):
                h.step(10)
                orc.run(10)
                p, sg, bd, al = h.get_state()
                assert np.array_equal(al, orc.alive) and np.array_equal(p, orc.pos), (case_no, chunk, kw)
                assert np.array_equal(sg, orc.spin) and np.array_equal(bd, orc.bound), (case_no, chunk, kw)
                if chunk == 2:
                    h.resort()
                if chunk in (0, 5):
                    check_lattice(h, orc)
            ex, ex0 = h.exits(), orc.exits()
            assert len(ex) == len(ex0)
            if len(ex0):
                assert np.array_equal(ex, ex0[np.lexsort((ex0[:, 2], ex0[:, 0]))]), (case_no, kw)
        finally:
            h.close()


@pytest.mark.parametrize("case", [dict(L=3000, K=1, sigma=0.01), dict(L=1200, K=2, sigma=0.02, periodic=True, k_exit=0.3, k_on=1.0, k_off=0.5,
                                                                     anchor_positions=[0.4], anchor_radius=0.1),
                                  dict(L=900, K=1, sigma=0.0)], ids=["reflect", "periodic_exits", "global_field"])
def test_lattice_graph_replay_long_run(capi, case):
    """Hundreds of steps in one aps_step call go through the captured-graph path (step index in device memory);
    state, exit log and the maintained field must still equal the oracle's step-by-step recomputation."""
    par = params(**case)
    rng = np.random.default_rng(77)
    N = int(0.5 * par.L * par.K)
    pos, spin = random_state(rng, par.L, N, par.K)
    orc = so.SyncOracle(par, dt=0.03, seed=99)
    orc.set_state(pos, spin)
    h = make_handle(capi, par, N, dt=0.03, seed=99, method="lattice")
    try:
        h.set_state(pos, spin)
        h.step(3)                               # odd start: one single step precedes the first replay
        h.step(300)
        h.step(41)
        orc.run(344)
        p, sg, bd, al = h.get_state()
        assert np.array_equal(al, orc.alive) and np.array_equal(p, orc.pos)
        assert np.array_equal(sg, orc.spin) and np.array_equal(bd, orc.bound)
        assert h.time()[1] == 344
        ex, ex0 = h.exits(), orc.exits()
        assert len(ex) == len(ex0)
        if len(ex0):
            assert np.array_equal(ex, ex0[np.lexsort((ex0[:, 2], ex0[:, 0]))])
        check_lattice(h, orc)
    finally:
        h.close()


def _merged_state(handles, n):
    """Global state from site-sharded handles: every particle is reported (alive != 2) by exactly one rank."""
    pos, spin, bound, alive = np.zeros(n, np.int32), np.zeros(n, np.int8), np.zeros(n, np.uint8), np.full(n, 255, np.uint8)
    seen = np.zeros(n, int)
    for h in handles:
        p, s, b, a = h.get_state()
        mine = a != 2
        pos[mine], spin[mine], bound[mine], alive[mine] = p[mine], s[mine], b[mine], a[mine]
        seen += mine
    return pos, spin, bound, alive, seen


@pytest.mark.parametrize("world,periodic,interval,sigma", [(2, False, 1, 0.002), (3, False, 0, 0.002), (2, True, 3, 0.002), (4, True, 0, 0.002),
                                                           (3, False, 2, 0.02), (2, True, 2, 0.02), (4, False, 5, 0.002)])
def test_site_sharded_tiles_equal_single_handle(capi, world, periodic, interval, sigma):
    """Site-range sharding of the tiles formulation, emulated with `world` handles on ONE device: every rank steps its own
    tiles (aps_propose), the halo (3 sites of cells, 2 of {W, S}, the deposit lists within reach) is copied from the
    neighbour handles (aps_halo_copy = what ncclSend / ncclRecv move between GPUs), aps_commit.  With halo interval k the
    exchange happens every k-th step only and carries (k - 1) * reach whole ghost tiles that the receiver steps itself
    in between (interval 0: the library's choice).  The merged state equals the single-handle run and the oracle after
    every block of steps -- blocks of 20 steps, so observations also fall between two exchanges --; every particle is
    owned by exactly one rank; W, S on the own sites equal the single handle's; exits add up."""
    par = params(L=6000, K=2, sigma=sigma, periodic=periodic, anchor_positions=[0.25, 0.5, 0.75], anchor_radius=0.02,
                 k_on=2.0, k_off=1.0, k_exit=0.7, rate_diffusion=3.0)
    rng = np.random.default_rng(77)
    N = 5200
    pos, spin = random_state(rng, par.L, N, par.K)
    orc = so.SyncOracle(par, dt=0.04, seed=31)
    orc.set_state(pos, spin)
    ranks = [make_handle(capi, par, N, dt=0.04, seed=31, rank=r, world=world, method="tiles", halo_interval=interval) for r in range(world)]
    single = make_handle(capi, par, N, dt=0.04, seed=31, method="tiles")
    try:
        k = ranks[0].halo_info()[0]
        assert k == interval if interval else k >= 1
        assert all(h.halo_info() == (k, 0, k == 1) for h in ranks) and single.halo_info() == (0, 0, False)
        exchanges = 0
        bounds = [h.owned_sites() for h in ranks]
        assert bounds[0][0] == 0 and bounds[-1][1] == par.L and all(a[1] == b[0] for a, b in zip(bounds[:-1], bounds[1:]))
        with pytest.raises(capi.ApsError):
            ranks[0].step(1)                                  # aps_step needs a communicator on sharded handles
        for h in ranks + [single]:
            h.set_state(pos, spin)
        for block in range(6):
            for _ in range(20):
                for h in ranks:
                    h.propose()
                due = ranks[0].halo_info()[2]
                assert all(h.halo_info()[2] == due for h in ranks)
                if not due:
                    with pytest.raises(capi.ApsError):
                        ranks[0].halo_from(ranks[1])          # nothing to exchange at this step
                else:
                    if exchanges == 0 and world > 1:
                        with pytest.raises(capi.ApsError):
                            ranks[0].commit()                 # the blocks of a due exchange have not arrived
                    exchanges += 1
                for r, h in enumerate(ranks):
                    nb = {(r - 1) % world, (r + 1) % world} if periodic else {q for q in (r - 1, r + 1) if 0 <= q < world}
                    for q in sorted(nb) if due else []:
                        h.halo_from(ranks[q])
                for h in ranks:
                    h.commit()
            single.step(20)
            orc.run(20)
            p, s, b, a, seen = _merged_state(ranks, N)
            assert np.array_equal(seen, np.ones(N, int)), block
            for x, y in zip((p, s, b, a), single.get_state()):
                assert np.array_equal(x, y), block
            assert np.array_equal(p, orc.pos) and np.array_equal(s, orc.spin) and np.array_equal(a, orc.alive)
        Ws, Ss, occs = single.get_lattice()
        for h, (lo, hi) in zip(ranks, bounds):
            W, S, occ = h.get_lattice()
            assert np.array_equal(W[lo:hi], Ws[lo:hi]) and np.array_equal(S[lo:hi], Ss[lo:hi]) and np.array_equal(occ[lo:hi], occs[lo:hi])
        ex = np.concatenate([h.exits() for h in ranks])
        ex = ex[np.lexsort((ex[:, 2], ex[:, 0]))]
        assert len(ex) > 0 and np.array_equal(ex, single.exits())
        assert exchanges == 120 // k
    finally:
        for h in ranks + [single]:
            h.close()


def test_config3_workload_as_eight_site_ranges(capi):
    """BASELINE config 3's own workload -- N = 1e5 particles on L = 2e5 sites (config 2's system) split 8 ways by site range --
    on ONE device: eight site-sharded handles with the library's halo interval (aps_propose / aps_halo_copy / aps_commit driven
    from here) against the single handle, 60 steps: merged state, {W, S, occupancy} on the own sites, ownership (the single
    handle equals the oracle at this size: test_full_baseline_size_*, test_resident_loop_equals_one_launch_per_step_at_config2).
    The same workload through aps_step with the peer-store transport, one PROCESS per range:
    tests/test_gpu_multiprocess.py::test_config3_workload_across_processes_by_peer_stores."""
    L, N, world, nsteps = 200000, 100000, 8, 60
    par = params(L=L, K=1, sigma=0.005, rate_diffusion=0.02, rate_active=5.0, beta=0.7)
    rng = np.random.default_rng(3)
    pos = rng.choice(L, size=N, replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    dt, seed = 0.0125, 99
    ranks = [make_handle(capi, par, N, dt=dt, seed=seed, rank=r, world=world, method="tiles") for r in range(world)]
    single = make_handle(capi, par, N, dt=dt, seed=seed, method="tiles")
    try:
        k = ranks[0].halo_info()[0]
        assert k >= 2, k                                     # 64 tiles per rank, reach 11: the library keeps a ghost zone
        for h in ranks + [single]:
            h.set_state(pos, spin)
        for _ in range(nsteps):
            for h in ranks:
                h.propose()
            if ranks[0].halo_info()[2]:
                for r, h in enumerate(ranks):
                    for q in (r - 1, r + 1):
                        if 0 <= q < world:
                            h.halo_from(ranks[q])
            for h in ranks:
                h.commit()
        single.step(nsteps)
        p, s, b, a, seen = _merged_state(ranks, N)
        assert np.array_equal(seen, np.ones(N, int))
        for x, y in zip((p, s, b, a), single.get_state()):
            assert np.array_equal(x, y)
        assert not np.array_equal(p, pos)
        Ws, Ss, occs = single.get_lattice()
        for h in ranks:
            lo, hi = h.owned_sites()
            W, S, occ = h.get_lattice()
            assert np.array_equal(W[lo:hi], Ws[lo:hi]) and np.array_equal(S[lo:hi], Ss[lo:hi]) and np.array_equal(occ[lo:hi], occs[lo:hi])
    finally:
        for h in ranks + [single]:
            h.close()


def _oracle_window_field(table, pos, spin, L, a, b):
    """W, S on the sites [a, b) by the oracle's lattice formula (histogram -> stencil with the reflected images, the
    arithmetic of oracle/sync_oracle.c:field_at) restricted to a window.  Every weight sits on the grid 2^-q, so these
    NumPy sums are exact whatever their order."""
    Rt = len(table) - 1
    cp = np.bincount(pos[spin > 0], minlength=L).astype(np.float64)
    cm = np.bincount(pos[spin < 0], minlength=L).astype(np.float64)
    tot, sgn = cp + cm, cp - cm
    lo, hi = a - Rt, b + Rt                                   # source sites that can reach the window; beyond a wall: the images
    idx = np.arange(lo, hi)
    refl = np.where(idx < 0, -1 - idx, np.where(idx >= L, 2 * L - 1 - idx, idx))   # site -1 - y mirrors y, 2L - 1 - y mirrors y
    wsym = np.concatenate([table[::-1], table[1:]])
    W = np.convolve(tot[refl], wsym, mode="valid")
    S = np.convolve(sgn[refl], wsym, mode="valid")
    assert len(W) == b - a
    return W, S


@pytest.mark.parametrize("method", ["lattice", "tiles", "tiles_sweep"])
def test_config5_scale_field_against_oracle_windows(capi, method):
    """BASELINE config 5 scale (N = 1e6, L = 2e6, 40 001-entry table beyond LDS) against the ORACLE, where the oracle can go:
    after 200 steps the incrementally kept W, S are compared bit for bit with the oracle's stencil formula on four windows
    -- at the left wall (image deposits), where the tiles change from the wall path to the windowed path (x ~ reach), deep
    in the interior, and at the right wall -- with the oracle's own weight table (equal to the library's)."""
    L, N = 2_000_000, 1_000_000
    par = LatticeGasParams.from_kwargs(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7,
                                       scale_rates=False, local_kernel_sigma=0.005, site_capacity=1)
    rng = np.random.default_rng(12)
    pos = rng.choice(L, size=N, replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    # tiles: the binary64 field by the exact convolution modulo two primes (the default at this size); tiles_sweep: the windowed sweep
    sweep, method = method == "tiles_sweep", method.split("_")[0]
    if sweep:
        os.environ["APS_NTT"] = "0"
    try:
        h = make_handle(capi, par, N, dt=0.0125, seed=6, method=method)
    finally:
        os.environ.pop("APS_NTT", None)
    try:
        assert h.method == method
        assert h.ntt_info()["on"] == (method == "tiles" and not sweep)
        tab, q = h.table()
        otab, oq = so.build_table(par.sigma_grid, L, 1, False)
        assert q == oq and np.array_equal(tab, otab[:len(tab)]) and len(tab) == 40001
        h.set_state(pos, spin)
        h.step(200)
        p, s, _, alive = h.get_state()
        assert alive.all() and (p != pos).mean() > 0.25 and np.bincount(p, minlength=L).max() <= 1
        W, S, occ = h.get_lattice(0)
        assert np.array_equal(occ, np.bincount(p, minlength=L))
        Rt = len(tab) - 1
        for a0, b0 in ((0, 700), (Rt - 900, Rt + 900), (L // 2 - 300, L // 2 + 500), (L - Rt - 700, L - Rt + 700), (L - 700, L)):
            W0, S0 = _oracle_window_field(tab, p.astype(np.int64), s, L, a0, b0)
            assert np.array_equal(W[a0:b0], W0), (method, a0)
            assert np.array_equal(S[a0:b0], S0), (method, a0)
    finally:
        h.close()


def test_config4_full_size_ensembles(capi):
    """BASELINE config 4 at its full size: 16 beta-ensembles x N = 5e4 on L = 1e5 in ONE handle.  After 150 steps: exclusion
    and conservation in every ensemble; the first and the last ensemble equal single-ensemble handles with the same
    Philox stream (ensemble_base) bit for bit; the incrementally kept W, S, occupancy of those two equal the ORACLE's
    recomputation from their final states; the beta = 0 and beta = 3 ensembles have drifted apart (ordering)."""
    L, N, E = 100_000, 50_000, 16
    betas = [3.0 * i / 15 for i in range(E)]
    par = LatticeGasParams.from_kwargs(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7,
                                       scale_rates=False, local_kernel_sigma=0.005, site_capacity=1)
    rng = np.random.default_rng(44)
    states = [(rng.choice(L, size=N, replace=False).astype(np.int32), rng.choice(np.array([1, -1], np.int8), size=N)) for _ in range(E)]
    big = make_handle(capi, par, N, dt=0.0125, seed=9, beta=betas)
    try:
        assert big.method == "tiles"
        for e, (p0, s0) in enumerate(states):
            big.set_state(p0, s0, ensemble=e)
        big.step(150)
        finals = [big.get_state(ensemble=e) for e in range(E)]
        for e, (p, s, b, al) in enumerate(finals):
            assert al.all() and np.bincount(p, minlength=L).max() <= 1 and np.abs(p.astype(int) - states[e][0]).max() <= 150
        assert abs(float(finals[15][1].mean())) > abs(float(finals[0][1].mean())) + 0.05     # beta = 3 orders, beta = 0 does not
        for e in (0, E - 1):
            one = make_handle(capi, par, N, dt=0.0125, seed=9, beta=[betas[e]], ensemble_base=e)
            try:
                one.set_state(*states[e])
                one.step(150)
                for x, y in zip(finals[e], one.get_state()):
                    assert np.array_equal(x, y), e
            finally:
                one.close()
            orc_par = LatticeGasParams.from_kwargs(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=betas[e],
                                                   scale_rates=False, local_kernel_sigma=0.005, site_capacity=1)
            orc = so.SyncOracle(orc_par, dt=0.0125, seed=9, ensemble=e)
            orc.set_state(finals[e][0], finals[e][1])
            check_lattice(big, orc, ensemble=e)
    finally:
        big.close()


FP32_CASES = [
    dict(tag="k1_reflect", L=3000, K=1, sigma=0.01, frac=0.5),
    dict(tag="k1_periodic", L=1200, K=1, sigma=0.02, periodic=True, frac=0.5),
    dict(tag="k3_small_box", L=200, K=3, sigma=0.3, frac=0.6),
    dict(tag="k2_anchors_exit", L=900, K=2, sigma=0.02, frac=0.5, anchor_positions=[0.3, 0.7], anchor_radius=0.05, k_on=3.0, k_off=1.0, k_exit=2.0),
    dict(tag="table_beyond_lds", L=60000, K=1, sigma=0.1, frac=0.05),
]


@pytest.mark.parametrize("case", FP32_CASES, ids=lambda c: c["tag"])
def test_fp32_field_is_exact_on_its_own_grid(capi, case):
    """aps_params.fp32 (BASELINE config 5 asks for float32): weights, W and S are int32 in units of 2^-q with q = 29 - bits
    of the largest possible sum.  Still integer arithmetic, so the run must equal -- bit for bit, state after every block and
    the field arrays -- the oracle stepped with the SAME coarse table (oracle/sync_oracle.c with sum_bits = 29), and the field
    must sit within float32-class distance of the exact (2^-q, q ~ 38) one."""
    case = dict(case)
    tag, frac = case.pop("tag"), case.pop("frac")
    par = params(**case)
    rng = np.random.default_rng(17)
    N = max(1, int(frac * par.L * par.K))
    pos, spin = random_state(rng, par.L, N, par.K)
    orc = so.SyncOracle(par, dt=0.04, seed=5, sum_bits=29)
    fine = so.SyncOracle(par, dt=0.04, seed=5)
    orc.set_state(pos, spin)
    h = make_handle(capi, par, N, dt=0.04, seed=5, method="tiles", fp32=True)
    try:
        tab, q = h.table()
        assert q == orc.q and q <= 29 and np.array_equal(tab, orc.table) and np.array_equal(tab * 2.0 ** q, np.rint(tab * 2.0 ** q))
        h.set_state(pos, spin)
        check_lattice(h, orc)
        for block in range(4):
            h.step(25)
            orc.run(25)
            p, sg, bd, al = h.get_state()
            assert np.array_equal(p, orc.pos) and np.array_equal(sg, orc.spin) and np.array_equal(bd, orc.bound) and np.array_equal(al, orc.alive), (tag, block)
            check_lattice(h, orc)
        # distance to the exact field: the same state under the fine table
        fine.set_state(p, sg, bound=bd, alive=al)
        _, _, m_fine = fine.field_sites()
        _, _, m_coarse = orc.field_sites()
        assert np.max(np.abs(m_fine - m_coarse)) < 2e-4, tag       # float32-class, far above the 2e-11 of the exact grid
    finally:
        h.close()


@pytest.mark.parametrize("periodic", [False, True], ids=["walls", "torus"])
def test_fp32_windowed_sweep_dense_buckets(capi, periodic):
    """32-bit field with a table beyond LDS even at 4 bytes per entry (52 718 entries): interior tiles sweep double-buffered
    windows of the table with directional gathers (tile_step.hpp: ts_group_dir), tiles near a wall and every tile of a torus
    gather from the table in global memory.  Two dense clusters (K = 3, ~300 events per tile and step: several list pages per
    bucket, the one-by-one path for buckets with more than 16 deposits) and a thin background, so the oracle's O(N taps) step
    stays affordable.  State after every block and {W, S, occupancy} on ALL sites against the oracle with the same coarse
    table, bit for bit.  (FP32_CASES' own "table_beyond_lds" fits LDS at 4 bytes per entry; config 5's windows are otherwise
    only met at N = 1e6, where the oracle can check windows of sites but no trajectory.)"""
    par = params(L=90000, K=3, sigma=0.15, rate_diffusion=3.0, periodic=periodic)
    rng = np.random.default_rng(1)
    L = par.L
    sites = np.concatenate([rng.integers(20000, 21500, 2500), rng.integers(60000, 61000, 2000), rng.integers(0, L, 1000), np.arange(L - 40, L), np.arange(0, 30)])
    u, c = np.unique(sites, return_counts=True)
    pos = rng.permutation(np.concatenate([np.repeat(x, min(k, 3)) for x, k in zip(u, c)])).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=len(pos))
    N = len(pos)
    orc = so.SyncOracle(par, dt=0.05, seed=5, sum_bits=29)
    orc.set_state(pos, spin)
    os.environ["APS_NTT"] = "0"                              # this test is about the sweep (between walls the convolution would take over)
    try:
        h = make_handle(capi, par, N, dt=0.05, seed=5, method="tiles", fp32=True)
    finally:
        del os.environ["APS_NTT"]
    try:
        assert not h.ntt_info()["on"]
        info = h.tiles_info()
        assert not info["table_in_lds"], info
        tab, q = h.table()
        assert q == orc.q and np.array_equal(tab, orc.table)
        h.set_state(pos, spin)
        for block, n in enumerate((1, 6, 5)):
            h.step(n)
            orc.run(n)
            p, sg, bd, al = h.get_state()
            assert np.array_equal(p, orc.pos) and np.array_equal(sg, orc.spin), block
            if block != 1:
                check_lattice(h, orc)
        assert (p != pos).mean() > 0.4
    finally:
        h.close()


@pytest.mark.parametrize("fp32", [True, False], ids=["i32", "f64"])
@pytest.mark.parametrize("sigma,L,fused", [(0.02, 90000, "1"), (0.02, 90000, "0"), (0.004, 16000, "1"), (0.1, 120000, "1")],
                         ids=["m17", "m17_five_launches", "m14_two_sweeps", "m18_beyond_lds"])
def test_fp32_field_update_by_exact_convolution(capi, sigma, L, fused, fp32):
    """csrc/ntt_conv.hpp: the step's deposits -> W, S of all sites by ONE number-theoretic convolution (mod 15 * 2^27 + 1, length
    2^m >= L + 2 reach, wall images entered as mirrored deposits) instead of the sweep -- exact integers, so state after every
    block and {W, S, occupancy} on ALL sites must equal the oracle's (which knows nothing of transforms) bit for bit.  Forced by
    APS_NTT=1 for tables that fit LDS (the first two cases: transforms of three and of two sweeps); the third is beyond LDS and
    takes the convolution by itself.  From m = 15 on the three middle launches are one (ntt_mid: a 128 x 128 slab per workgroup);
    APS_NTT_FUSED=0 keeps the five launches, which m = 14 always takes.  Dense clusters at both walls and inside, K = 3, a thin
    background.  f64: the binary64 field (integers below 2^51 in units of its 2^-q) by the same transform modulo two primes, put
    together by the last sweep (Chinese remainder) -- from m = 15 on (the binary64 table of the second case is longer: m = 15)."""
    par = params(L=L, K=3, sigma=sigma, rate_diffusion=3.0)
    rng = np.random.default_rng(2)
    sites = np.concatenate([rng.integers(L // 4, L // 4 + 1500, 2500), rng.integers(0, L, 1500), np.arange(L - 300, L), np.arange(0, 200)])
    u, c = np.unique(sites, return_counts=True)
    pos = rng.permutation(np.concatenate([np.repeat(x, min(k, 3)) for x, k in zip(u, c)])).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=len(pos))
    N = len(pos)
    orc = so.SyncOracle(par, dt=0.05, seed=7, **(dict(sum_bits=29) if fp32 else {}))
    orc.set_state(pos, spin)
    os.environ["APS_NTT"] = "1"
    os.environ["APS_NTT_FUSED"] = fused
    try:
        h = make_handle(capi, par, N, dt=0.05, seed=7, method="tiles", fp32=fp32)
    finally:
        del os.environ["APS_NTT"], os.environ["APS_NTT_FUSED"]
    try:
        info = h.ntt_info()
        assert info["on"] and (1 << info["log2_m"]) >= L + 2 * (len(h.table()[0]) - 1), info
        assert info["launches"] == (3 if fused == "1" and info["log2_m"] >= 15 else 5 if info["log2_m"] >= 15 else 3), info
        h.set_state(pos, spin)
        check_lattice(h, orc)
        for block, n in enumerate((1, 2, 37)):               # single steps and graph replay
            h.step(n)
            orc.run(n)
            p, sg, bd, al = h.get_state()
            assert np.array_equal(p, orc.pos) and np.array_equal(sg, orc.spin), block
            check_lattice(h, orc)
        assert (p != pos).mean() > 0.4
    finally:
        h.close()


def test_fp32_convolution_with_ensembles_anchors_and_exits(capi):
    """The convolution path (forced, APS_NTT=1) on a handle of three ensembles with anchors, bind / unbind and exits (an exit is a
    deposit like any other: -1 in c_W, -sigma in c_S), K = 2: every ensemble against its own oracle after blocks of steps -- state,
    exit log, {W, S, occupancy} on all sites."""
    betas = [0.4, 1.3, 2.2]
    kw = dict(L=20000, K=2, sigma=0.01, rate_diffusion=2.0, anchor_positions=[0.3, 0.7], anchor_radius=0.02, k_on=2.0, k_off=1.0, k_exit=0.8)
    par0 = params(**kw)
    rng = np.random.default_rng(9)
    N = 15000
    states = [random_state(rng, par0.L, N, par0.K) for _ in betas]
    os.environ["APS_NTT"] = "1"
    try:
        h = make_handle(capi, par0, N, dt=0.04, seed=12, method="tiles", fp32=True, beta=betas)
    finally:
        del os.environ["APS_NTT"]
    orcs = []
    for e, b in enumerate(betas):
        orc = so.SyncOracle(params(beta=b, **kw), dt=0.04, seed=12, ensemble=e, sum_bits=29)
        orc.set_state(*states[e])
        orcs.append(orc)
    try:
        assert h.ntt_info()["on"]
        for e, (p, s) in enumerate(states):
            h.set_state(p, s, ensemble=e)
        for n in (3, 30):
            h.step(n)
            for e, orc in enumerate(orcs):
                orc.run(n)
                got = h.get_state(ensemble=e)
                assert np.array_equal(got[0], orc.pos) and np.array_equal(got[1], orc.spin) and np.array_equal(got[2], orc.bound) and np.array_equal(got[3], orc.alive), (n, e)
                check_lattice(h, orc, ensemble=e)
        assert any((orc.alive == 0).any() for orc in orcs)
        for e, orc in enumerate(orcs):
            assert np.array_equal(h.exits(ensemble=e), orc.exits())
    finally:
        h.close()


@pytest.mark.parametrize("fp32", [True, False], ids=["i32", "f64"])
@pytest.mark.parametrize("L,sigma", [(70000, 0.05), (65537, 0.3)], ids=["even_short_table", "odd_ring_wide_table"])
def test_convolution_on_a_torus(capi, L, sigma, fp32):
    """The convolution path with periodic boundaries, where the reference itself multiplies FFTs (PARTICLE_solver_CLASS.py:223-227):
    a deposit within the table's reach of either end of [0, L) is entered a second time one period on (s + L, s - L), the tap at
    distance L / 2 of an even ring counts once.  A table that ends before half the ring and one that spans it (odd L: every distance
    up to (L - 1) / 2), K = 2, clusters across the seam; state and {W, S, occupancy} on all sites against the oracle, bit for bit."""
    par = params(L=L, K=2, sigma=sigma, periodic=True, rate_diffusion=3.0)
    rng = np.random.default_rng(4)
    sites = np.concatenate([rng.integers(0, L, 3000), np.arange(L - 400, L), np.arange(0, 300), rng.integers(L // 2 - 500, L // 2 + 500, 1200)])
    u, c = np.unique(sites, return_counts=True)
    pos = rng.permutation(np.concatenate([np.repeat(x, min(k, 2)) for x, k in zip(u, c)])).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=len(pos))
    N = len(pos)
    orc = so.SyncOracle(par, dt=0.05, seed=17, **(dict(sum_bits=29) if fp32 else {}))
    orc.set_state(pos, spin)
    os.environ["APS_NTT"] = "1"
    try:
        h = make_handle(capi, par, N, dt=0.05, seed=17, method="tiles", fp32=fp32)
    finally:
        del os.environ["APS_NTT"]
    try:
        info = h.ntt_info()
        assert info["on"] and info["launches"] == 3, info
        if sigma > 0.2:
            assert len(h.table()[0]) - 1 == L // 2                # ring-wide
        h.set_state(pos, spin)
        check_lattice(h, orc)
        for block, n in enumerate((1, 2, 21)):
            h.step(n)
            orc.run(n)
            p, sg, bd, al = h.get_state()
            assert np.array_equal(p, orc.pos) and np.array_equal(sg, orc.spin), block
            check_lattice(h, orc)
        assert (p != pos).mean() > 0.4
    finally:
        h.close()


def test_fp32_convolution_with_six_cells_per_site(capi):
    """csrc/tile_dense.hpp with K = 6: a frame's cells, particle list and proposals (80 KB) need the enlarged LDS limit; the general
    exclusion rule (rank among the proposers of a site against its free capacity) on crowded sites, bit for bit against the oracle."""
    par = params(L=30000, K=6, sigma=0.01, rate_diffusion=4.0)
    rng = np.random.default_rng(21)
    N = 60000
    pos, spin = random_state(rng, par.L, N, par.K)
    orc = so.SyncOracle(par, dt=0.04, seed=3, sum_bits=29)
    orc.set_state(pos, spin)
    os.environ["APS_NTT"] = "1"
    try:
        h = make_handle(capi, par, N, dt=0.04, seed=3, method="tiles", fp32=True)
    finally:
        del os.environ["APS_NTT"]
    try:
        assert h.ntt_info()["on"] and h.ntt_info()["launches"] == 3
        h.set_state(pos, spin)
        for n in (1, 12):
            h.step(n)
            orc.run(n)
            p, sg, bd, al = h.get_state()
            assert np.array_equal(p, orc.pos) and np.array_equal(sg, orc.spin), n
            check_lattice(h, orc)
        assert (p != pos).mean() > 0.3
    finally:
        h.close()


@pytest.mark.parametrize("update", ["convolution", "sweep"])
def test_fp32_config5_scale_against_oracle_windows(capi, update):
    """BASELINE config 5 as it is worded (N = 1e6, float32): the int32 field after 200 steps against the oracle's stencil
    with the coarse table on wall / transition / interior windows, bit for bit -- the field kept by the exact convolution
    (csrc/ntt_conv.hpp, the default at this size) and by the windowed sweep (APS_NTT=0)."""
    L, N = 2_000_000, 1_000_000
    par = LatticeGasParams.from_kwargs(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7,
                                       scale_rates=False, local_kernel_sigma=0.005, site_capacity=1)
    rng = np.random.default_rng(12)
    pos = rng.choice(L, size=N, replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=N)
    os.environ["APS_NTT"] = "1" if update == "convolution" else "0"
    try:
        h = make_handle(capi, par, N, dt=0.0125, seed=6, method="tiles", fp32=True)
    finally:
        del os.environ["APS_NTT"]
    try:
        assert h.ntt_info()["on"] == (update == "convolution")
        tab, q = h.table()
        otab, oq = so.build_table(par.sigma_grid, L, 1, False, 29)
        assert q == oq == 12 and np.array_equal(tab, otab[:len(tab)]) and len(tab) == 40001
        h.set_state(pos, spin)
        h.step(200)
        p, s, _, alive = h.get_state()
        assert alive.all() and (p != pos).mean() > 0.25 and np.bincount(p, minlength=L).max() <= 1
        W, S, occ = h.get_lattice(0)
        Rt = len(tab) - 1
        for a0, b0 in ((0, 700), (Rt - 900, Rt + 900), (L // 2 - 300, L // 2 + 500), (L - Rt - 700, L - Rt + 700), (L - 700, L)):
            W0, S0 = _oracle_window_field(tab, p.astype(np.int64), s, L, a0, b0)
            assert np.array_equal(W[a0:b0], W0) and np.array_equal(S[a0:b0], S0), a0
    finally:
        h.close()
