"""Diagnostic (GPU box): random geometries through the convolution path (APS_NTT=1: tile_dense + ntt_conv.hpp, one or two primes, walls or
torus, ensembles, anchors / exits) against the CPU oracle, bit for bit -- a differential run over more parameter combinations than the
suite holds.  Usage: python tools/dev/fuzz_convolution.py [cases] [seed]"""
import os, sys, importlib
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sync_oracle as so
from test_gpu_parity import check_lattice, make_handle, params, random_state
capi = importlib.import_module("hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd.capi")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(ncase):
    periodic = bool(rng.integers(0, 2))
    fp32 = bool(rng.integers(0, 2))
    K = int(rng.choice([1, 2, 3, 5]))
    L = int(rng.integers(9000, 40000))                       # (the CPU oracle sets the pace: ~1 s per step and ensemble at L = 4e4 with a wide table)
    sigma = float(rng.choice([0.004, 0.01, 0.03, 0.08, 0.2] if not periodic else [0.004, 0.02, 0.1, 0.4]))
    kw = dict(L=L, K=K, sigma=sigma, periodic=periodic, rate_diffusion=float(rng.choice([0.5, 3.0])))
    if K > 1 and rng.integers(0, 2):
        kw.update(anchor_positions=[0.25, 0.6], anchor_radius=0.03, k_on=3.0, k_off=0.7, k_exit=float(rng.choice([0.0, 1.5])))
    betas = [0.5, 1.7] if rng.integers(0, 3) == 0 else [1.1]
    par0 = params(**kw)
    N = int(rng.uniform(0.05, 0.6) * L * K)
    states = [random_state(rng, L, N, K) for _ in betas]
    dt, seed = 0.04, int(rng.integers(1, 1 << 30))
    os.environ["APS_NTT"] = "1"
    os.environ["APS_NTT_FUSED"] = "1" if rng.integers(0, 4) else "0"
    try:
        h = make_handle(capi, par0, N, dt=dt, seed=seed, method="tiles", fp32=fp32, beta=betas)
    finally:
        fused = os.environ.pop("APS_NTT_FUSED"); del os.environ["APS_NTT"]
    info = h.ntt_info()
    tag = dict(case=case, L=L, K=K, sigma=sigma, periodic=periodic, fp32=fp32, E=len(betas), N=N, exits=kw.get("k_exit", 0.0), fused=fused, on=info["on"], m=info["log2_m"])
    if not info["on"]:
        print("not eligible", tag, flush=True); h.close(); continue
    orcs = []
    for e, b in enumerate(betas):
        orc = so.SyncOracle(params(beta=b, **kw), dt=dt, seed=seed, ensemble=e, **(dict(sum_bits=29) if fp32 else {}))
        orc.set_state(*states[e]); orcs.append(orc)
        h.set_state(*states[e], ensemble=e)
    ok = True
    try:
        for n in (1, 7):
            h.step(n)
            for e, orc in enumerate(orcs):
                orc.run(n)
                got = h.get_state(ensemble=e)
                assert np.array_equal(got[0], orc.pos) and np.array_equal(got[1], orc.spin) and np.array_equal(got[2], orc.bound) and np.array_equal(got[3], orc.alive), ("state", n, e)
                check_lattice(h, orc, ensemble=e)
                assert np.array_equal(h.exits(ensemble=e), orc.exits()), ("exits", n, e)
    except AssertionError as ex:
        ok = False; bad += 1
        print("MISMATCH", tag, ex.args[:1], flush=True)
    if ok: print("ok", tag, flush=True)
    h.close()
print("cases", ncase, "mismatches", bad)
sys.exit(1 if bad else 0)
