"""Diagnostic (GPU box): long runs of the two production paths, then the incrementally kept field against a from-scratch rebuild of the
final state (second handle, field_sites kernel) on ALL sites, bit for bit -- config 2 through the resident loop, config 5 (32-bit and
binary64 field) through tile_dense + the exact convolution.  Usage: python tools/dev/soak.py [steps_config2] [steps_config5]"""
import os, sys, time, importlib
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
capi = importlib.import_module(bench.PKG + ".capi")
n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
n5 = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000

def run(tag, w, steps, chunk, fp32):
    mk = lambda: capi.Handle(L=w["L"], K=1, periodic=False, sigma_grid=w["sigma"] * w["L"], rate_diffusion=w["rate_diffusion"], rate_active=w["rate_active"],
                             beta=[w["beta"]], dt=w["dt"], seed=0, n_particles=w["N"], fp32=fp32)
    a = mk()
    pos, spin = bench.initial_state(w)
    a.set_state(pos, spin)
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        a.step(min(chunk, steps - done)); done += min(chunk, steps - done)
    dt = time.perf_counter() - t0
    p, s, b, alive = a.get_state()
    W, S, occ = a.get_lattice(0)
    assert alive.all() and occ.max() <= 1 and int(occ.sum()) == w["N"]
    bh = mk()
    bh.set_state(p, s)
    W2, S2, occ2 = bh.get_lattice(0)
    same = np.array_equal(W, W2) and np.array_equal(S, S2) and np.array_equal(occ, occ2)
    print(f"{tag}: {steps} steps in {dt:.2f} s ({dt / steps * 1e6:.2f} us/step), moved {(p != pos).mean():.3f}, loop {a.loop_info()[:2]}, convolution {a.ntt_info()['on']}, "
          f"field == from-scratch rebuild on all {w['L']} sites: {same}", flush=True)
    a.close(); bh.close()
    return same

ok = run("config 2 (f64, resident loop)", dict(bench.WORK), n2, 20_000, False)
ok &= run("config 5 (32-bit field, convolution)", dict(bench.EXTRA["config5"]), n5, 5_000, True)
ok &= run("config 5 (binary64 field, two primes)", dict(bench.EXTRA["config5"]), n5, 5_000, False)
sys.exit(0 if ok else 1)
