"""Diagnostic: build the library with -DAPS_STAMPS into /tmp and print per-phase cycles of the exact event loop
(system 0).  Usage (GPU box): python tools/stamps_gillespie.py"""
import importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"
lib = "/tmp/libaps_stamps_gil.so"
src = [os.path.join(ROOT, PKG, "csrc", f) for f in ("aps_hip.hip", "pde_hip.hip", "gillespie_hip.hip", "gillespie_big_hip.hip")]
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DAPS_STAMPS",
                "-I", os.path.join(ROOT, "include"), "-o", lib] + src, check=True)
capi = importlib.import_module(PKG + ".capi")
capi.LIB_PATH = lib
gil = importlib.import_module(PKG + ".gillespie")
L, N = 1000, 500
rng = np.random.default_rng(0)
for nsys in (1, 1024):
    states = [(rng.choice(L, size=N, replace=False).astype(np.int32), rng.choice(np.array([1, -1], np.int8), size=N)) for _ in range(nsys)]
    r = gil.run_raw(L=L, K=1, periodic=False, sigma_grid=5.0, rate_diffusion=0.02, rate_active=5.0, betas=np.full(nsys, 0.7), states=states,
                    times_obs=np.arange(0.0, 20.0, 0.1), T=20.0, seed=1, want_states=False)
    ev = int(r["n_events"][0])
    st = r["exits"][0, :2].ravel()[:5]
    print(f"{nsys} system(s): kernel {r['kernel_ms']:.1f} ms, system 0: {ev} events, {r['kernel_ms'] * 1e3 / max(r['n_events'].max(), 1):.2f} us/event")
    for name, c in zip(("A rates+scan", "B draws+select", "C apply", "D field", "E time/obs"), st):
        print(f"   {name:16s} {c / ev:8.0f} cycles/event")

# the large-system kernel at the BASELINE size
L, N = 200_000, 100_000
pos = np.sort(rng.choice(L, size=N, replace=False)).astype(np.int32)
sg = rng.choice(np.array([1, -1], np.int8), size=N)
r = gil.run_large_raw(L=L, K=1, periodic=False, sigma_grid=0.005 * L, rate_diffusion=0.02, rate_active=5.0, beta=0.7, state=(pos, sg),
                      times_obs=np.array([0.0, 1e9]), T=1e9, seed=1, max_events=5000, want_states=False)
ev = r["n_events"]
print(f"large system N={N}: {ev} events, {r['kernel_ms'] * 1e3 / ev:.1f} us/event")
names = ("A1 work list", "A2 rates+blocks", "B5 in-block choice", "C apply", "D field", "B1 sums+scan", "B2 draws", "B3 thread choice", "B4 block choice")
for name, c in zip(names, r["exits"].ravel()[:9]):
    print(f"   {name:18s} {c / ev:8.0f} cycles/event")
