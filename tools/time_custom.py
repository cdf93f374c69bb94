"""usage: python tools/time_custom.py L N sigma_g [fp32]  -- us per step of the tiles stepper at a custom size (diagnostic)"""
import sys, time, importlib
sys.path.insert(0, ".")
import numpy as np
import bench
capi = importlib.import_module(bench.PKG + ".capi")
L, N, sg = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
fp32 = len(sys.argv) > 4 and sys.argv[4] == "1"
h = capi.Handle(L=L, K=1, periodic=False, sigma_grid=sg, rate_diffusion=0.02, rate_active=5.0, beta=[0.7], dt=0.0125, seed=0, n_particles=N, fp32=fp32)
rng = np.random.default_rng(0)
pos = rng.choice(L, size=N, replace=False).astype(np.int32)
spin = rng.choice(np.array([1, -1], np.int8), size=N)
h.set_state(pos, spin)
h.step(64)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); h.step(1024); ts.append((time.perf_counter() - t0) / 1024 * 1e6)
prof = h.step_profile(20)
print("L", L, "N", N, "sigma_g", sg, "us/step", [round(t, 2) for t in ts], {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1]})
h.close()
