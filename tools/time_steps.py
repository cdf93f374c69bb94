import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
capi = importlib.import_module(bench.PKG + ".capi")
if os.environ.get("APS_LIB"):                      # an alternative build of the library (tuning experiments)
    capi.LIB_PATH = os.environ["APS_LIB"]
wl = sys.argv[1]; fp32 = bool(int(sys.argv[2]))
w = dict(bench.WORK) if wl == "config2" else dict(bench.EXTRA[wl])
h = capi.Handle(L=w["L"], K=1, periodic=False, sigma_grid=w["sigma"] * w["L"], rate_diffusion=w["rate_diffusion"], rate_active=w["rate_active"],
                beta=w.get("betas", [w["beta"]]), dt=w["dt"], seed=0, n_particles=w["N"], fp32=fp32)
pos, spin = bench.initial_state(w)
for e in range(len(w.get("betas", [0]))): h.set_state(pos, spin, ensemble=e)
h.step(64)
n = 512 if wl != "config5" else 128
ts = []
for _ in range(3):
    t0 = time.perf_counter(); h.step(n); ts.append((time.perf_counter() - t0) / n * 1e6)
print(wl, "fp32" if fp32 else "f64", "R", os.environ.get("APS_TS_R"), "us/step", [round(t, 2) for t in ts])
h.close()
