"""Diagnostic: config 2's parameters on a torus (ring-wide table: L / 2 + 1 entries, beyond LDS) -- the convolution against the sweep that gathers from global memory."""
import os, sys, time, importlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
capi = importlib.import_module(bench.PKG + ".capi")
w = dict(bench.WORK)
for fp32 in (False, True):
    for ntt in ("1", "0"):
        os.environ["APS_NTT"] = ntt
        h = capi.Handle(L=w["L"], K=1, periodic=True, sigma_grid=w["sigma"] * w["L"], rate_diffusion=w["rate_diffusion"], rate_active=w["rate_active"],
                        beta=[w["beta"]], dt=w["dt"], seed=0, n_particles=w["N"], fp32=fp32)
        del os.environ["APS_NTT"]
        h.set_state(*bench.initial_state(w))
        h.step(8)
        n = 64 if ntt == "1" else 8
        t0 = time.perf_counter(); h.step(n); dt = (time.perf_counter() - t0) / n * 1e6
        print("torus L=%d N=%d %s convolution=%s table %d entries: %.1f us per step" % (w["L"], w["N"], "i32" if fp32 else "f64", h.ntt_info()["on"], len(h.table()[0]), dt), flush=True)
        h.close()
