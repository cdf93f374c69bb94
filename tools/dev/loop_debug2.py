import importlib, sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_parity import make_handle, params, random_state
capi = importlib.import_module("hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd.capi")
par = params(L=3000, K=1, sigma=0.01, beta=float(os.environ.get("DBG_BETA", 0)))
rng = np.random.default_rng(11)
N = 1500
pos, spin = random_state(rng, par.L, N, par.K)
def run(n, loop):
    h = make_handle(capi, par, N, dt=0.04, seed=20260202, method="tiles")
    h.set_resident_loop(loop)
    h.set_state(pos, spin); h.step(n)
    out = h.get_state()[:2], h.loop_info()
    h.close()
    return out
for n in (3, 5):
    (pa, sa), info = run(n, True)
    print("loop n", n, info)
    for m in range(n - 2, n + 2):
        (pb, sb), _ = run(m, False)
        print("   vs per-step m", m, "differing", int(((pa != pb) | (sa != sb)).sum()))
