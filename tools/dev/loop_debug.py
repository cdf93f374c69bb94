"""Development aid: where does the resident loop first differ from one launch per step?"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_parity import make_handle, params, random_state
capi = importlib.import_module("hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd.capi")
case = dict(L=int(os.environ.get("DBG_L", 3000)), K=1, sigma=float(os.environ.get("DBG_SIGMA", 0.01)), beta=float(os.environ.get("DBG_BETA", 1.1)))
par = params(**case)
rng = np.random.default_rng(11)
N = par.L // 2
pos, spin = random_state(rng, par.L, N, par.K)
for n in [int(x) for x in os.environ.get("DBG_N", "3,5,7").split(",")]:
    a = make_handle(capi, par, N, dt=0.04, seed=20260202, method="tiles")
    b = make_handle(capi, par, N, dt=0.04, seed=20260202, method="tiles")
    b.set_resident_loop(False)
    a.set_state(pos, spin); b.set_state(pos, spin)
    a.step(n); b.step(n)
    print("n", n, "loop_info", a.loop_info())
    pa, sa, _, _ = a.get_state(); pb, sb, _, _ = b.get_state()
    bad = np.nonzero((pa != pb) | (sa != sb))[0]
    print("  differing pos", int((pa != pb).sum()), "spin", int((sa != sb).sum()), "pos%60 of differing", np.bincount(pb[pa != pb] % 60, minlength=60))
    print("  differing particles", len(bad), [(int(i), int(pa[i]), int(pb[i]), int(sa[i]), int(sb[i])) for i in bad[:10]])
    Wa, Sa, oa = a.get_lattice(); Wb, Sb, ob = b.get_lattice()
    for name, x, y in (("W", Wa, Wb), ("S", Sa, Sb), ("occ", oa, ob)):
        d = np.nonzero(x != y)[0]
        print("  ", name, "differs on", len(d), "sites", d[:12], "tiles(60)", np.unique(d // 60)[:12])
    a.close(); b.close()
