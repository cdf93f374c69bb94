import importlib, sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_parity import make_handle, params, random_state
capi = importlib.import_module("hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd.capi")
L = int(os.environ.get("DBG_L", 180))
par = params(L=L, K=1, sigma=0.05, beta=0.0)
rng = np.random.default_rng(11)
N = L // 2
pos, spin = random_state(rng, par.L, N, par.K)
def run(n, loop):
    h = make_handle(capi, par, N, dt=0.04, seed=20260202, method="tiles")
    h.set_resident_loop(loop)
    h.set_state(pos, spin)
    if n: h.step(n)
    out = h.get_state()[:2]
    h.close()
    return out
n = int(os.environ.get("DBG_N", 2))
pa, sa = run(n, True)
B = [run(m, False) for m in range(n + 1)]
bad = np.nonzero((pa != B[n][0]) | (sa != B[n][1]))[0]
print("differing", len(bad))
for i in bad[:40]:
    print(i, "B:", [(int(B[m][0][i]), int(B[m][1][i])) for m in range(n + 1)], "A final:", (int(pa[i]), int(sa[i])))
for m in range(1, n + 1):
    print("B step", m, "changed particles", int(((B[m][0] != B[m-1][0]) | (B[m][1] != B[m-1][1])).sum()))
print("A vs B[1] differing", int(((pa != B[1][0]) | (sa != B[1][1])).sum()), " A vs B[0]", int(((pa != B[0][0]) | (sa != B[0][1])).sum()))
from oracle import sync_oracle as so
orc = so.SyncOracle(par, dt=0.04, seed=20260202); orc.set_state(pos, spin)
for m in range(1, n + 1):
    orc.step()
    print("oracle step", m, "B==oracle", bool(np.array_equal(B[m][0], orc.pos) and np.array_equal(B[m][1], orc.spin)), "A==oracle", bool(np.array_equal(pa, orc.pos) and np.array_equal(sa, orc.spin)))
