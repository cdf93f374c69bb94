import importlib, sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_parity import make_handle, params, random_state
capi = importlib.import_module("hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd.capi")
L = 180
par = params(L=L, K=1, sigma=0.05, beta=0.0)
rng = np.random.default_rng(11)
N = L // 2
pos, spin = random_state(rng, par.L, N, par.K)
n = 3
os.environ["APS_LOOP_DUMP"] = "/tmp/loop_dump.bin"
h = make_handle(capi, par, N, dt=0.04, seed=20260202, method="tiles")
h.set_state(pos, spin); h.step(n); print(h.loop_info())
pa, sa, _, _ = h.get_state()
h.close()
d = np.fromfile("/tmp/loop_dump.bin", dtype=np.uint32).reshape(n, 3, L)
def cells_of(m):
    b = make_handle(capi, par, N, dt=0.04, seed=20260202, method="tiles"); b.set_resident_loop(False)
    b.set_state(pos, spin)
    if m: b.step(m)
    p, s, _, _ = b.get_state(); b.close()
    c = np.full(L, -1, np.int64); c[p] = np.arange(N)          # particle id per site (K = 1)
    sp = np.zeros(L, np.int64); sp[p] = s
    return c, sp
for it in range(n):
    cb, sb = cells_of(it + 1)
    cells = d[it, 0]; ids = np.where(cells == 0xFFFFFFFF, -1, (cells & 0x3FFFFFFF).astype(np.int64))
    c0, s0 = cells_of(it)
    seen = d[it, 2]; ids_seen = np.where(seen == 0xFFFFFFFF, -1, (seen & 0x3FFFFFFF).astype(np.int64))
    occ = d[it, 1] & 0xFF; prop = (d[it, 1] >> 8) & 0xFF; ol = (d[it, 1] >> 16) & 0xFF; orr = d[it, 1] >> 24
    occ_true = (c0 >= 0).astype(int)
    print("it", it, "cells-in wrong on", np.nonzero(ids_seen != c0)[0][:20], "occ wrong on", np.nonzero(occ != occ_true)[0][:20],
          "occ-left wrong", np.nonzero(ol[1:] != occ_true[:-1])[0][:10] + 1, "occ-right wrong", np.nonzero(orr[:-1] != occ_true[1:])[0][:10])
    bad = np.nonzero(ids != cb)[0]
    print("      cells-out wrong on", bad[:30], "events:", [(int(x), int(prop[x]) & 7) for x in np.nonzero((prop & 7) != 0)[0]][:40])

cb, sb = cells_of(n)
ca = np.full(L, -1, np.int64); ca[pa] = np.arange(N)
print("get_state(A) ids vs B:", np.nonzero(ca != cb)[0][:20], " dump(last) vs get_state(A):", np.nonzero(ids != ca)[0][:20])
spa = np.zeros(L, np.int64); spa[pa] = sa
dump_spin = np.where(cells == 0xFFFFFFFF, 0, np.where(cells >> 31, 1, -1))
print("spin: get_state(A) vs B", np.nonzero(spa != sb)[0][:20], "dump vs B", np.nonzero(dump_spin != sb)[0][:20])
