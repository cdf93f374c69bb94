"""CPU side of SURVEY 8(d)'s throughput table, on the host cores of the GPU box: the reference's loop (NumPy
restatement, oracle/gillespie_numpy.py) in events/s with its m-field / rates split, for N in {2e3, 1e4, 1e5} and the
three kernels.  Test infrastructure timing only.  Usage: python tools/cpu_table.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gillespie_numpy import GillespieOracle

print(f"host cores visible: {os.cpu_count()} (NumPy runs these loops on one)")
print("   N        kernel            events/s   m-field ms   rates+choice ms")
for N, n_ev in ((2000, 300), (10000, 100), (100000, 20)):
    for label, kw in (("reflect s=0.005", dict(local_kernel_sigma=0.005, periodic=False)),
                      ("periodic s=0.005", dict(local_kernel_sigma=0.005, periodic=True)),
                      ("global mean", dict(local_kernel_sigma=0.0, periodic=False))):
        L = 2 * N
        orc = GillespieOracle(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7, N=N, scale_rates=False, site_capacity=1,
                              rng=np.random.default_rng(0), **kw)
        pos, sigma = orc.init_particles()
        bound = np.zeros(N, bool)
        cp, cm = np.bincount(pos[sigma == 1], minlength=L), np.bincount(pos[sigma == -1], minlength=L)
        t_field = t_rest = 0.0
        for _ in range(n_ev):
            t0 = time.perf_counter()
            field = orc.mean_field(cp, cm)
            t1 = time.perf_counter()
            pos, sigma, bound, tau = orc.fire_event(pos, sigma, bound, field, cp, cm, 0.0, ([], []))
            t2 = time.perf_counter()
            t_field += t1 - t0
            t_rest += t2 - t1
        print(f"{N:7d}  {label:18s} {n_ev / (t_field + t_rest):9.1f}   {t_field / n_ev * 1e3:9.2f}   {t_rest / n_ev * 1e3:9.2f}")
