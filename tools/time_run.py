import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from PARTICLE_solver_CLASS import ParticleSystem
kw = dict(L=1000, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, init="fixed", N=500, scale_rates=False, local_kernel_sigma=0.005,
          site_capacity=1, k_on=0.0, k_off=0.0, k_exit=0.0)
for mode in ("sync", "gillespie_gpu"):
    for rep in range(2):
        ps = ParticleSystem(beta=0.7, rng=np.random.default_rng(1), seed=5, mode=mode, **kw)
        t0 = time.perf_counter()
        out = ps.run(T=20.0, obs_dt=0.1, record_fft=True, record_var=True)
        print(mode, rep, "run() wall %.3f s" % (time.perf_counter() - t0), getattr(ps, "steps_done", None), getattr(ps, "n_events", None))
import cProfile, pstats
ps = ParticleSystem(beta=0.7, rng=np.random.default_rng(1), seed=5, mode="sync", **kw)
pr = cProfile.Profile(); pr.enable(); ps.run(T=20.0, obs_dt=0.1); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
