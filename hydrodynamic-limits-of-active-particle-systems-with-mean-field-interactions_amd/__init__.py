"""MI355X-native time-stepper for the 1-D active lattice gas with Curie-Weiss mean-field interactions.

Only what the hot path and its neighbouring rows of SURVEY 8 need lives here:
  csrc/aps_hip.hip        HIP kernels (gfx950) + the C ABI of include/aps.h: synchronous stepper, lattice and all-pairs
                          formulations, observation kernels
  csrc/gillespie_hip.hip  the reference's exact event loop resident on the GPU (include/gillespie.h)
  csrc/pde_hip.hip        hydrodynamic-limit PDE + Euler-Maruyama tracers (include/pde.h)
  csrc/aps_common.hpp     device code shared by the translation units (rates, Philox, exact-grid weight table)
  capi.py                 ctypes binding of include/aps.h
  particle_system.py      ParticleSystem: the reference's construct / run() surface
  gillespie.py            run_batched_exact: many systems of the exact loop in one launch
  pde.py                  IMEXPDE: the reference's PDE class surface
  observables.py          the sweep drivers' observables (host formulas; device-side integer sums)
  sharded.py              particle-index sharding across GPUs (one process per GPU, one all-gather per step)
  ensemble.py             batched independent ensembles (beta sweeps) on one GPU
"""
from .particle_system import ParticleSystem  # noqa: F401
from .pde import IMEXPDE  # noqa: F401
