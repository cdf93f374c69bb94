"""MI355X-native time-stepper for the 1-D active lattice gas with Curie-Weiss mean-field interactions.

Only what the hot path needs lives here:
  csrc/aps_hip.hip     HIP kernels (gfx950) + the C ABI of include/aps.h
  capi.py              ctypes binding
  particle_system.py   ParticleSystem: the reference's construct / run() surface
  sharded.py           particle-index sharding across GPUs (one process per GPU, one all-gather per step)
  ensemble.py          batched independent ensembles (beta sweeps) on one GPU
"""
from .particle_system import ParticleSystem  # noqa: F401
