"""Drop-in module name of the reference: `from PARTICLE_solver_CLASS import ParticleSystem`
(PARTICLE_solver_BIOLOGY_EXCLUSION*.py line 12/13).  Re-exports the MI355X-backed class."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module(
    "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd")
ParticleSystem = _pkg.ParticleSystem
