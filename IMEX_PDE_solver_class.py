"""Drop-in module name of the reference (`from IMEX_PDE_solver_class import IMEXPDE`,
IMEX_PDE_solver_run*.py:1): re-exports the MI355X-hosted class."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
IMEXPDE = importlib.import_module(
    "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd.pde").IMEXPDE
