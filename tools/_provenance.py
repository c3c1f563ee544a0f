"""the kernel-source hashes every committed counter summary records (agri-fly_amd/provenance.py, loaded by path: the tools
run without importing the package)"""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    "afe_provenance", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "agri-fly_amd", "provenance.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
STEP_KERNEL, PLANNER_KERNEL, RENDER_KERNEL = _mod.STEP_KERNEL, _mod.PLANNER_KERNEL, _mod.RENDER_KERNEL
kernel_source_hashes = _mod.kernel_source_hashes
