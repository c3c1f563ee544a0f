import importlib
import os
import sys

import pytest

# torch bundles its own HIP runtime (same SONAME as /opt/rocm's).  Whichever
# libamdhip64 is loaded first serves the whole process, so a process that uses
# both torch.cuda and the engine must import torch BEFORE the engine library is
# loaded (otherwise torch finds "No HIP GPUs").  See INTEGRATION.md.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def afa():
    """The product package (directory name has a hyphen)."""
    return importlib.import_module("agri-fly_amd")


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """GPU parity tests record their measured worst errors (tests/scenarios.py LEDGER); on the GPU box
    the ledger travels back through gpurun_out/ and is committed as profiles/parity_rNN.json."""
    try:
        from tests import scenarios
    except Exception:
        return
    if not scenarios.LEDGER and not scenarios.MEASUREMENTS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_ledger.json"), "w") as f:
        json.dump({"definition": "per vehicle ||engine - oracle||_2 / max(||oracle||_2, floor), worst vehicle; "
                                 "tolerance 1e-5 for the fp32 engine, <= 1e-10 for the fp64 engine",
                   "floors": scenarios.FLOORS, "tests": scenarios.LEDGER, "measurements": scenarios.MEASUREMENTS}, f, indent=1, sort_keys=True)
