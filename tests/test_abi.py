"""The C-ABI library loads (without a GPU) and exports every symbol the header
declares; host-only entry points behave; no compute calls are made here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "agrifly_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(afe_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree(afa):
    assert _header_functions() == sorted(afa.ABI_FUNCTIONS)


def test_library_exports_every_declared_symbol(afa):
    lib = C.CDLL(afa.library_path())
    for name in _header_functions():
        assert hasattr(lib, name), "missing export: " + name
    assert afa.library().afe_abi_version() == 3


def test_library_carries_gfx950_code_object(afa):
    """the product is HIP for gfx950, not a host stub"""
    blob = open(afa.library_path(), "rb").read()
    assert b"__CLANG_OFFLOAD_BUNDLE__" in blob
    assert b"hipv4-amdgcn-amd-amdhsa--gfx950" in blob
    # gfx950 only: no other offload target in the bundle
    import re as _re
    targets = set(_re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "agrifly_engine.h"\nint main(void){afe_vehicle_params p; (void)p; return AFE_ABI_VERSION==3?0:1;}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "t2.o")])


def test_struct_layout_matches_header(afa, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "agrifly_engine.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu\\n", sizeof(afe_vehicle_params),'
                   ' offsetof(afe_vehicle_params, lin_drag_coeff_b), offsetof(afe_vehicle_params, imu_yaw),'
                   ' sizeof(afe_device_view)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    a, b, c, d = map(int, subprocess.check_output([str(exe)]).split())
    assert a == C.sizeof(afa.VehicleParams)
    assert b == afa.VehicleParams.lin_drag_coeff_b.offset
    assert c == afa.VehicleParams.imu_yaw.offset
    assert d == C.sizeof(afa.DeviceView)


def test_status_strings(afa):
    L = afa.library()
    assert L.afe_status_string(0) == b"ok"
    assert b"gfx950" in L.afe_status_string(2)


def test_plan_ticks_argument_checks(afa):
    L = afa.library()
    assert L.afe_plan_ticks(0.002, None, 1000, 1, None) == 1  # AFE_ERR_INVALID_ARG
    ticks, el = afa.plan_ticks(0.002, 0, 0, 5)                # dt == 0: Run() returns early
    assert ticks.tolist() == [0] * 5 and el == 0


def test_no_device_means_loud_failure_not_fallback(afa):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(afa.AfeError) as ei:
        afa.Ensemble(16)
    assert ei.value.status == 2  # AFE_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import,
    link or call anything under oracle/"""
    bad = re.compile(r"(from\s+oracle|import\s+oracle|oracle_py|agrifly_oracle|\bora_[a-z_]+\s*\(|oracle/)")
    roots = [os.path.join(ROOT, "agri-fly_amd"), os.path.join(ROOT, "include")]
    for root in roots:
        for dirpath, _, files in os.walk(root):
            if os.path.basename(dirpath) in ("lib", "asm", "__pycache__"):
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp", "Makefile")):
                    text = open(os.path.join(dirpath, f)).read()
                    m = bad.search(text)
                    assert not m, "%s references the oracle: %r" % (os.path.join(dirpath, f), m.group(0))


def test_every_entry_point_survives_null_and_zero_arguments(afa):
    """The boundary's error behaviour: called with NULL for every pointer / handle and 0 for every number, each of the
    header's functions RETURNS -- an AFE_ERR_* where it takes a handle or has to write somewhere -- instead of
    dereferencing.  In a child process, so that a crash is this test's failure and not the run's end.  (With live handles
    and bad data arguments: tests/test_gpu_abi_abuse.py.)"""
    code = r'''
import ctypes as C, importlib, sys
sys.path.insert(0, %r)
afa = importlib.import_module("agri-fly_amd")
L = afa.library()
def zero(t):
    if t is C.c_void_p or (hasattr(t, "_type_") and not isinstance(t._type_, str)): return None
    return t(0)
ok_with_nothing = {"afe_abi_version", "afe_has_dev_hooks", "afe_planner_release_scratch", "afe_device_free", "afe_type_from_id"}
for name in sorted(afa.ABI_FUNCTIONS):
    fn = getattr(L, name)
    rc = fn(*[zero(t) for t in fn.argtypes])
    if fn.restype is C.c_int and name not in ok_with_nothing:
        assert rc != 0, name + " accepted NULL / 0 for everything"
print("survived", len(afa.ABI_FUNCTIONS))
''' % ROOT
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "survived" in out.stdout, (out.returncode, out.stdout[-500:], out.stderr[-1500:])
