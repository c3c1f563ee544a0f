"""The resident grid's residency is a property of the BUILD: an fp32 one-step instantiation of the persistent step kernel
that needs more than 80 vector registers keeps five waves per SIMD resident instead of six (5 119 workers instead of
6 143) and the 2^20-vehicle headline goes from 19.3 to 20.8 us per step -- which is what four registers in the worker's
path once did.  Read from the code object inside the built library: no GPU needed."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "agri-fly_amd", "lib", "libagrifly_engine.so")
READELF = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"


def gfx950_code_objects(path):
    data = open(path, "rb").read()
    at = 0
    while True:
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", at)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", data, i + 24)
        off = i + 32
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode(errors="replace")
            off += ts
            if "gfx950" in triple and s:
                yield data[i + o:i + o + s]
        at = i + 24


def kernel_metadata(tmp_path):
    kernels = {}
    for k, blob in enumerate(gfx950_code_objects(LIB)):
        f = tmp_path / ("co%d.elf" % k)
        f.write_bytes(blob)
        notes = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            if name:
                kernels[name.group(1)] = {key: int(val) for key, val in re.findall(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count):\s+(\d+)", block)}
    return kernels


@pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists(READELF), reason="needs the built library and llvm-readelf")
def test_fp32_one_step_grid_kernels_keep_six_waves_per_simd(tmp_path):
    kernels = kernel_metadata(tmp_path)
    grid = {n: m for n, m in kernels.items() if "afe_step_persistent_kernelIf" in n}
    assert len(grid) == 24, sorted(grid)                      # FEXT x NOISE(3) x LOGIC x RESIDENT, fp32
    one_step = {n: m for n, m in grid.items() if n.split("afe_step_persistent_kernelIf")[1].split("EEEv")[0].endswith("Lb0")}   # RESIDENT = false
    assert len(one_step) == 12
    for name, m in one_step.items():
        assert m["vgpr_count"] <= 80, (name, m)               # 512 / 80 = 6 waves per SIMD
        assert m["private_segment_fixed_size"] == 0, (name, m)   # and nothing spilled to scratch to get there
    headline = [m for n, m in grid.items() if "IfLb1ELi2ELb0ELb0E" in n]      # FEXT, counter noise, no logic, one step
    assert len(headline) == 1 and headline[0]["vgpr_count"] <= 80
