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


def kernel_arguments(tmp_path):
    """{kernel name: ([(offset, size, value_kind), ...], kernarg_segment_size)} of every kernel in the gfx950 code objects"""
    kernels = {}
    for k, blob in enumerate(gfx950_code_objects(LIB)):
        f = tmp_path / ("args%d.elf" % k)
        f.write_bytes(blob)
        notes = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            seg = re.search(r"\.kernarg_segment_size:\s+(\d+)", block)
            if not name or not seg:
                continue
            args_text = block.split(".args:")[1].split(".group_segment_fixed_size")[0] if ".args:" in block else ""
            args = []
            for item in args_text.split("- .")[1:]:
                fields = dict(re.findall(r"\.?(offset|size|value_kind):\s+(\S+)", item))
                args.append((int(fields["offset"]), int(fields["size"]), fields["value_kind"]))
            kernels[name.group(1)] = (args, int(seg.group(1)))
    return kernels


@pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists(READELF), reason="needs the built library and llvm-readelf")
def test_host_kernarg_layout_is_the_code_objects_argument_table(tmp_path):
    """Round-4 review: the resident grid's dispatch on the engine's own AQL queue packs the kernel-argument segment by
    hand and checked the TOTAL size only.  The host now packs through one struct (afe_device.h PersistKernarg) and says
    where it puts each argument (afe_persistent_kernarg_layout); here every one of the 48 instantiations' argument tables
    in the code object -- offset, size, by-value kind of each of the four arguments, and the segment size, hidden
    arguments included if the compiler ever adds any -- must be exactly that."""
    import ctypes as C
    import importlib
    afa = importlib.import_module("agri-fly_amd")
    L = afa.library()
    kernels = kernel_arguments(tmp_path)
    seen = 0
    for precision, tag in ((afa.AFE_F32, "If"), (afa.AFE_F64, "Id")):
        off, size, seg = (C.c_int32 * 4)(), (C.c_int32 * 4)(), C.c_int32(0)
        assert L.afe_persistent_kernarg_layout(precision, off, size, C.byref(seg)) == 0
        host = [(off[k], size[k], "by_value") for k in range(4)]
        grid = {n: v for n, v in kernels.items() if "afe_step_persistent_kernel" + tag in n}
        assert len(grid) == 24, sorted(grid)
        for name, (args, segment) in grid.items():
            assert args == host, (name, args, host)
            assert segment == seg.value, (name, segment, seg.value)       # the segment ends with its last argument: no hidden arguments
            seen += 1
    assert seen == 48
