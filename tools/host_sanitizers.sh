#!/bin/bash
# The HOST side of the engine library -- afe_engine.cpp (rings, handshakes, park / relaunch), afe_aql.cpp (own queue, hand-packed
# kernel arguments), afe_comm.cpp, the launchers in the .hip files (entry tables, planner batches) -- under AddressSanitizer +
# UndefinedBehaviorSanitizer ON THE GPU BOX, beneath the GPU tests that drive those paths.  The six host sources are compiled
# by g++ with gcc's sanitizers, the device objects are the dev-hooks build's (no xnack, no device sanitizer -- the pool has
# neither; ROCm's own ASan runtime intercepts the HSA allocator and aborts without them: tried first).  The launchers inside the
# .hip files stay uninstrumented.  Round-5 review, weak point 8: "host-side engine code is never under a sanitizer".
#
#   make -C agri-fly_amd/csrc asan                                        (here, 1 min; the .so travels to the box with the tree)
#   gpurun --timeout 1800 -- 'bash tools/host_sanitizers.sh'             -> gpurun_out/host_sanitizers/{summary.txt, asan.*}
#   TESTS=tests/test_gpu_abi_abuse.py SOAK="12000 1500" bash tools/host_sanitizers.sh <dir>    one file, and the hand-shake soak
set -u
cd "$(dirname "$0")/.."
LIB=$PWD/agri-fly_amd/lib/asan/libagrifly_engine.so
RT=$(gcc -print-file-name=libasan.so)
OUT=${1:-gpurun_out/host_sanitizers}
mkdir -p "$OUT"; rm -f "$OUT"/asan.* "$OUT"/summary.txt
[ -f "$LIB" ] && [ -f "$RT" ] || { echo "no sanitized library ($LIB) or runtime ($RT)" | tee "$OUT/summary.txt"; exit 2; }
# (and torch's own directory on the search path: under the runtime's dlopen interceptor torch's lazy dlopen("libcaffe2_nvrtc.so")
#  no longer sees the RUNPATH of the library that asks, and every test that makes a device tensor fails in torch's init)
export LD_LIBRARY_PATH=$(python -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))')${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}
# (libstdc++ beside the runtime: the interpreter does not link it, and the runtime resolves __cxa_throw when IT is loaded --
#  the first C++ exception inside torch otherwise ends the process in the interceptor)
export AGRIFLY_ENGINE_LIB=$LIB LD_PRELOAD="$RT $(gcc -print-file-name=libstdc++.so.6)"
# (leaks off: the interpreter's own; every report goes to a file so that a child process's is not lost in a captured pipe; no
#  alternate signal stacks: the runtime cannot unmap the one of a thread RCCL ends and dies in its own CHECK -- the two RCCL tests)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:use_sigaltstack=0:log_path=$PWD/$OUT/asan
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=$PWD/$OUT/asan
{
  echo "library: $LIB"; echo "runtime: $RT"
  echo "== probe"
  timeout 300 python -c '
import importlib, numpy as np, torch
afa = importlib.import_module("agri-fly_amd")
assert "asan" in afa.library()._name, afa.library()._name
e = afa.Ensemble(4096, precision=afa.AFE_F32)
e.set_type_table([afa.params_from_type(5)])
e.step(1000, 50); e.sync(); s = e.get_state(); e.close()
print("probe ok", float(np.abs(s["pos"]).max()))
' 2>&1 | tail -3
  echo "== python -m pytest ${TESTS:-tests} -m gpu"
  # shellcheck disable=SC2086
  timeout 3000 python -m pytest ${TESTS:-tests} -m gpu -q -p no:cacheprovider 2>&1 | tail -15
  if [ -n "${SOAK:-}" ]; then     # SOAK="<first seed> <seeds>": tools/handshake_soak.py on the sanitized library as well
    echo "== python tools/handshake_soak.py $SOAK"
    # shellcheck disable=SC2086
    timeout 3000 python tools/handshake_soak.py $SOAK 2>&1 | grep -v ": ok (" | tail -12
  fi
  echo "== sanitizer reports: $(ls "$OUT" | grep -c '^asan\.')"
  for f in "$OUT"/asan.*; do [ -f "$f" ] && { echo "--- $f"; head -40 "$f"; }; done
} > "$OUT/summary.txt" 2>&1
tail -60 "$OUT/summary.txt"
