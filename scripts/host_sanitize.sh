#!/bin/bash
# Host-side sanitizer pass (runs in the build container, no GPU): the library's HOST code compiled with AddressSanitizer (or
# UndefinedBehaviorSanitizer: SAN=undefined) — device code is left alone (-fno-gpu-sanitize; GPU sanitizers are not available
# on the pool) — and the CPU suite run against that build. Outputs under build/ (git-ignored, not shipped to the GPU box).
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd); SAN=${SAN:-address}; B=$R/build/$SAN; mkdir -p $B; cd $R/aindex_amd/csrc
for f in $(ls *.hip | sed 's/\.hip$//'); do
  [ $B/$f.o -nt $f.hip ] || /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=$SAN -fno-gpu-sanitize -fno-sanitize-recover=all -Wno-unused-result -c $f.hip -o $B/$f.o || exit 2
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=$SAN -fno-gpu-sanitize -shared-libsan -o $B/libaindex_hip.so $B/*.o || exit 2
cat > $B/run.py <<PY
import sys
sys.path.insert(0, "$R")
import aindex_amd._lib as L
L.LIB_PATH = "$B/libaindex_hip.so"
import pytest
# the C-link test builds a plain gcc program against the library (no sanitizer runtime); the gloo tests start fresh interpreters
sys.exit(pytest.main(["-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "$R/tests", "-k", "not test_header_is_plain_c_and_links", "--ignore", "$R/tests/test_dist_gloo.py"]))
PY
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
[ "$SAN" = address ] || RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
cd $R && LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python $B/run.py
