#!/bin/bash
# AddressSanitizer + UBSan over the product's host sequencing code (csrc/tnn_mlp.cpp is compiled into the CPU twin of the C-ABI,
# oracle/cpu_twin) — GPU sanitizers are not available on this pool, the CPU build is where they run:
#     bash tools/asan_twin.sh [pytest -k expression]
# Builds a sanitized twin, swaps it in for the run (oracle/_build is not tracked), restores the normal one afterwards.
set -u
cd "$(dirname "$0")/.."
GCCLIB=$(dirname "$(g++ -print-file-name=libasan.so)")
TWIN=oracle/_build/libtnn_cpu.so
mkdir -p oracle/_build
python3 -c "import sys; sys.path.insert(0, 'tests'); import conftest; conftest.build_twin()"
cp $TWIN /tmp/libtnn_cpu_normal.so
g++ -O1 -g -std=c++17 -shared -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I include oracle/cpu_twin/tnn_cpu.cpp -o $TWIN
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD=$GCCLIB/libasan.so:$GCCLIB/libubsan.so \
    TNN_HOST_COMPILED=0 python3 -m pytest tests/test_host_logic.py tests/test_bf16_twin.py tests/test_dist_gloo.py tests/test_reference_dropin.py \
    -x -q ${1:+-k "$1"} > /tmp/asan_twin_out.txt 2>&1
rc=$?
cp /tmp/libtnn_cpu_normal.so $TWIN; touch $TWIN
tail -3 /tmp/asan_twin_out.txt
echo "UBSan reports: $(grep -c 'runtime error' /tmp/asan_twin_out.txt)   (full output: /tmp/asan_twin_out.txt)"
exit $rc
