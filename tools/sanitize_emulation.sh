#!/bin/bash
# AddressSanitizer / UBSan over the kernel bodies: the CPU emulation libraries of tests/emulation.py (disco_core.hpp, disco_vec_core.hpp,
# ts_core.hpp compiled for the host) rebuilt with -fsanitize=address,undefined and the emulation tests run against them.
# (GPU sanitizers are not available on the pool; this is the host-side check of the same arithmetic and indexing.)
set -e
cd "$(dirname "$0")/.."
export SASKTRAN2_AMD_EMU_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
python -m pytest tests/test_kernel_emulation.py tests/test_vector_emulation.py tests/test_ts_emulation.py -x -q "$@"
