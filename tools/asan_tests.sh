#!/bin/bash
# the CPU test suite on the sanitized host code (make -C spacepharer_amd/csrc asan): AddressSanitizer + UBSan reports abort the run
cd "$(dirname "$0")/.."
make -C spacepharer_amd/csrc asan > /dev/null || exit 1
export SP_LIB_NAME=libspacepharer_amd_asan.so
export LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python -m pytest ${@:-tests/test_host_logic.py tests/test_set_modules.py tests/test_host_fuzz.py} -x -q -m "not gpu" -p no:cacheprovider
