#!/bin/bash
# The library's HOST sources (runtime, EQ runtime, parsers, table builders) under AddressSanitizer + UBSan, CPU only (the GPU boxes
# offer no sanitizer runs).  Host sources are compiled by g++ against the HIP headers and linked with the device objects of the normal
# build; the result, airwave_amd/libairwave_hip_asan.so, is loaded only through AIRWAVE_HIP_LIBRARY.  Then: the CPU test suite's
# library-facing tests and tools/fuzz_host.py (byte-level fuzz of every parser) run against it with UBSan set to halt.
#   bash tools/asan_host.sh [fuzz seconds, default 60]
set -e
cd "$(dirname "$0")/.."
python airwave_amd/build.py > /dev/null
B=airwave_amd/_build_asan; mkdir -p $B
for f in runtime.cpp eq_runtime.cpp host/eq.cpp host/tables.cpp host/host_api.cpp; do
    g++ -O1 -g -std=c++17 -fPIC -pthread -fvisibility=hidden -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ \
        -I/opt/rocm/include -w -c airwave_amd/csrc/$f -o $B/$(echo $f | tr / _).o &
done
wait
g++ -shared -fPIC -pthread -fsanitize=address,undefined -o airwave_amd/libairwave_hip_asan.so $B/*.o airwave_amd/_build/device_*.o \
    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_asan.so
python -m pytest tests -q -m "not gpu" -x -k "capi or eq_host or eq_fold_host or data_model or mixed or effect or contract or abi or provenance" 2>&1 | tail -3
python tools/fuzz_host.py --seconds "${1:-60}" 2>&1 | grep -v RuntimeWarning | grep -v "astype" | tail -3
