#!/bin/bash
# The CPU parity tests on the host code under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU needed):
#   tools/sanitize.sh [pytest arguments]      default: the device-code-on-CPU, hostile-scene and golden-fixture suites
# Builds build/san/ (python -m raytracer_amd.build --sanitize), then runs pytest with the sanitized libraries in place of
# the product library's host side, the host simulator (also -DRT_BOUNDS_CHECK: every reference checked against its section)
# and the oracle. python itself is not instrumented, so the clang ASan runtime is preloaded and leak checking is off.
set -e
cd "$(dirname "$0")/.."
python -m raytracer_amd.build --sanitize
rt=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
tests=${@:-tests/test_device_code_on_cpu.py tests/test_hostile_scenes.py tests/test_golden.py tests/test_abi_and_compiler.py tests/test_oracle.py}
LD_PRELOAD="$rt${LD_PRELOAD:+:$LD_PRELOAD}" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
RT_HIP_LIB=build/san/librt_hip_san.so RT_HOSTSIM_SO=build/san/libhostsim_san.so RT_ORACLE_SO=build/san/liboracle_san.so \
python -m pytest $tests -x -q -m "not gpu" -p no:cacheprovider
