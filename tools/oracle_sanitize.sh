#!/bin/bash
# The CPU oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer: rebuilds oracle/libpwn_oracle.so with the
# sanitizers, runs the CPU tests that drive it, and restores the normal build.  CPU only (GPU sanitizers are not available on this pool).
set -e
cd "$(dirname "$0")/.."
trap 'make -C oracle -B >/dev/null' EXIT
make -C oracle -B CXXFLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-sanitize-recover=undefined" >/dev/null
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest -q -x -m "not gpu" tests/test_oracle_cpu.py tests/test_golden.py tests/test_oracle_vs_numpy_model.py tests/test_tracker.py tests/test_priors.py \
  tests/test_statistics.py tests/test_scene.py tests/test_matcher.py tests/test_reference_octave_model.py
