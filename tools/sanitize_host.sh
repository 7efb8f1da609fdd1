#!/bin/bash
# Host-side sanitizer runs (GPU sanitizers are not available on this pool): the random stream, the noise / finish / chain /
# spectral threads and the search's use of them under ThreadSanitizer and under AddressSanitizer + UBSan.
#   tools/sanitize_host.sh [tsan|asan] [pytest args ...]
# Builds fokl_gpy_amd/libfokl_host_<kind>.so (csrc/Makefile) and runs tests/test_sampler_host.py plus the pool / search
# tests of tests/test_host_logic.py on it; reports go to profiles/sanitize_<kind>_r06.txt.
set -u
cd "$(dirname "$0")/.."
kind=${1:-tsan}; shift || true
make -C fokl_gpy_amd/csrc host-$kind >/dev/null || exit 1
case $kind in
  tsan) runtime=$(gcc -print-file-name=libtsan.so)
        export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan.supp second_deadlock_stack=1" ;;
  asan) runtime="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
        export ASAN_OPTIONS="detect_leaks=0 abort_on_error=0 halt_on_error=0" UBSAN_OPTIONS="print_stacktrace=1" ;;
  *) echo "tsan or asan"; exit 2 ;;
esac
out=profiles/sanitize_${kind}_r06.txt
# (tests that start other processes are left out: a fork from a sanitized, multi-threaded interpreter does not come back)
tests=${*:-"tests/test_sampler_host.py tests/test_stream_engine.py tests/test_native_search.py tests/test_eigen_update.py tests/test_host_logic.py -k '(native_search or eigen or pool or pipelined or device_chains or reap or tape or sampler or stream or chain or numpy or legacy or gibbs or fit) and not recorders_agree and not statements_agree and not rendezvous and not stale_file'"}
{
  echo "# $(date -u +%F) $kind: FOKL_HIP_LIBRARY=fokl_gpy_amd/libfokl_host_$kind.so FOKL_HOST_ONLY_LIBRARY=1, runtime preloaded"
  echo "# python -m pytest $tests"
} > "$out"
# (the runtime is preloaded into the interpreter only: a shell under ThreadSanitizer does not get far)
# OpenBLAS on one thread: its own (uninstrumented) thread pool dead-locks under ThreadSanitizer's interposed pthread calls
# in numpy's dot as well as in scipy's eigh -- with or without this library's threads in the process.
OPENBLAS_NUM_THREADS=1 FOKL_HIP_LIBRARY=$PWD/fokl_gpy_amd/libfokl_host_$kind.so FOKL_HOST_ONLY_LIBRARY=1 FOKL_PIN_L3=0 \
  bash -c "LD_PRELOAD='$runtime' timeout 1500 python -m pytest $tests -q -s -p no:cacheprovider" >> "$out" 2>&1
rc=$?
echo "# exit code $rc; reports: $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error:' "$out")" >> "$out"
tail -5 "$out"
exit $rc
