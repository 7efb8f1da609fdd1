#!/bin/bash
# tools/knob_suite.sh [out]: the fit-level GPU tests (goldens of every BASELINE config, reference fixtures, device chains,
# native search) once per value of every environment knob the PRODUCT library and driver read (README's table; knobs
# that only select kernel variants exist in DEV=1 builds and are not part of the product).  One pytest process per mode,
# one after the other; stops at the first failing mode.
out=${1:-gpurun_out/knob_suite.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
TESTS="tests/test_config_goldens.py tests/test_gpu_parity.py tests/test_chain_device.py tests/test_fitupdate.py tests/test_derivatives.py"
KEYS="full_size_config_on_gpu or fit_matches_reference or reference_test_dataset or fitupdate or derivatives or against_oracle or kill_test_bic"
modes=(
  "FOKL_X=default"
  "FOKL_SEARCH=python"
  "FOKL_CHAIN=host"
  "FOKL_DCHAIN_ROWS=0"
  "FOKL_DCHAIN_RECURSION=exact"
  "FOKL_EIGH=device"
  "FOKL_EIGH=hybrid"
  "FOKL_EIGH_DC_FROM=0"
  "FOKL_EIGH_DC_FROM=8"
  "FOKL_EIGH_UPDATE=0"
  "FOKL_EIGH_UPDATE_DEPTH=64"
  "FOKL_BUILD_AHEAD=tests"
  "FOKL_LOOKAHEAD_DERIVED=24"
  "FOKL_CLEAN=host"
  "FOKL_KILL_DECIDE=g2"
  "FOKL_G2_DEFER_FROM=8"
  "FOKL_SPECULATION=4"
  "FOKL_SPECULATE_ACROSS=0"
  "FOKL_FORECAST_EARLY=0"
  "FOKL_STATS=numpy"
  "FOKL_SPIN=0"
  "FOKL_GUESS_MARGIN=0.02"
  "FOKL_EIGH_UPDATE_DEPTH=6"
  "FOKL_SEARCH_DIST=python"
  "FOKL_K3=columns"
  "FOKL_KILL_BIC=device"
  "FOKL_KILL_BIC=check"
  "FOKL_FINISH_LOG=exact"
  "FOKL_NOISE_PIPELINE=0"
  "FOKL_TENTATIVE_TAPES=0"
  "FOKL_TENTATIVE_TAPES=test"
  "FOKL_FORESIGHT=0"
  "FOKL_PIN_L3=0"
  "FOKL_SAMPLER_ISA=base"
  "FOKL_SYNC=blocking"
  "FOKL_HEAD_START=0"
  "FOKL_EIGH_DGEMM_FROM=32"
  "FOKL_SUBSTAGE_LOOP=python"
  "FOKL_WALK_HELPERS=0"
  "FOKL_WALK_HELPERS=3"
  "FOKL_WALK_CPUS=same"
  "FOKL_BULK_CPUS=same"
  "FOKL_CREW_INLINE=0"
  "FOKL_STREAM_WALK=positions"
  "FOKL_SEGMENT_STORES=cached"
  "FOKL_TEMPER_CHUNK=16"
  "FOKL_STREAM_AHEAD=fixed"
  "FOKL_DCHAIN_RECURSION=serial"
  "FOKL_ROW_STORES=cached"
  "FOKL_BUILD_AHEAD_AT=model"
  "FOKL_WALK_PREFETCH=first"
  "FOKL_CREW_DEPTH=16"
)
# KNOB_PART=k/n: every n-th mode from the k-th on (a gpurun call is limited to 20 minutes)
part=${KNOB_PART:-1/1}; k=${part%%/*}; n=${part##*/}; i=0
for mode in "${modes[@]}"; do
  i=$((i + 1)); if [ $(( (i - k) % n )) -ne 0 ] || [ $i -lt $k ]; then continue; fi
  echo "== $mode" | tee -a "$out"
  env $mode timeout -k 10 900 python -m pytest $TESTS -m gpu -q -x -k "$KEYS" 2>&1 | tail -2 | tee -a "$out"
  if [ "${PIPESTATUS[0]}" -ne 0 ]; then echo "FAILED under $mode" | tee -a "$out"; exit 1; fi
done
echo "all modes passed" | tee -a "$out"
