set -u
run() { echo "== $*"; env "$@" python -m pytest tests/test_config_goldens.py tests/test_gpu_parity.py -m gpu -x -q -k "cfg4_unit0 or cfg1 or cfg3 or fit_matches or testdata10" 2>&1 | tail -1; }
run FOKL_TENTATIVE_TAPES=0
run FOKL_TENTATIVE_TAPES=test
run FOKL_SPECULATION=1
run FOKL_SPECULATION=16 FOKL_LOOKAHEAD=6
run FOKL_FORESIGHT=0
run FOKL_KILL_BIC=device
run FOKL_KILL_BIC=check
run FOKL_NOISE_PIPELINE=0
run FOKL_FINISH_THREADS=0
run FOKL_FINISH_THREADS=3 FOKL_FINISH_LOG=exact
run FOKL_CHAIN_THREADS=1 FOKL_SPECTRAL_THREADS=1
run FOKL_CHAIN_ISA=base FOKL_SAMPLER_ISA=base
run FOKL_K3=columns
