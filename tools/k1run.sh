for t in 256 128 64; do echo "== FOKL_K1_THREADS=$t"; FOKL_K1_THREADS=$t python tools/k1_experiment.py 2>&1 | grep -E "one launch|T=28|T=8"; done
echo "== heuristic"; python tools/k1_experiment.py 2>&1 | grep -E "one launch|T=28|T=8"
