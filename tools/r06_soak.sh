#!/bin/bash
# round 6: many fits back to back on the final tree (every configuration; the headline one with and without walk helpers, the
# Python sub-stage loop, four processes side by side): parity on every line, no search repeated, no failure
set -o pipefail
run() { name=$1; shift
  "$@" > gpurun_out/soak_$name.json 2> gpurun_out/soak_$name.err || { echo "FAILED $name"; tail -c 600 gpurun_out/soak_$name.err; exit 1; }
  python - $name <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/soak_{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d.get("host_main_thread_s_per_step", {})
print(sys.argv[1], "steps", d["steps"], "ms", round(d["ms_per_step"], 2), "parity", d.get("parity", {}).get("ok"), "searches repeated per step", h.get("searches_repeated"),
      "guessed == verified", h.get("guessed") == h.get("guesses_verified"), "throughput", (d.get("throughput_mode") or {}).get("value"), flush=True)
PY
}
F="--no-cpu-baseline --no-microbench"
run cfg2_600 python bench.py --steps 600 --warmup 5 $F --no-throughput
run cfg2_h0_300 env FOKL_WALK_HELPERS=0 python bench.py --steps 300 --warmup 5 $F --no-throughput
run cfg2_h3_300 env FOKL_WALK_HELPERS=3 python bench.py --steps 300 --warmup 5 $F --no-throughput
run cfg2_pyloop_200 env FOKL_SUBSTAGE_LOOP=python python bench.py --steps 200 --warmup 5 $F --no-throughput
run cfg1_400 python bench.py --config 1 --steps 400 --warmup 5 $F --no-throughput
run cfg4_32 python bench.py --config 4 --steps 32 --warmup 2 $F --no-throughput
run cfg3_12 python bench.py --config 3 --steps 12 --warmup 1 $F --no-throughput
for i in 1 2 3 4; do run tp_$i python bench.py --steps 10 --warmup 2 $F; done
echo soak done
