run() { c=$1; ch=$2; fi=$3; sp=$4; FOKL_BENCH_PIN=0 FOKL_CHAIN_THREADS=$ch FOKL_FINISH_THREADS=$fi FOKL_SPECTRAL_THREADS=$sp taskset -c 0-$((c-1)) python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-microbench --no-parity --no-throughput > /tmp/s.json 2>/dev/null; python -c "
import json; d=json.load(open('/tmp/s.json')); h=d['host_main_thread_s_per_step']; print('cpus $c plan $ch+$fi+$sp  ms/fit %.1f  pool CPU %.3f' % (d['ms_per_step'], h['pool_noise_s']+h['pool_chain_s']+h['pool_finish_s']+h['pool_spectral_s']))"; }
for plan in "16 2 1 4" "16 1 1 4" "16 2 1 5" "16 2 2 4" "16 2 1 3" "8 1 1 3" "8 2 1 3" "8 1 1 4" "8 2 1 4" "6 1 1 3" "6 1 1 4" "6 2 1 3" \
            "5 1 1 3" "5 1 1 2" "4 1 1 2" "4 1 0 2" "4 1 1 3" "3 1 1 2" "3 1 0 2" "3 1 1 1" "2 1 0 1" "2 1 0 2" "2 1 1 1"; do
  run $plan
done
