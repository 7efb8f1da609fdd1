#!/bin/bash
# PMC counters of the Gram launches where the device binds (N = 2e7, the shapes N >= 1e7 fits spend their
# time on), one shape per run, separate passes (at most eight SQ counters / four TCC slots per pass, never with a sys trace).
#   gpurun: bash tools/r05_k2_pmc.sh            SHAPES / K2_N override
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_k2_pmc
rm -rf $OUT; mkdir -p $OUT
export K2_N=${K2_N:-20000000}
export SHAPES=${SHAPES:-"56x75 56x101 56x128"}
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
C="FETCH_SIZE"
D="TCC_HIT_sum TCC_MISS_sum"
for shape in $SHAPES; do
  # the plain timing first (no counters): the microseconds the ratios below are read against
  K2_SHAPES=$shape timeout -k 10 300 python3 tools/k2_experiment.py 10 2 > $OUT/t_$shape.log 2> $OUT/t_$shape.err || exit 1
  for pass in a b c d; do
    case $pass in a) P=$A;; b) P=$B;; c) P=$C;; d) P=$D;; esac
    K2_SHAPES=$shape timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${pass}_$shape -- python3 tools/k2_experiment.py 3 2 > $OUT/${pass}_$shape.log 2> $OUT/${pass}_$shape.err
    rc=$?
    if [ $rc -ne 0 ]; then
      echo "pass $pass of $shape: exit $rc" >> $OUT/failed.txt
      [ $pass = a ] || [ $pass = b ] && exit 1          # (TCC passes: names may not exist on this build; go on without)
    fi
  done
done
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections, os, re
n = int(os.environ['K2_N'])
print(f"# tools/r05_k2_pmc.sh: gram_tiles_dma_kernel at N = {n:,d}, per launch (averages over the launches of tools/k2_experiment.py 3 2)")
for shape in os.environ['SHAPES'].split():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in 'abcd':
        for path in glob.glob(f'gpurun_out/r05_k2_pmc/{sub}_{shape}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(path)):
                if 'gram_tiles_dma' in r['Kernel_Name']:
                    name = r['Kernel_Name'][r['Kernel_Name'].index('gram_tiles'):r['Kernel_Name'].index('>') + 1]
                    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== shape', shape)
    for line in open(f'gpurun_out/r05_k2_pmc/t_{shape}.log'):
        if line.startswith('gram'):
            print('   timing (no counters):', line.strip())
    for name, d in sorted(agg.items()):
        c = {k: sum(v) / len(v) for k, v in d.items()}
        print('  ', name, {k: round(v) for k, v in sorted(c.items())})
        g = c.get
        if g('SQ_BUSY_CU_CYCLES') and g('SQ_WAVE_CYCLES'):
            wc = g('SQ_WAVE_CYCLES')
            print(f"     matrix pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * g('SQ_BUSY_CU_CYCLES')):.3f} of the CUs' busy cycles; of the wavefronts' cycles: "
                  f"parked (s_waitcnt / barrier) {g('SQ_WAIT_ANY', 0) / wc:.3f}, issue-stalled {g('SQ_WAIT_INST_ANY', 0) / wc:.3f} "
                  f"(of which on LDS issue {g('SQ_WAIT_INST_LDS', 0) / wc:.3f})")
        if g('SQ_INSTS_MFMA'):
            mf = g('SQ_INSTS_MFMA')
            print(f"     per MFMA: LDS instructions {g('SQ_INSTS_LDS', 0) / mf:.2f}, VALU other than MFMA {(g('SQ_INSTS_VALU', 0) - mf) / mf:.2f}, "
                  f"SALU {g('SQ_INSTS_SALU', 0) / mf:.2f}; LDS bank-conflict cycles {g('SQ_LDS_BANK_CONFLICT', 0):.0f}; "
                  f"algorithmic MFMAs (tiles run x rows / 4): see the plan in the timing line")
        if g('FETCH_SIZE'):
            print(f"     FETCH_SIZE {g('FETCH_SIZE'):.0f} KB as reported = {2 * g('FETCH_SIZE') * 1024 / 1e9:.3f} GB after the gfx950 x2 correction (MI355X_MICROARCH.md, HBM section)")
        if g('TCC_HIT_sum') is not None and g('TCC_MISS_sum') is not None and g('TCC_HIT_sum') + g('TCC_MISS_sum') > 0:
            print(f"     L2 hit rate {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}")
if os.path.exists('gpurun_out/r05_k2_pmc/failed.txt'):
    print(open('gpurun_out/r05_k2_pmc/failed.txt').read())
PY
cat $OUT/summary.txt
