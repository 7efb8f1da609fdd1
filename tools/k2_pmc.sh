#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/k2_pmc
rm -rf $OUT; mkdir -p $OUT
python3 tools/k2_experiment.py > $OUT/plain.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 tools/k2_experiment.py 2 > $OUT/pmc.log 2> $OUT/pmc.err
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/k2_pmc/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'gram_mfma' in r['Kernel_Name']:
            name = r['Kernel_Name'][r['Kernel_Name'].index('<'):r['Kernel_Name'].index('>')+1]
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, d in sorted(agg.items()):
    print(name, {k: round(sum(v)/len(v)) for k, v in d.items()})
PY
