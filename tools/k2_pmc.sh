#!/bin/bash
# PMC counters of the Gram kernels on the benchmark fit's block shapes (tools/k2_experiment.py), both MFMA forms.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/k2_pmc
rm -rf $OUT; mkdir -p $OUT
for form in 0 1; do
  FOKL_GRAM_MFMA4=$form rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $OUT/pmc$form -- python3 tools/k2_experiment.py 2 2 > $OUT/pmc$form.log 2> $OUT/pmc$form.err || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for form in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f'gpurun_out/k2_pmc/pmc{form}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            if 'gram_tiles' in r['Kernel_Name']:
                name = r['Kernel_Name'][r['Kernel_Name'].index('gram_tiles'):r['Kernel_Name'].index('>') + 1] + ' grid ' + r.get('Grid_Size', '?') + ' lds ' + r.get('LDS_Block_Size', '?')
                agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== FOKL_GRAM_MFMA4 =', form)
    for name, d in sorted(agg.items()):
        print(name, {k: round(sum(v) / len(v)) for k, v in d.items()})
PY
