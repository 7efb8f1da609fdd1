#!/bin/bash
# PMC counters of the Gram kernels on one block shape (default 56 x 128), 16x16x4 tile lists (FOKL_GRAM_MFMA4=0) against the
# opt-in 4x4x4 form (=2); two passes of eight counters each.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/k2_pmc
rm -rf $OUT; mkdir -p $OUT
export K2_SHAPES=${K2_SHAPES:-56x128}
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
for form in 0 2; do
  FOKL_GRAM_MFMA4=$form rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/a$form -- python3 tools/k2_experiment.py 3 2 > $OUT/a$form.log 2> $OUT/a$form.err || exit 1
  FOKL_GRAM_MFMA4=$form rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/b$form -- python3 tools/k2_experiment.py 3 2 > $OUT/b$form.log 2> $OUT/b$form.err || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for form in (0, 2):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ('a', 'b'):
        for path in glob.glob(f'gpurun_out/k2_pmc/{sub}{form}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(path)):
                if 'gram_tiles' in r['Kernel_Name']:
                    name = r['Kernel_Name'][r['Kernel_Name'].index('gram_tiles'):r['Kernel_Name'].index('>') + 1]
                    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== FOKL_GRAM_MFMA4 =', form)
    for name, d in sorted(agg.items()):
        print(name, {k: round(sum(v) / len(v)) for k, v in sorted(d.items())})
PY
