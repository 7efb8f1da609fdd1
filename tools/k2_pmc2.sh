#!/bin/bash
# PMC counters of gram_tiles_dma_kernel per block shape (one shape per run: K2_SHAPES), two passes of eight counters.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/k2_pmc2
rm -rf $OUT; mkdir -p $OUT
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
for shape in ${SHAPES:-56x98 56x176}; do
  K2_SHAPES=$shape rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/a_$shape -- python3 tools/k2_experiment.py 3 2 > $OUT/a_$shape.log 2> $OUT/a_$shape.err || exit 1
  K2_SHAPES=$shape rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/b_$shape -- python3 tools/k2_experiment.py 3 2 > $OUT/b_$shape.log 2> $OUT/b_$shape.err || exit 1
done
python3 - <<'PY'
import csv, glob, collections, os
for shape in os.environ.get('SHAPES', '56x98 56x176').split():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ('a', 'b'):
        for path in glob.glob(f'gpurun_out/k2_pmc2/{sub}_{shape}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(path)):
                if 'gram_tiles_dma' in r['Kernel_Name']:
                    name = r['Kernel_Name'][r['Kernel_Name'].index('gram_tiles'):r['Kernel_Name'].index('>') + 1]
                    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== shape', shape)
    for name, d in sorted(agg.items()):
        print(name, {k: round(sum(v) / len(v)) for k, v in sorted(d.items())})
PY
