"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

    python tools/rocprof_summary.py stats   <dir> <out.md>     # --kernel-trace --stats run
    python tools/rocprof_summary.py pmc     <dir> <out.md> COUNTER [COUNTER...]   # --pmc run (+ --kernel-trace)

Per kernel: calls, total / average / min / max duration (from the kernel trace), and for PMC runs the per-dispatch
average of each counter.  FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced streams (MI355X_MICROARCH.md, HBM section), so the summary
prints both the raw value and the doubled "corrected" read bytes.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(dirname, suffix):
    hits = sorted(glob.glob(os.path.join(dirname, '**', '*' + suffix), recursive=True))
    return hits


def short(name):
    name = name.replace('void ', '')
    for key in ('basis_build_reg_kernel', 'basis_build_kernel', 'gram_tiles_dma_kernel', 'gram_tiles4s_kernel', 'gram_tiles_kernel',
                'gram_mfma_kernel', 'gram_valu_kernel',
                'resid_quadratic_kernel', 'resid_terms_lds_kernel', 'resid_terms_kernel', 'resid_kernel', 'reduce_slabs_sym_kernel', 'reduce_slabs_kernel',
                'transpose_inputs_kernel', 'predict_mfma_kernel', 'predict_kernel', 'gibbs_chain_kernel', 'tape_gather_kernel'):
        if key in name:
            tag = ''
            if '<' in name:
                tag = name[name.index('<'):name.index('>') + 1] if '>' in name else ''
            return 'fokl::' + key + tag
    return name[:70]


def kernel_trace_rows(dirname):
    rows = []
    for path in find(dirname, 'kernel_trace.csv'):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    return rows


def cmd_stats(dirname, out):
    agg = defaultdict(list)
    for r in kernel_trace_rows(dirname):
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3     # us
        agg[short(r['Kernel_Name'])].append(dur)
    total = sum(sum(v) for v in agg.values())
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % of GPU time |', '|---|---|---|---|---|---|---|']
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f'| `{name}` | {len(v)} | {sum(v) / 1e3:.3f} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} | '
                     f'{100 * sum(v) / total:.1f} |')
    with open(out, 'a') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


def cmd_pmc(dirname, out, counters):
    agg = defaultdict(lambda: defaultdict(list))
    for path in find(dirname, 'counter_collection.csv'):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    lines = ['| kernel | dispatches | ' + ' | '.join(f'avg {c}' for c in counters) + ' |',
             '|---|---|' + '---|' * len(counters)]
    for name, d in sorted(agg.items()):
        n = max(len(v) for v in d.values())
        cells = []
        for c in counters:
            v = d.get(c, [])
            cells.append(f'{sum(v) / len(v):.4g}' if v else '-')
        lines.append(f'| `{name}` | {n} | ' + ' | '.join(cells) + ' |')
    with open(out, 'a') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


def family(name):
    for key, fam in (('basis_build_reg_kernel', 'basis_build'), ('basis_build_kernel', 'basis_build'),
                     ('gram_tiles_dma_kernel', 'gram'), ('gram_tiles4s_kernel', 'gram'), ('gram_tiles_kernel', 'gram'),
                     ('gram_mfma_kernel', 'gram'), ('gram_valu_kernel', 'gram'),
                     ('resid_quadratic_kernel', 'resid_matrix_free'), ('resid_terms_lds_kernel', 'resid_matrix_free'), ('resid_terms_kernel', 'resid_matrix_free'),
                     ('resid_kernel', 'resid'), ('gibbs_chain_kernel', 'gibbs_chain'), ('tape_gather_kernel', 'tape_gather')):
        if key in name:
            return fam
    return None


def cmd_json(fetch_dir, write_dir, out, rows, inputs, fetch_trace=None, write_trace=None):
    """Per kernel family: average raw FETCH_SIZE / WRITE_SIZE (KiB per dispatch) -> HBM bytes per launch with the
    gfx950 correction (FETCH_SIZE x 2 for wide coalesced streams), for bench.py's `roofline.traffic`.
    The Gram launches are split into the two classes the library books them under (FOKL_K_GRAM: HBM-bound, FOKL_K_GRAM_MFMA:
    fp64-MFMA-bound) with the FOKL_GRAM_TRACE file of the same run: its i-th line belongs to the i-th Gram dispatch."""
    import json
    res = {}
    for counter, dirname, trace in (('FETCH_SIZE', fetch_dir, fetch_trace), ('WRITE_SIZE', write_dir, write_trace)):
        classes = None
        if trace and os.path.exists(trace):
            classes = [line.split()[3] for line in open(trace) if line.strip()]
        rows_ = []
        for path in find(dirname, 'counter_collection.csv'):
            with open(path) as fh:
                rows_ += [r for r in csv.DictReader(fh) if r['Counter_Name'] == counter]
        rows_.sort(key=lambda r: int(r['Dispatch_Id']))
        agg = defaultdict(list)
        seen_gram = 0
        for r in rows_:
            fam = family(r['Kernel_Name'])
            if not fam:
                continue
            if fam == 'gram' and classes is not None:
                if seen_gram >= len(classes):
                    raise SystemExit(f'{trace}: fewer lines than Gram dispatches')
                fam = classes[seen_gram]
                seen_gram += 1
            agg[fam].append(float(r['Counter_Value']))
        if classes is not None and seen_gram != len(classes):
            raise SystemExit(f'{trace}: {len(classes)} lines for {seen_gram} Gram dispatches')
        for fam, v in agg.items():
            res.setdefault(fam, {})[counter + '_KiB_avg'] = sum(v) / len(v)
            res[fam]['dispatches'] = len(v)
    for fam, d in res.items():
        d['hbm_bytes_per_launch'] = 1024.0 * (2.0 * d.get('FETCH_SIZE_KiB_avg', 0.0) + d.get('WRITE_SIZE_KiB_avg', 0.0))
    payload = dict(workload=dict(rows=int(rows), inputs=int(inputs)), correction='FETCH_SIZE x 2 (gfx950), WRITE_SIZE exact',
                   command='rocprofv3 --pmc <COUNTER> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 '
                           '--no-cpu-baseline --no-microbench --no-throughput',
                   kernels=res)
    with open(out, 'w') as fh:
        json.dump(payload, fh, indent=1)
    print(json.dumps(payload, indent=1))


if __name__ == '__main__':
    mode = sys.argv[1]
    if mode == 'stats':
        cmd_stats(sys.argv[2], sys.argv[3])
    elif mode == 'json':
        cmd_json(*sys.argv[2:9])
    else:
        cmd_pmc(sys.argv[2], sys.argv[3], sys.argv[4:])
