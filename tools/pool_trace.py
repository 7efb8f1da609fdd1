"""
Lines up the noise thread's trace with the driver's (FOKL_POOL_TRACE=<file>, csrc/fokl_hostpool.cpp + engine._mark).

    FOKL_POOL_TRACE=/tmp/trace.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench --no-throughput
    python tools/pool_trace.py /tmp/trace.txt [--fit -1]

reports, for one fit of the file (default: the last): where the noise thread was idle (gap before each tape: no request
queued, or the verdict on a tentative tape outstanding), by position in the sub-stage, and what the driver was doing
during the largest stalls.
"""
import argparse
import bisect


def analyse(path, which=-1, show=12):
    fits, last = [], 'driver'
    for line in open(path):
        f = line.split()
        if f[0] == 'noise' and last == 'driver':
            fits.append(([], []))
        last = f[0]
        if f[0] == 'noise':
            fits[-1][0].append(tuple(int(v) for v in f[1:8]))
        else:
            fits[-1][1].append((int(f[1]), f[2], ' '.join(f[3:])))
    print(f"{len(fits)} fits in {path}; fit {which}:")
    noise, driver = fits[which]
    noise = [r for r in noise if r[1] > 0]                 # tapes aborted before they were started never ran
    noise.sort(key=lambda r: r[1])
    driver.sort()
    t_begin, t_end = noise[0][0], max(r[2] for r in noise)
    rewound = sum(1 for r in noise if r[5] and r[6] < 0)
    print(f"{len(noise)} tapes recorded ({rewound} of them rewound) over {(t_end - t_begin) / 1e6:.1f} ms; "
          f"recording {sum(r[2] - r[1] for r in noise) / 1e6:.1f} ms")
    edges = {tag: t for t, tag, _ in driver if tag in ('search_begin', 'run_end', 'search_end')}
    if 'search_begin' in edges and 'search_end' in edges:
        print(f"search {(edges['search_end'] - edges['search_begin']) / 1e6:.1f} ms: first recording starts "
              f"{(t_begin - edges['search_begin']) / 1e6:.2f} ms in, last one ends "
              f"{(edges['search_end'] - t_end) / 1e6:.2f} ms before the end"
              + (f" (the search loop itself {(edges['run_end'] - t_end) / 1e6:.2f} ms of them)" if 'run_end' in edges else ''))
    # the recorder is idle between the end of one recording and the start of the next
    gaps = []
    for k in range(1, len(noise)):
        prev_end = noise[k - 1][2]
        submit, start = noise[k][0], noise[k][1]
        gaps.append((start - prev_end, max(0, min(submit, start) - prev_end), k))
    print(f"idle between recordings {sum(g[0] for g in gaps) / 1e6:.1f} ms, of which the next request was not queued "
          f"yet {sum(g[1] for g in gaps) / 1e6:.1f} ms (the rest: queued, but waiting for verdicts on older tapes)")
    hist = [0] * 8
    for idle, _, _ in gaps:
        b = 0
        while b < 7 and idle > 1000 * (4 ** b):
            b += 1
        hist[b] += 1
    print("gap histogram (<=1us, 4, 16, 64, 256, 1024, 4096, more):", hist)
    print(f"largest {show} gaps, with the driver's marks inside them:")
    for idle, late, k in sorted(gaps, reverse=True)[:show]:
        a, b = noise[k - 1][2], noise[k][1]
        marks = [(t, tag, info) for t, tag, info in driver if a - 300_000 <= t <= b + 20_000]
        print(f"  before tape {k} (p1 {noise[k][4]}, tentative {noise[k][5]}): idle {idle / 1e3:.0f} us "
              f"(request late by {late / 1e3:.0f} us)")
        for t, tag, info in marks:
            if 'known=1' not in info:
                print(f"      {(t - a) / 1e3:8.1f} us  {tag} {info}")


def cost(path, which=-1, draws=2000):
    """ns per Gibbs iteration of the recorder by model size, from the tapes that ran to the end (to set beside
    tools/tape_smt_bench.cpp's stand-alone figures)."""
    fits, last = [], 'driver'
    for line in open(path):
        f = line.split()
        if f[0] == 'noise' and last == 'driver':
            fits.append([])
        last = f[0]
        if f[0] == 'noise':
            fits[-1].append(tuple(int(v) for v in f[1:8]))
    by_p = {}
    for r in fits[which]:
        if r[1] > 0 and not (r[5] and r[6] < 0):
            by_p.setdefault(r[4], []).append((r[2] - r[1]) / draws)
    print("p1: tapes, ns per iteration (min / median / mean)")
    tot = cnt = 0
    for p1 in sorted(by_p):
        v = sorted(by_p[p1])
        tot += sum(v)
        cnt += len(v)
        print(f"  {p1:4d}: {len(v):3d}  {v[0]:7.1f} {v[len(v) // 2]:7.1f} {sum(v) / len(v):7.1f}")
    print(f"all: {cnt} tapes, mean {tot / max(cnt, 1):.1f} ns per iteration")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cost', action='store_true', help='ns per iteration by model size instead of the gap report')
    ap.add_argument('trace')
    ap.add_argument('--fit', type=int, default=-1)
    ap.add_argument('--show', type=int, default=12)
    args = ap.parse_args()
    if args.cost:
        cost(args.trace, args.fit)
        return
    analyse(args.trace, args.fit, args.show)


if __name__ == '__main__':
    main()
