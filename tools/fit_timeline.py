"""Chronological view of one fit from a FOKL_POOL_TRACE file: the walker's tapes (start, length, size, tentative / verdict)
merged with the driver's marks, one line per event, times in microseconds from search_begin.

    python tools/fit_timeline.py trace.txt [--fit K] [--from US] [--to US]
"""
import argparse


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('path')
    ap.add_argument('--fit', type=int, default=-2)
    ap.add_argument('--from', dest='lo', type=float, default=0.0)
    ap.add_argument('--to', dest='hi', type=float, default=1e12)
    ap.add_argument('--tapes', action='store_true', help='every tape (default: runs of back-to-back tapes are folded)')
    args = ap.parse_args()
    noise, marks = [], []
    for line in open(args.path):
        f = line.split()
        if f[0] == 'noise':
            noise.append(tuple(int(v) for v in f[1:8]))
        else:
            marks.append((int(f[1]), f[2], ' '.join(f[3:])))
    begins = [t for t, tag, _ in marks if tag == 'search_begin']
    ends = [t for t, tag, _ in marks if tag == 'search_end']
    b, e = begins[args.fit], [x for x in ends if x > begins[args.fit]][0]
    ev = [(t, 'D', tag, info) for t, tag, info in marks if b <= t <= e]
    tapes = sorted(r for r in noise if r[1] and b <= r[1] <= e)
    print(f"fit {args.fit}: {(e - b) / 1e3:.0f} us, {len(tapes)} tapes, walking {sum(r[2] - r[1] for r in tapes) / 1e3:.0f} us")
    # fold runs of tapes with gaps < 5 us
    run = None
    for r in tapes:
        sub, st, rec, seen, p1, tent, verdict = r
        if args.tapes or run is None or st - run[1] > 5000:
            if run is not None:
                ev.append((run[0], 'W', f"{run[2]} tapes p1 {run[3]}..{run[4]} rewound {run[5]}", f"until {(run[1] - b) / 1e3:.0f}"))
            run = [st, rec, 1, p1, p1, int(verdict < 0)]
        else:
            run[1], run[2], run[4], run[5] = rec, run[2] + 1, p1, run[5] + int(verdict < 0)
    if run is not None:
        ev.append((run[0], 'W', f"{run[2]} tapes p1 {run[3]}..{run[4]} rewound {run[5]}", f"until {(run[1] - b) / 1e3:.0f}"))
    last_w_end = None
    for t, kind, tag, info in sorted(ev):
        us = (t - b) / 1e3
        if args.lo <= us <= args.hi:
            print(f"{us:9.0f} {kind} {tag} {info}")


if __name__ == '__main__':
    main()
