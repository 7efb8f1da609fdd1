"""Development aid: CPU load per L3 domain over one second (are the neighbours on this host busy?)."""
import time, glob
def snap():
    out = {}
    for line in open('/proc/stat'):
        if line.startswith('cpu') and line[3].isdigit():
            f = line.split(); v = list(map(int, f[1:9])); out[int(f[0][3:])] = (sum(v) - v[3] - v[4], sum(v))
    return out
a = snap(); time.sleep(1.0); b = snap()
busy = {c: (b[c][0] - a[c][0]) / max(1, b[c][1] - a[c][1]) for c in a}
doms = {}
for c in busy:
    try: key = open(f'/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list').read().strip()
    except OSError: key = '?'
    doms.setdefault(key, []).append(busy[c])
for k, v in sorted(doms.items(), key=lambda kv: int(kv[0].split('-')[0].split(',')[0])):
    print(k, 'mean busy %.2f max %.2f' % (sum(v) / len(v), max(v)))
import os; print('allowed', len(os.sched_getaffinity(0)))
