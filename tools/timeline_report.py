#!/usr/bin/env python
"""Busy / gap breakdown of gpurun_out/kt_tail.csv (tools/timeline.sh)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/kt_tail.csv')))
def short(n):
    m = re.match(r'(?:void )?(?:lq::)?(\w+)', n)
    k = m.group(1) if m else n[:30]
    if k == 'k_nn_scan':                       # k_nn_scan<System, S form, TRI[, PATCH]>: the in-wave (triangular) scan is its own line
        a = re.search(r'k_nn_scan<(.*?)>\(', n)
        args = [x.strip() for x in a.group(1).split(',')] if a else []
        if len(args) >= 3 and args[2] == 'true':
            k += '<TRI>'
    return k
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
ev = ev[300:3800]
tot = ev[-1][1] - ev[0][0]
busy = collections.Counter(); cnt = collections.Counter(); gap = collections.Counter()
prev = None
for a, b, k in ev:
    busy[k] += b - a; cnt[k] += 1
    if prev is not None: gap[k] += max(0, a - prev)
    prev = b
waves = max(1, cnt.get('k_nn_scan', 1))        # one tree scan per wave
print("window %.1f ms, %d dispatches, %d waves, %.1f us/wave" % (tot / 1e6, len(ev), waves, tot / 1e3 / waves))
for k, v in busy.most_common():
    print("%-18s n/wave %5.2f  busy %5.1f%%  avg %6.1f us  gap-before avg %5.1f us (%4.1f%%)" % (
        k, cnt[k] / waves, 100 * v / tot, v / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3, 100 * gap[k] / tot))
print("total busy %.1f%%" % (100 * sum(busy.values()) / tot))
