import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t = sorted((r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:48], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
t.sort(key=lambda x: x[1])
n = len(t); last = t[int(n * 0.4):]     # the timed steps (roughly)
span = (last[-1][2] - last[0][1]) / 1e3; busy = sum(e - s for _, s, e in last) / 1e3
print("kernels %d, span %.0f us, busy %.0f us (%.0f%%)" % (len(last), span, busy, 100 * busy / span))
agg = collections.Counter(); cnt = collections.Counter()
for k, s, e in last: agg[k] += (e - s) / 1e3; cnt[k] += 1
for k, v in agg.most_common(22): print("%9.0f us  x%-5d %s" % (v, cnt[k], k))
