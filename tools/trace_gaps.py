#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace csv: busy time, span, idle gaps, per-kernel totals."""
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
# keep only the last decode: after the largest gap region... simply take all
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
gaps_sorted = sorted(gaps, reverse=True)
print(f"kernels {len(rows)} busy {busy/1e6:.2f} ms span {span/1e6:.2f} ms; gaps: total {sum(g for g in gaps if g > 0)/1e6:.2f} ms, "
      f"median {sorted(gaps)[len(gaps)//2]/1e3:.1f} us, top {[round(g/1e3) for g in gaps_sorted[:8]]} us")
acc = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    acc[n][0] += 1
    acc[n][1] += e - s
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{n:62s} {c:6d} {t/1e6:9.2f} ms {t/c/1e3:9.1f} us")
