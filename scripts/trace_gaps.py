"""Summarise a rocprofv3 kernel trace csv: per-kernel average duration and the idle gap before each kernel."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[len(rows) // 2:]   # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in tail:
    name = r["Kernel_Name"].split("(")[0].replace("clrs::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev_end is not None: gap[name].append(s - prev_end)
    prev_end = e
tot = 0
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[k], gap[k] or [0]
    med = sorted(g)[len(g) // 2]
    print(f"{k[:40]:40s} n={len(d):5d} avg dur {sum(d)/len(d)/1e3:7.2f} us  median gap before {med/1e3:7.2f} us")
