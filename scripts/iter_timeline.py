"""Per-iteration timeline from a rocprofv3 kernel-trace csv of scripts/mw_iter_profile.py: takes the iterations of the LAST solve
(an iteration starts at k_mwi_dots(mode 1) on the side stream = the first kernel after the host's per-iteration sync), averages over them
kernel by kernel in launch order: queue, start offset from the iteration's first kernel, duration, idle gap on its own queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].replace("clrs::", "").replace("void ", "")
# split into iterations at gaps > 15 us between the end of everything and the next start (the host sync + copy)
its, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 12000 and cur:
        its.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, e)
if cur: its.append(cur)
lens = collections.Counter(len(i) for i in its)
L = lens.most_common(1)[0][0]
its = [i for i in its if len(i) == L]
its = its[len(its) // 2:]
print("iterations used: %d of %d launches each" % (len(its), L))
tot = []
for k in range(L):
    off = [int(i[k]["Start_Timestamp"]) - int(i[0]["Start_Timestamp"]) for i in its]
    dur = [int(i[k]["End_Timestamp"]) - int(i[k]["Start_Timestamp"]) for i in its]
    q = its[0][k].get("Queue_Id", "?")
    grid = its[0][k].get("Grid_Size_X", "?") if "Grid_Size_X" in its[0][k] else its[0][k].get("Grid_Size", "?")
    wg = its[0][k].get("Workgroup_Size_X", its[0][k].get("Workgroup_Size", "?"))
    print("%2d q%-3s %-34s grid %-7s wg %-5s start %8.1f us  dur %7.1f us" % (k, q, nm(its[0][k])[:34], grid, wg, sum(off) / len(off) / 1e3, sum(dur) / len(dur) / 1e3))
span = [max(int(r["End_Timestamp"]) for r in i) - int(i[0]["Start_Timestamp"]) for i in its]
per = [int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"]) for a, b in zip(its, its[1:])]
print("iteration span (first start to last end) %.1f us; start-to-start %.1f us; sum of durations %.1f us" %
      (sum(span) / len(span) / 1e3, (sum(per) / len(per) / 1e3) if per else 0, sum(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in i) for i in its) / len(its) / 1e3))
