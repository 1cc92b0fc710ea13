"""Per-iteration timeline from a rocprofv3 kernel-trace csv of scripts/mw_iter_profile.py.  An iteration starts with k_mw_potrf_x (the
first kernel of the decomposition on the context's stream); the iterations of the last solve are averaged kernel by kernel in start
order: queue, start offset from the iteration's k_mw_potrf_x, duration.  With the host one iteration ahead of the records, the time
between two k_mw_potrf_x starts is the iteration time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].replace("clrs::", "").replace("void ", "")
starts = [i for i, r in enumerate(rows) if nm(r).startswith("k_mw_potrf_x")]
its = [rows[a:b] for a, b in zip(starts, starts[1:])]
lens = collections.Counter(len(i) for i in its)
L = lens.most_common(1)[0][0]
per_all = [int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"]) for a, b in zip(its, its[1:]) if len(a) == L]
its = [i for i in its if len(i) == L]
its = its[len(its) // 2:]
# kernels of one iteration can interleave differently from iteration to iteration across streams: align by (queue, order within queue)
def keyed(it):
    cnt = collections.Counter(); out = {}
    for r in it:
        q = r.get("Queue_Id", "?"); cnt[q] += 1
        out[(q, cnt[q])] = r
    return out
ks = [keyed(i) for i in its]
common = [k for k in ks[0] if all(k in d for d in ks)]
order = sorted(common, key=lambda k: sum(int(d[k]["Start_Timestamp"]) - int(i[0]["Start_Timestamp"]) for d, i in zip(ks, its)))
print("iterations used: %d of %d launches each" % (len(its), L))
for k in order:
    off = [int(d[k]["Start_Timestamp"]) - int(i[0]["Start_Timestamp"]) for d, i in zip(ks, its)]
    dur = [int(d[k]["End_Timestamp"]) - int(d[k]["Start_Timestamp"]) for d in ks]
    r0 = ks[0][k]
    print("q%-3s %-34s grid %-7s wg %-5s start %8.1f us  dur %7.1f us  end %8.1f" % (k[0], nm(r0)[:34], r0.get("Grid_Size_X", "?"), r0.get("Workgroup_Size_X", "?"),
          sum(off) / len(off) / 1e3, sum(dur) / len(dur) / 1e3, (sum(off) + sum(dur)) / len(off) / 1e3))
per_all.sort()
print("iteration time (k_mw_potrf_x start to start): median %.1f us, mean %.1f us over %d; sum of kernel durations %.1f us" %
      (per_all[len(per_all) // 2] / 1e3, sum(per_all) / len(per_all) / 1e3, len(per_all),
       sum(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in i) for i in its) / len(its) / 1e3))
