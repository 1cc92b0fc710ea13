"""rocprofv3 --pmc counter_collection csv -> per (kernel, workgroups) mean counter value per dispatch."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2]
acc = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") != name:
        continue
    k = r["Kernel_Name"].split("(")[0]
    wg = int(r.get("Grid_Size", 0)) // max(1, int(r.get("Workgroup_Size", 1)))
    acc[(k, wg)].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["Kernel", "Workgroups", "Dispatches", name + "_mean", name + "_min", name + "_max"])
for (k, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([k, wg, len(v), round(sum(v) / len(v), 3), min(v), max(v)])
