"""rocprofv3 kernel trace csv -> per (kernel, grid size) call count and average / min / max duration (us).
bench.py launches the same kernels on two workloads (the named 2-cluster problem and the many-cluster roofline
instance); `--stats` averages them together, this keeps them apart."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
    acc[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Kernel", "Workgroups", "Calls", "AverageUs", "MinUs", "MaxUs"])
for (name, grid), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, grid, len(d), round(sum(d) / len(d) / 1e3, 3), round(min(d) / 1e3, 3), round(max(d) / 1e3, 3)])
