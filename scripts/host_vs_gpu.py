"""From a rocprofv3 --hip-trace --kernel-trace run of scripts/mw_iter_profile.py: for the kernels of the main stream, how far ahead of the GPU the host's
launch call was (kernel start - launch call end), and the idle time of the stream in front of the kernel (kernel start - previous kernel end)."""
import csv, sys, collections, glob, os
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
ht = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0]
K = list(csv.DictReader(open(kt)))
H = [r for r in csv.DictReader(open(ht)) if "LaunchKernel" in r["Function"]]
K.sort(key=lambda r: int(r["Start_Timestamp"]))
by_corr = {r["Correlation_Id"]: r for r in H}
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
last_end = {}
rows = collections.defaultdict(list)
for r in K[len(K) // 2:]:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    h = by_corr.get(r["Correlation_Id"])
    lead = (s - int(h["End_Timestamp"])) / 1e3 if h else float("nan")
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    rows[(q, nm(r))].append((lead, gap))
print("%-4s %-30s %8s %12s %12s" % ("q", "kernel", "calls", "lead us", "gap us"))
for (q, n), v in sorted(rows.items(), key=lambda kv: -len(kv[1])):
    lead = sorted(x[0] for x in v if x[0] == x[0]); gap = sorted(x[1] for x in v if x[1] == x[1])
    if len(v) < 20: continue
    print("%-4s %-30s %8d %12.1f %12.1f" % (q, n, len(v), lead[len(lead) // 2] if lead else -1, gap[len(gap) // 2] if gap else -1))
