"""counter_collection.csv + kernel_trace.csv of scripts/mw_iter_profile.py -> per-kernel means (SQ_INSTS_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES,
duration, workgroups, issue utilisation of the occupied compute units) and the json bench.py reads for `roofline_timed`."""
import collections, csv, json, sys
pmc, trace, out_csv, out_json, tag = sys.argv[1:6]
def nm(r): return r["Kernel_Name"].split("(")[0].replace("clrs::", "").replace("void ", "")
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
wgs = {}
for r in csv.DictReader(open(pmc)):
    k = nm(r)
    cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    wgs[k] = int(r.get("Grid_Size", 0)) // max(1, int(r.get("Workgroup_Size", 1)))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[nm(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k in sorted(cnt, key=lambda k: -sum(dur.get(k, [0]))):
    c = cnt[k]
    m = {n: sum(v) / len(v) for n, v in c.items()}
    d = sum(dur[k]) / max(len(dur[k]), 1) / 1e3 if k in dur else float("nan")
    if k.startswith("k_mw_factor_pipe"):      # grid of 64 blocks per 8 clusters, of which 8 per cluster work (4 stages + 4 for the inverse; the others return at once): 2 clusters here
        wgs[k] = 16
    if k.startswith("k_mw_potrf_q_pipe"):     # the first 64 blocks are one matrix' roles (4 stages + 4 for the inverse work), the blocks behind them the riding forward products (2 per cluster)
        wgs[k] = 8 + max(0, wgs.get(k, 64) - 64)
    cus = min(wgs.get(k, 0), 256)
    # share of the fp64 VALU issue slots (4 cycles per wave instruction, 4 SIMDs per CU) of the compute units the kernel occupies, over its duration at 2.4 GHz
    util = m.get("SQ_INSTS_VALU", 0) * 4.0 / (max(cus, 1) * 4 * d * 1e-6 * 2.4e9) if d == d and d > 0 else float("nan")
    rows.append((k, len(next(iter(c.values()))), wgs.get(k, 0), m.get("SQ_INSTS_VALU", 0), m.get("SQ_BUSY_CYCLES", 0), m.get("SQ_WAVE_CYCLES", 0), d, util))
with open(out_csv, "w") as fh:
    fh.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python3 scripts/mw_iter_profile.py ce_8_15 2 (counters) and --kernel-trace of the same command (durations)\n")
    fh.write("# issue_utilisation = SQ_INSTS_VALU x 4 cycles / (min(workgroups, 256) CUs x 4 SIMDs x duration x 2.4 GHz)\n")
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "workgroups", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "duration_us", "issue_utilisation"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), round(r[4], 1), round(r[5], 1), round(r[6], 2), round(r[7], 4)])
d = {r[0]: r for r in rows}
fk = "k_mw_factor_pipe<5>" if "k_mw_factor_pipe<5>" in d else "k_mw_factor<5>"
f, l = d[fk], d["k_mw_linvb<5, 2>"]
ent = lambda r: {"SQ_INSTS_VALU": r[3], "SQ_BUSY_CYCLES": r[4], "workgroups": r[2], "duration_us": r[6], "issue_utilisation": r[7]}
json.dump({"source": f"profiles/r05/{tag}_pmc_iter_sq_counters.csv (rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python3 scripts/mw_iter_profile.py ce_8_15 2; durations from --kernel-trace of the same command)",
           "limbs": 5, "factor_limbs": "4 in 55 of the 56 iterations of the solve (mixed-precision refinement), 5 in the last", "k_mw_factor": dict(ent(f), kernel=fk.split("<")[0], share_of_stage=f[6] / (f[6] + l[6])),
           "k_mw_potrf_q": ent(d["k_mw_potrf_q<5>"]) if "k_mw_potrf_q<5>" in d else ent(d["k_mw_potrf_q_pipe<5>"]),
           "k_mw_potrf_x": ent(d["k_mw_potrf_x<5>"])},
          open(out_json, "w"), indent=1)
