cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err; echo "bench rc=$?"; tail -5 gpurun_out/bench_r3a.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r3a.json')); 
for k in ('value','ms_per_step','speedup_vs_cpu_baseline'): print(k, d.get(k))
print(json.dumps(d.get('full_solve'))[:600]); print(json.dumps(d.get('hot_path'))[:800]); print(json.dumps(d.get('roofline_timed'))[:700]); print(json.dumps(d.get('cpu_baseline'))[:300]); print(json.dumps(d.get('roofline'))[:400])"
