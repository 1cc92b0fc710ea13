cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --split --skip-cpu --skip-fp64 > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err; echo "split rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/bench_split.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['config'].get('multi_gpu','')[:80])"
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_ipm.py -m gpu -x -q 2>&1 | tail -3
