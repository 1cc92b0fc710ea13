"""`python bench.py --gpus N` must produce an N-rank job by itself (round-3 review): the launcher part of bench.py, on the CPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_launcher_starts_one_process_per_rank_with_the_torchrun_environment(tmp_path):
    import bench
    out = tmp_path / "ranks"
    out.mkdir()
    code = ("import os, sys; open(os.path.join(sys.argv[1], os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')))")
    assert bench.launch_ranks(3, cmd=[sys.executable, "-c", code, str(out)], check_devices=False) == 0
    seen = {p.name: p.read_text().split() for p in out.iterdir()}
    assert sorted(seen) == ["0", "1", "2"]
    ports = set()
    for r, v in seen.items():
        assert v[0] == r and v[1] == r and v[2] == "3" and v[3] == "127.0.0.1" and v[5] == "0"
        ports.add(v[4])
    assert len(ports) == 1 and 1024 < int(next(iter(ports))) < 65536


def test_launcher_reports_a_failing_rank_and_stops_the_others():
    import bench
    code = "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(60)"
    import time
    t0 = time.time()
    assert bench.launch_ranks(2, cmd=[sys.executable, "-c", code], check_devices=False) == 7
    assert time.time() - t0 < 30


def test_bench_refuses_more_gpus_than_the_node_has_instead_of_timing_one():
    # (no GPU in the build container: device_count() is 0, and asking for 2 must exit non-zero without printing a JSON line)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and r.stdout.strip() == "" and "needs 2 GPUs" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
