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


def test_launcher_counts_gpus_without_touching_hip(monkeypatch):
    """the launcher must not initialise HIP before it starts the ranks: the device count comes from sysfs, narrowed by *_VISIBLE_DEVICES."""
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("def main")]
    assert "import torch" not in body and "device_count" not in body and "visible_gpus()" in body
    n = bench.visible_gpus()
    assert isinstance(n, int) and n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpus() <= 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0


def test_weak_scaling_instance_hands_every_rank_its_clusters():
    """`--clusters-per-rank C` / the filled regime of the N-rank job: N C clusters, C per rank by `partition_clusters`, every copy the constraints of the
    2N-cluster multi-radius problem (so the optimum is the named problem's)."""
    import numpy as np
    import bench
    from clrs_amd.sharded import partition_clusters
    full2 = bench.weak_scaling_instance(2, 2)
    assert full2.n_clusters == 4
    full = bench.weak_scaling_instance(2, 6)
    assert full.n_clusters == 12 and full.n_free == full2.n_free and int(np.sum(full.cluster_P)) == 3 * int(np.sum(full2.cluster_P))
    parts = partition_clusters(full, 2)
    assert sorted(len(p) for p in parts) == [6, 6] and sorted(j for p in parts for j in p) == list(range(12))
    import pytest
    with pytest.raises(ValueError):
        bench.weak_scaling_instance(2, 3)


def test_bench_refuses_more_gpus_than_the_node_has_instead_of_timing_one():
    # (no GPU in the build container: device_count() is 0, and asking for 2 must exit non-zero without printing a JSON line)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300)
    import bench
    if bench.visible_gpus() < 2:
        assert r.returncode != 0 and r.stdout.strip() == "" and "needs 2 GPUs" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
