"""CPU: bench.py's own rank launcher (`--gpus N` without an external launcher) and the strong-scaling job split, rehearsed
with `--dry-run` (gloo rendezvous, real shard.partition + gather_results, no GPU work).  The reference has no counterpart:
it is single-stream (src/lib.rs:462-464); the N > 1 path is new-build work named by BASELINE.json's north_star."""
import json
import os
import subprocess
import sys

import pytest

import common

BENCH = os.path.join(common.ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=common.ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_rank_environments_describe_one_rank_per_gpu():
    sys.path.insert(0, common.ROOT)
    import bench
    envs = bench.rank_environments(4, base_env={"PATH": "/usr/bin"}, port=29999)
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


@pytest.mark.parametrize("n,split", [(2, [10, 10]), (3, [7, 7, 6])])
def test_gpus_flag_spawns_that_many_ranks_and_the_20_chunk_job_is_gathered_in_order(n, split):
    p = _run(["--gpus", str(n), "--workload", "longform20", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["scaling"] == "strong" and j["dry_run"] is True
    assert j["config"]["chunks_per_rank"] == split and j["gathered"] == 20
    assert j["gathered_third_tokens"] == [1000 + k for k in range(20)]   # chunk order survives partition + gather


def test_weak_default_workload_reports_the_ranks_that_ran():
    p = _run(["--gpus", "2", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["chunks"] == 64


def test_gpus_flag_must_agree_with_an_external_launcher():
    p = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
