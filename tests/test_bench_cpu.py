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


def test_variable_length_job_is_dealt_by_measured_length_and_gathered_back_in_chunk_order():
    """workload varlen, rehearsed: placeholder decode lengths, the real shard.partition_balanced (longest-processing-time
    deal; the reference's loop ends per sequence at eot, model.rs:317, so chunks are NOT equally expensive) and the real
    gather with its chunk-order restoration, 3 gloo ranks."""
    p = _run(["--gpus", "3", "--workload", "varlen", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 3 and j["gathered"] == 64 and sum(j["config"]["chunks_per_rank"]) == 64
    assert j["gathered_third_tokens"] == [1000 + k for k in range(64)]     # chunk order restored after the balanced deal
    bal, con = j["balanced_loads"], j["contiguous_loads"]
    assert sum(bal) == sum(con) and max(bal) - min(bal) <= 300 and max(bal) <= max(con)


def test_rank_share_and_length_bucket_helpers():
    from norma_amd import shard
    steps = [400, 30, 35, 390, 40, 380, 45, 50]
    arrival = [[0, 1, 2, 3], [4, 5, 6, 7]]
    buckets = shard.length_buckets(list(range(8)), steps, 4)
    assert buckets == [[1, 2, 4, 6], [7, 5, 3, 0]]
    run_a, need = shard.wasted_row_steps(arrival, steps)
    run_b, need_b = shard.wasted_row_steps(buckets, steps)
    assert need == need_b == sum(steps) and run_a == 4 * 400 + 4 * 380 and run_b == 4 * 45 + 4 * 400 and run_b < run_a
