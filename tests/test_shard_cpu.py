"""N > 1 path on CPU: world_size-2 gloo processes shard chunks with no data-path collective and
gather the results; the union equals the single-process result.  The per-chunk compute here is the
CPU oracle on a reduced config (test infrastructure), the sharding/gather code is the product's."""
import os
import socket
import sys

import numpy as np
import pytest

import common
from norma_amd import shard


def test_partition_matches_the_survey_examples():
    assert [c for _, c in shard.partition(20, 8)] == [3, 3, 3, 3, 2, 2, 2, 2]
    assert [c for _, c in shard.partition(64, 8)] == [8] * 8
    p = shard.partition(5, 2)
    assert p == [(0, 3), (3, 2)]
    assert shard.partition(1, 4) == [(0, 1), (1, 0), (1, 0), (1, 0)]


def test_pack_unpack_roundtrip():
    res = [dict(tokens=[1, 2, 3], avg_logprob=-0.25, no_speech_prob=1e-5, no_speech_exit=False),
           dict(tokens=[9] * 448, avg_logprob=float("nan"), no_speech_prob=0.9, no_speech_exit=True)]
    out = shard.unpack_results(shard.pack_results(res, 448, 3), 448, 2)
    assert out[0]["tokens"] == [1, 2, 3] and out[0]["avg_logprob"] == -0.25 and out[0]["no_speech_prob"] == 1e-5
    assert out[1]["tokens"] == [9] * 448 and np.isnan(out[1]["avg_logprob"]) and out[1]["no_speech_exit"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      NORMA_ORACLE_THREADS="2")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _transcribe_range(*shard.partition(n_chunks, world)[rank])
    allr = shard.gather_results(res, n_chunks, 448)
    dist.barrier()
    if rank == 0:
        q.put(allr)
    dist.destroy_process_group()


def _transcribe_range(start, count):
    from norma_amd import assets_io, config, synth
    from oracle import oracle as O
    cfg = config.preset("test-d128")
    tk = common.tokens_for("test-d128")
    script = common.transcript_script(tk, n_segments=2, words_per_segment=3)
    om = common.build_oracle(cfg, tk, overrides=common.scripted_overrides(cfg, tk, script))
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    out = []
    for k in range(start, start + count):
        xa = om.encoder_forward(O.pcm_to_mel(synth.synth_pcm(k, 160000), filt))
        r = om.decode(xa)
        out.append(dict(tokens=r["tokens"], avg_logprob=r["avg_logprob"], no_speech_prob=r["no_speech_prob"],
                        no_speech_exit=False))
    return out


def test_world_size_2_gloo_sharding_equals_single_process():
    import torch.multiprocessing as mp
    n_chunks, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _transcribe_range(0, n_chunks)
    assert len(got) == n_chunks
    for a, b in zip(got, ref):
        assert a["tokens"] == b["tokens"]
        assert a["avg_logprob"] == b["avg_logprob"] and a["no_speech_prob"] == b["no_speech_prob"]


def test_balanced_partition_for_unequal_chunks():
    from norma_amd import shard
    # equal costs: same counts as the contiguous split of BASELINE config 4
    eq = shard.partition_balanced([1.0] * 20, 8)
    assert sorted(len(v) for v in eq) == sorted(c for _, c in shard.partition(20, 8))
    assert sorted(i for v in eq for i in v) == list(range(20))
    # unequal: one 30 s clip next to many short ones -- the long ones are spread, the maximum load is near the mean
    costs = [30, 30, 30, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 2, 2, 2, 2, 2]
    a = shard.partition_balanced(costs, 4)
    loads = [sum(costs[i] for i in v) for v in a]
    assert max(loads) <= 1.15 * sum(costs) / 4 and sorted(i for v in a for i in v) == list(range(20))
    contiguous = [sum(costs[s:s + c]) for s, c in shard.partition(20, 4)]
    assert max(loads) < max(contiguous)
    inv = shard.scatter_order(a)
    flat = [i for v in a for i in v]
    assert [flat[inv[i]] for i in range(20)] == list(range(20))
