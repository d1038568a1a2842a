"""Chunk-level data parallelism over the GPUs of one node.

Independent 30-second chunks shard embarrassingly: weights are replicated, each rank owns a
contiguous range of chunks, and there is NO data-path collective.  The only exchange is the optional
gather of the (tiny) results, done with one all_gather over RCCL (backend "nccl" on ROCm) or gloo.
The reference has no counterpart (it is single-stream, `src/lib.rs:462-464`).
"""
from typing import List, Sequence, Tuple

import numpy as np


def partition(n_chunks: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous (start, count) per rank; the first n % world ranks take one more
    (20 chunks over 8 ranks -> 3,3,3,3,2,2,2,2)."""
    q, r = divmod(n_chunks, world)
    out, s = [], 0
    for k in range(world):
        c = q + (1 if k < r else 0)
        out.append((s, c))
        s += c
    return out


def partition_balanced(costs: Sequence[float], world: int) -> List[List[int]]:
    """Chunk indices per rank when the chunks are NOT equally expensive (clips of different length; a job whose first
    pass measured decode lengths): longest-processing-time greedy -- chunks by decreasing cost, each to the least loaded
    rank, ties to the lower rank; indices inside a rank stay in chunk order.  With equal costs this degenerates to a
    round-robin deal whose counts equal partition()'s (20 over 8 -> 3,3,3,3,2,2,2,2)."""
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        k = min(range(world), key=lambda r: (load[r], len(out[r]), r))
        out[k].append(i)
        load[k] += costs[i]
    return [sorted(v) for v in out]


def scatter_order(assignment: Sequence[Sequence[int]]) -> List[int]:
    """Flattened (rank-major) chunk order of an assignment: results gathered rank by rank are put back in chunk order with
    `[gathered[pos] for pos in inverse]`, inverse[chunk] = position of that chunk in the flattened list."""
    flat = [i for v in assignment for i in v]
    inv = [0] * len(flat)
    for pos, i in enumerate(flat):
        inv[i] = pos
    return inv


def pack_results(results: Sequence[dict], ctx_len: int, slots: int) -> np.ndarray:
    """results -> int32 [slots][ctx_len + 6]: tokens | n | no_speech_exit | avg_logprob (2 x i32) |
    no_speech_prob (2 x i32), doubles bit-cast so one integer tensor carries everything."""
    buf = np.zeros((slots, ctx_len + 6), dtype=np.int32)
    for i, r in enumerate(results):
        t = r["tokens"]
        buf[i, :len(t)] = t
        buf[i, ctx_len] = len(t)
        buf[i, ctx_len + 1] = int(r.get("no_speech_exit", False))
        buf[i, ctx_len + 2:ctx_len + 6] = np.array([r["avg_logprob"], r["no_speech_prob"]], dtype=np.float64).view(np.int32)
    return buf


def unpack_results(buf: np.ndarray, ctx_len: int, count: int) -> List[dict]:
    out = []
    for i in range(count):
        n = int(buf[i, ctx_len])
        f = buf[i, ctx_len + 2:ctx_len + 6].copy().view(np.float64)
        out.append(dict(tokens=buf[i, :n].tolist(), no_speech_exit=bool(buf[i, ctx_len + 1]),
                        avg_logprob=float(f[0]), no_speech_prob=float(f[1])))
    return out


def gather_results(local: Sequence[dict], n_chunks: int, ctx_len: int, device=None,
                   assignment: Sequence[Sequence[int]] = None) -> List[dict]:
    """All ranks call this; every rank gets the results of all chunks in chunk order.
    Requires an initialised torch.distributed process group (nccl -> pass the rank's cuda device).
    assignment: chunk indices per rank when the job was dealt with partition_balanced (rank r passes the results of
    assignment[r], in that order); None = the contiguous partition()."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [c for _, c in partition(n_chunks, world)] if assignment is None else [len(a) for a in assignment]
    slots = max(max(counts), 1)
    assert len(local) == counts[rank]
    mine = torch.from_numpy(pack_results(local, ctx_len, slots))
    if device is not None:
        mine = mine.to(device)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    out: List[dict] = []
    for k, c in enumerate(counts):
        out.extend(unpack_results(bufs[k].cpu().numpy(), ctx_len, c))
    if assignment is not None:
        inv = scatter_order(assignment)
        out = [out[inv[i]] for i in range(n_chunks)]
    return out


def length_buckets(indices: Sequence[int], costs: Sequence[float], batch: int) -> List[List[int]]:
    """One rank's chunks cut into batches of <= `batch` with similar cost: a batch decodes until its LONGEST sequence ends
    (the reference's loop ends per sequence at eot, model.rs:317; in a batch the finished rows wait), so bucketing by the
    measured (or estimated) decode length is what keeps the wasted row-steps small.  Shortest first."""
    order = sorted(indices, key=lambda i: (costs[i], i))
    return [order[i:i + batch] for i in range(0, len(order), batch)]


def wasted_row_steps(batches: Sequence[Sequence[int]], steps: Sequence[int]) -> Tuple[int, int]:
    """(row-steps run, row-steps needed) when every batch runs max(steps) of its members."""
    run = sum(len(b) * max(steps[i] for i in b) for b in batches if b)
    need = sum(steps[i] for b in batches for i in b)
    return run, need
