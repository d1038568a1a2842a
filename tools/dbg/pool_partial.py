"""diagnostic: does a partial encoder submission (n < staging) through the pool change a clip's bits?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, bench
from norma_amd import config, pool, synth
import test_gpu_pool as T

name = "distil-large-v3"
cfg = config.preset(name); tk = common.tokens_for(name)
hm = T._varlen_weights(cfg, tk, eot_steps=bench.VARLEN_EOT_STEPS, text_steps=bench.VARLEN_TEXT_STEPS, n_calib=16, max_batch=96, seed=77)
N = 64
clips = np.stack([synth.synth_pcm(k) for k in range(N)])
want = []
for g in range(0, N, 32):
    hm.logmel_array(np.ascontiguousarray(clips[g:g + 32])); hm.encode(); want.extend(hm.decode_greedy())
enc32 = hm.encoder_output(5)   # clip 37 in a batch of 32 at row 5
for sizes in ([32, 32], [32, 11], [11, 32], [22, 21, 21]):
    order, k = [], 0
    # stream = clips 0.., cut into the given submissions by using staging = sizes one after the other
    n = sum(sizes)
    class P(pool.DecodePool):
        pass
    dp = pool.DecodePool(hm, rows=64, staging=32, check_every=16)
    cuts = list(np.cumsum([0] + sizes))
    calls = []
    def encode(first, cnt, row0, must=True, cuts=cuts, calls=calls):
        # honour the requested cut: encode only up to the next boundary
        nxt = min(c for c in cuts if c > first)
        cnt = nxt - first
        hm.logmel_array_rows(np.ascontiguousarray(clips[first:first + cnt]), row0); hm.encode_rows(row0, cnt)
        calls.append((first, cnt))
        return cnt
    # drive by hand: the stock policy asks for min(staging, left); emulate with staging = each size in turn
    got = [None] * n
    hm.pool_begin(64, 0, False)
    row = 0
    for first, cnt in zip(cuts[:-1], sizes):
        hm.logmel_array_rows(np.ascontiguousarray(clips[first:first + cnt]), 64); hm.encode_rows(64, cnt)
        for i in range(cnt):
            hm.pool_admit(64 + i, row); row += 1
    owner = list(range(n))
    left = set(range(n))
    while left:
        flags = hm.pool_step(16)
        fin = [r for r in sorted(left) if flags[r] in (1, 2)]
        if fin:
            for r, res in zip(fin, hm.pool_collect(fin)):
                got[r] = res; left.discard(r)
    bad = [(i, got[i]["tokens"] == want[i]["tokens"], got[i]["avg_logprob"] - want[i]["avg_logprob"]) for i in range(n)
           if got[i]["tokens"] != want[i]["tokens"] or got[i]["avg_logprob"] != want[i]["avg_logprob"]]
    print(sizes, "mismatches:", len(bad), bad[:6], flush=True)
# encoder output of clip 37 when it is row 5 of an 11-clip submission at row0 = 64
hm.pool_begin(64, 0, False)
hm.logmel_array_rows(np.ascontiguousarray(clips[32:43]), 64); hm.encode_rows(64, 11)
e11 = hm.encoder_output(64 + 5)
print("encoder output 11-clip submission vs 32-clip batch: max abs diff", float(np.abs(e11 - enc32).max()), "equal", bool(np.array_equal(e11, enc32)))
hm.close()
