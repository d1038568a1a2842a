"""diagnostic: the fed pool case rows=5 batch=7 encoders=3, repeated; which clip / field differs"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from norma_amd import config, hip, pool, synth
import test_gpu_pool as T
rows, batch, n_enc = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (5, 7, 3)
name, N = "test-d128", 31
cfg = config.preset(name); tk = common.tokens_for(name)
hm = T._varlen_weights(cfg, tk, eot_steps=[2, 5, 9, 14, 22], text_steps=40, n_calib=8, max_batch=N)
clips = np.stack([synth.synth_pcm(k) for k in range(N)])
hm.logmel_array(clips); hm.encode(); want = hm.decode_greedy()
hp = hip.HipWhisper(cfg, device=0, max_batch=rows + 1, share_with=hm)
encs = [hip.HipWhisper(cfg, device=0, max_batch=batch, share_with=hm) for _ in range(n_enc)]
for h in [hp] + encs: h.set_tokens(tk, tk.en, tk.transcribe)
log = []
def encode(i, first, n):
    encs[i].logmel_array(np.ascontiguousarray(clips[first:first + n])); encs[i].encode()
    if os.environ.get("FED_SYNC"): encs[i].synchronize()
    log.append((i, first, n))
for rep in range(int(os.environ.get("FED_REPS", "6"))):
    got = pool.FedDecodePool(hp, encs, rows=rows, batch=batch, check_every=3).run(N, encode)
    bad = [(i, g["tokens"] == w["tokens"], len(g["tokens"]), len(w["tokens"]), g["avg_logprob"] - w["avg_logprob"], g["no_speech_prob"] - w["no_speech_prob"]) for i, (g, w) in enumerate(zip(got, want)) if not T._same(g, w)]
    print("rep", rep, "bad", bad, flush=True)
