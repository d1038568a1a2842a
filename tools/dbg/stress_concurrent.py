"""diagnostic: do results stay bit-identical when several contexts run lockstep decodes concurrently under full load?
usage: stress_concurrent.py [threads] [rounds]"""
import sys, os, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, bench
from norma_amd import config, hip, synth
import test_gpu_pool as T

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
name = "distil-large-v3"
cfg = config.preset(name); tk = common.tokens_for(name)
hm = T._varlen_weights(cfg, tk, eot_steps=bench.VARLEN_EOT_STEPS, text_steps=bench.VARLEN_TEXT_STEPS, n_calib=16, max_batch=32, seed=77)
clips = np.stack([synth.synth_pcm(k) for k in range(32)])
hm.logmel_array(clips); hm.encode(); want = hm.decode_greedy()
hs = [hm] + [hip.HipWhisper(cfg, device=0, max_batch=32, share_with=hm) for _ in range(NT - 1)]
for h in hs[1:]:
    h.set_tokens(tk, tk.en, tk.transcribe)
bad = []
lock = threading.Lock()
def run(i):
    for r in range(ROUNDS):
        if os.environ.get("STRESS_NO_LOCK"):      # encoders of different contexts overlap freely
            hs[i].logmel_array(clips); hs[i].encode()
        else:
            with lock:
                hs[i].logmel_array(clips); hs[i].encode(); hs[i].synchronize()
        got = hs[i].decode_greedy()
        for k, (g, w) in enumerate(zip(got, want)):
            if g["tokens"] != w["tokens"] or g["avg_logprob"] != w["avg_logprob"] or g["no_speech_prob"] != w["no_speech_prob"]:
                bad.append((i, r, k, g["tokens"] == w["tokens"], g["avg_logprob"] - w["avg_logprob"], g["no_speech_prob"] - w["no_speech_prob"]))
ths = [threading.Thread(target=run, args=(i,)) for i in range(NT)]
for t in ths: t.start()
for t in ths: t.join()
print(os.environ.get("NORMA_HIP_LIB", "default lib"), "threads", NT, "rounds", ROUNDS, "decodes compared", NT * ROUNDS * 32, "mismatches", len(bad), bad[:8], flush=True)
for h in hs[1:]: h.close()
hm.close()
