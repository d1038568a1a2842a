"""diagnostic: which side breaks under concurrency?  T1: context A only encodes (output compared every round) while B, C only
decode; T2: A only decodes (results compared) while B, C only encode (one at a time)."""
import sys, os, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, bench
from norma_amd import config, hip, synth
import test_gpu_pool as T

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
name = "distil-large-v3"
cfg = config.preset(name); tk = common.tokens_for(name)
hm = T._varlen_weights(cfg, tk, eot_steps=bench.VARLEN_EOT_STEPS, text_steps=bench.VARLEN_TEXT_STEPS, n_calib=16, max_batch=32, seed=77)
hs = [hm] + [hip.HipWhisper(cfg, device=0, max_batch=32, share_with=hm) for _ in range(2)]
for h in hs[1:]:
    h.set_tokens(tk, tk.en, tk.transcribe)
clipsets = [np.stack([synth.synth_pcm(k + 100 * i) for k in range(32)]) for i in range(3)]
for i in range(3):
    hs[i].logmel_array(clipsets[i]); hs[i].encode()
want_dec = hs[0].decode_greedy()
want_enc = np.stack([hs[0].encoder_output(b) for b in range(32)])
stop = threading.Event()

def t1():
    bad = []
    def enc_loop():
        for r in range(ROUNDS):
            hs[0].logmel_array(clipsets[0]); hs[0].encode()
            got = np.stack([hs[0].encoder_output(b) for b in range(32)])
            for b in range(32):
                if not np.array_equal(got[b], want_enc[b]):
                    d = np.abs(got[b] - want_enc[b])
                    rows = np.nonzero(d.max(1))[0]
                    bad.append((r, b, float(d.max()), int(rows.min()), int(rows.max()), int(len(rows))))
        stop.set()
    def dec_loop(i):
        while not stop.is_set():
            hs[i].decode_greedy()
    ths = [threading.Thread(target=enc_loop)] + [threading.Thread(target=dec_loop, args=(i,)) for i in (1, 2)]
    for t in ths: t.start()
    for t in ths: t.join()
    print("T1 encoder beside two decoding contexts: rounds", ROUNDS, "mismatching (round, clip, max abs, first row, last row, rows):", len(bad), bad[:10], flush=True)

def t2():
    stop.clear()
    bad = []
    lock = threading.Lock()
    def dec_loop():
        hs[0].logmel_array(clipsets[0]); hs[0].encode(); hs[0].synchronize()
        for r in range(ROUNDS):
            got = hs[0].decode_greedy()
            for k, (g, w) in enumerate(zip(got, want_dec)):
                if g["tokens"] != w["tokens"] or g["avg_logprob"] != w["avg_logprob"] or g["no_speech_prob"] != w["no_speech_prob"]:
                    bad.append((r, k, g["tokens"] == w["tokens"], g["avg_logprob"] - w["avg_logprob"], g["no_speech_prob"] - w["no_speech_prob"]))
        stop.set()
    def enc_loop(i):
        while not stop.is_set():
            with lock:
                hs[i].logmel_array(clipsets[i]); hs[i].encode(); hs[i].synchronize()
    ths = [threading.Thread(target=dec_loop)] + [threading.Thread(target=enc_loop, args=(i,)) for i in (1, 2)]
    for t in ths: t.start()
    for t in ths: t.join()
    print("T2 decode beside encoding contexts: rounds", ROUNDS, "mismatching (round, clip, tokens equal, d avg_logprob, d no_speech):", len(bad), bad[:10], flush=True)

t1(); t2()
for h in hs[1:]: h.close()
hm.close()
