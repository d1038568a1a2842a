"""diagnostic: three decode pools in flight; every encoder submission's mel and encoder output are compared with what the
same clip gives alone, every decode result too -- which stage is the first to differ when results differ under load?"""
import sys, os, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, bench
from norma_amd import config, hip, pool, synth
import test_gpu_pool as T

NPOOL = int(sys.argv[1]) if len(sys.argv) > 1 else 3
PER = int(sys.argv[2]) if len(sys.argv) > 2 else 192
CHECK = (sys.argv[3] if len(sys.argv) > 3 else "mel,xa").split(",")
KEEP = len(sys.argv) > 4 and sys.argv[4] == "keep"
kept = []
DEVPCM = len(sys.argv) > 4 and sys.argv[4] == "devpcm"   # PCM resident in HBM, never rewritten: no copy of any kind in front of log-mel
if DEVPCM:
    import torch
    pcm_all = None
name, JOB = "distil-large-v3", 64
cfg = config.preset(name); tk = common.tokens_for(name)
hm = T._varlen_weights(cfg, tk, eot_steps=bench.VARLEN_EOT_STEPS, text_steps=bench.VARLEN_TEXT_STEPS, n_calib=16, max_batch=32, seed=77)
clips = np.stack([synth.synth_pcm(k) for k in range(JOB)])
want, mel_ref, xa_ref = [], [], []
for g in range(0, JOB, 32):
    hm.logmel_array(np.ascontiguousarray(clips[g:g + 32])); hm.encode()
    want.extend(hm.decode_greedy())
    mel_ref.extend(hm.get_mel(b) for b in range(32)); xa_ref.extend(hm.encoder_output(b) for b in range(32))
if DEVPCM:
    pcm_all = torch.from_numpy(clips).to("cuda:0"); torch.cuda.synchronize()
hps = [hip.HipWhisper(cfg, device=0, max_batch=96, share_with=hm) for _ in range(NPOOL)]
for h in hps:
    h.set_tokens(tk, tk.en, tk.transcribe)
lock = threading.Lock()
events, bad = [], []

def one(i):
    hp = hps[i]
    first = i * PER
    def encode(f, n, row0, must):
        if not lock.acquire(blocking=must):
            return False
        try:
            ids = [(first + f + k) % JOB for k in range(n)]
            part = np.ascontiguousarray(clips[ids])
            if KEEP:
                kept.append(part)      # the host buffer outlives the call
            if DEVPCM:
                assert ids == list(range(ids[0], ids[0] + n))
                hp.logmel_device_rows(pcm_all[ids[0]].data_ptr(), [synth.N_SAMPLES] * n, synth.N_SAMPLES, row0)
            else:
                hp.logmel_array_rows(part, row0)
            hp.encode_rows(row0, n); hp.synchronize()
            for k, c in enumerate(ids):
                if "mel" in CHECK:
                    m = hp.get_mel(row0 + k)
                    if not np.array_equal(m, mel_ref[c]):
                        dm = np.abs(m - mel_ref[c]); fr = np.nonzero(dm.max(0))[0]
                        events.append((i, f, k, "mel", float(dm.max()), int(fr.min()), int(fr.max()), len(fr), int((dm > 0).sum())))
                if "xa" in CHECK:
                    x = hp.encoder_output(row0 + k)
                    if not np.array_equal(x, xa_ref[c]):
                        d = np.abs(x - xa_ref[c]); rows = np.nonzero(d.max(1))[0]
                        events.append((i, f, k, "xa", float(d.max()), int(rows.min()), int(rows.max()), len(rows)))
        finally:
            lock.release()
        return True
    got = pool.DecodePool(hp, rows=64, staging=32, check_every=16).run(PER, encode)
    for j, g in enumerate(got):
        w = want[(first + j) % JOB]
        if g["tokens"] != w["tokens"] or g["avg_logprob"] != w["avg_logprob"] or g["no_speech_prob"] != w["no_speech_prob"]:
            bad.append((i, j, g["tokens"] == w["tokens"], g["avg_logprob"] - w["avg_logprob"], g["no_speech_prob"] - w["no_speech_prob"]))
ths = [threading.Thread(target=one, args=(i,)) for i in range(NPOOL)]
for t in ths: t.start()
for t in ths: t.join()
print("pools", NPOOL, "clips per pool", PER, "checked", CHECK, "keep host buffers", KEEP)
print("encoder-side events (pool, first clip of the submission, index in it, stage, max abs[, first row, last row, rows]):", len(events), events[:16])
print("decode results that differ (pool, clip, tokens equal, d avg_logprob, d no_speech):", len(bad), bad[:16], flush=True)
for h in hps: h.close()
hm.close()
