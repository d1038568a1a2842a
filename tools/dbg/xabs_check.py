"""diagnostic: NH_OPT_ABSORBED_XATTN 0 / 1 / 2 side by side: teacher-forced hidden states and greedy decodes"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from norma_amd import config, hip, synth
name = sys.argv[1] if len(sys.argv) > 1 else "distil-large-v3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = config.preset(name); tk = common.tokens_for(name)
script = common.transcript_script(tk, n_segments=3, words_per_segment=6, seed=5)
over = common.scripted_overrides(cfg, tk, script)
hm = common.build_hip(cfg, tk, overrides=over, max_batch=B)
clips = np.stack([synth.synth_pcm(k) for k in range(B)])
hm.logmel_array(clips); hm.encode()
toks = np.array([[tk.sot, tk.en, tk.transcribe] + script[:9]] * B, dtype=np.int32)
hid, dec, tms = {}, {}, {}
for opt in (0, 1, 2):
    hm.set_option(hip.NH_OPT_ABSORBED_XATTN, opt)
    hid[opt] = hm.decoder_forward(toks)
    t0 = time.perf_counter(); dec[opt] = hm.decode_greedy(); tms[opt] = time.perf_counter() - t0
print(name, "rows", B)
for a, b in ((0, 1), (1, 2), (0, 2)):
    print(f"hidden max |opt{a} - opt{b}| = {np.abs(hid[a] - hid[b]).max():.3e}   rms {np.sqrt(((hid[a] - hid[b]) ** 2).mean()):.3e}   nan: {np.isnan(hid[b]).any()}")
for opt in (0, 1, 2):
    same = all(x["tokens"] == y["tokens"] for x, y in zip(dec[opt], dec[0]))
    dl = max(abs(x["avg_logprob"] - y["avg_logprob"]) for x, y in zip(dec[opt], dec[0]))
    print(f"opt {opt}: tokens == opt0: {same}  max |d avg_logprob| {dl:.2e}  decode {tms[opt]*1e3:.1f} ms  script ok: {dec[opt][0]['tokens'][3:] == script}")
hm.close()
