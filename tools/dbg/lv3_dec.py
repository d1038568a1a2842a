import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import common
from norma_amd import assets_io, config, synth
from oracle import oracle as O
name = "large-v3"
cfg = common.make_config(name, encoder_layers=int(os.environ.get("ENC", "2")), decoder_layers=int(os.environ.get("DEC", "32")))
tk = common.tokens_for(name)
script = common.transcript_script(tk, n_segments=4, words_per_segment=7, seed=11)
want = tk.en + 23
sys.path.insert(0, 'tests')
import test_gpu_configs as T
over = T._multilingual_overrides(cfg, tk, script, want)
om, (h1,) = common.build_together(cfg, tk, overrides=over, batches=(1,), lang=-1)
filt = assets_io.mel_filters(cfg.num_mel_bins)
clip = synth.synth_pcm(9)
h1.logmel([clip]); h1.encode()
xa = om.encoder_forward(O.pcm_to_mel(clip, filt))
print("enc err", np.abs(h1.encoder_output(0) - xa).max())
h1.set_languages([want]); om.set_language(want)
got = h1.decode_greedy()[0]; ref = om.decode(xa, want_steps=True)
print("got", got["tokens"][:12]); print("ref", ref["tokens"][:12]); print("scr", ([tk.sot, want, tk.transcribe] + script)[:12])
toks = np.array([ref["tokens"][:8]], dtype=np.int32)
hh = h1.decoder_forward(toks)[0]
ho = om.decoder_forward(toks[0], xa, True)
print("hidden err per pos", np.abs(hh - ho).max(axis=1), "scale", np.abs(ho).max())
lh = h1.final_linear(hh); lo = om.final_linear(ho)
for p in range(2, 8):
    for nm, l in (("hip", lh[p]), ("ora", lo[p])):
        l = l.astype(np.float64); pr = np.exp(l - l.max()); pr /= pr.sum()
        sup = np.zeros(cfg.vocab_size, bool); sup[cfg.suppress_tokens] = True; sup[tk.no_timestamps] = True
        ts = pr[tk.no_timestamps + 1:].sum(); txt = np.where(sup[:tk.no_timestamps], 0, pr[:tk.no_timestamps]).max()
        print(p, nm, "sum_ts %.6f max_text %.6f argmax %d" % (ts, txt, int(np.argmax(l))))
print("steps", ref["steps"][:8])
