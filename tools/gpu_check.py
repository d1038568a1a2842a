"""Stage-by-stage diagnostic of the HIP path against the CPU oracle (run on the GPU box).
Not a test: prints error magnitudes per stage for a few reduced configs so a wrong kernel can be
located quickly.  Usage: python tools/gpu_check.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from norma_amd import assets_io, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

report = {}


def stat(name, a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    r = dict(max_abs=float(d.max()), mean_abs=float(d.mean()), ref_absmax=float(np.abs(b).max()),
             ref_rms=float(np.sqrt((b * b).mean())), nan=int(np.isnan(a).sum()))
    report[name] = r
    print(f"{name:40s} max|d|={r['max_abs']:.3e} mean|d|={r['mean_abs']:.3e} ref_rms={r['ref_rms']:.3e} nan={r['nan']}", flush=True)
    return r


def run(name, enc, dec, B=2, base="test-d128"):
    print(f"=== {name}: {base} enc={enc} dec={dec} B={B}", flush=True)
    cfg = common.make_config(base, encoder_layers=enc, decoder_layers=dec)
    tk = common.tokens_for(base)
    om = common.build_oracle(cfg, tk)
    hm = common.build_hip(cfg, tk, max_batch=B)
    filt = assets_io.mel_filters(cfg.num_mel_bins)
    clips = [synth.synth_pcm(k) for k in range(B)]
    if B > 1:
        clips[1] = clips[1][:400000]
    t0 = time.time(); hm.logmel(clips); hm.synchronize(); print("  logmel %.3fs" % (time.time() - t0), flush=True)
    mels = []
    for b in range(B):
        ref = O.pcm_to_mel(clips[b], filt)[:, :3000]
        got = hm.get_mel(b)
        stat(f"{name}/mel[{b}]", got, ref)
        mels.append(ref)
    # encoder from the ORACLE mel (isolates the encoder) and from the device mel
    hm.set_mel(np.stack(mels)); hm.encode()
    xas = []
    for b in range(B):
        xa = om.encoder_forward(mels[b]); xas.append(xa)
        stat(f"{name}/enc_from_oracle_mel[{b}]", hm.encoder_output(b), xa)
    hm.logmel(clips); hm.encode()
    for b in range(B):
        stat(f"{name}/enc_end_to_end[{b}]", hm.encoder_output(b), xas[b])
    # teacher-forced decoder
    toks = np.array([[tk.sot, tk.en, tk.transcribe, tk.zero_sec, 100, 2000, 30000, tk.zero_sec + 40]] * B, dtype=np.int32)
    toks[-1, 4] = 777
    hid = hm.decoder_forward(toks)
    for b in range(B):
        ref = om.decoder_forward(toks[b], xas[b], True)
        stat(f"{name}/dec_hidden[{b}]", hid[b], ref)
        lg = hm.final_linear(hid[b][-2:])
        rl = om.final_linear(ref[-2:])
        stat(f"{name}/logits[{b}]", lg, rl)
    # greedy decode
    res = hm.decode_greedy()
    for b in range(B):
        r = om.decode(xas[b], use_kv_cache=True, want_steps=True)
        same = r["tokens"] == res[b]["tokens"]
        first = next((i for i, (x, y) in enumerate(zip(r["tokens"], res[b]["tokens"])) if x != y), None)
        print(f"  decode[{b}] identical={same} n_ref={len(r['tokens'])} n_hip={len(res[b]['tokens'])} first_diff={first}"
              f" nsp ref={r['no_speech_prob']:.3e} hip={res[b]['no_speech_prob']:.3e} alp ref={r['avg_logprob']} hip={res[b]['avg_logprob']}", flush=True)
        if first is not None:
            print("    ref", r["tokens"][max(0, first - 3):first + 4], "hip", res[b]["tokens"][max(0, first - 3):first + 4])
            st = r["steps"][max(0, first - 3 - 3):first - 3 + 2]
            print("    oracle step probs (p_next, second, sum_ts, max_text):", st.tolist())
        report[f"{name}/decode[{b}]"] = dict(identical=bool(same), first_diff=first, n_ref=len(r["tokens"]))
    print("  timings", hm.timings(), flush=True)
    hm.close(); om.close()


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else None
    which = sys.argv[2:] if len(sys.argv) > 2 else ["a", "b", "c", "d"]
    if "a" in which: run("stem", 0, 1)
    if "b" in which: run("one_layer", 1, 1)
    if "c" in which: run("two_layer", 2, 2)
    if "d" in which: run("mel128", 2, 2, B=3, base="test-d256-mel128")
    if "e" in which: run("tiny.en", 4, 4, B=2, base="tiny.en")
    if out:
        with open(out, "w") as f:
            json.dump(report, f, indent=1)
