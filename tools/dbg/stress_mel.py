"""diagnostic: context A computes the log-mel of the same 32 clips (PCM static in HBM) over and over and compares; contexts
B, C run decode-pool traffic without any encoder work (staging encoded once, clips re-admitted as rows free up).
usage: stress_mel.py rounds mode   (mode: pool | lockstep | idle)"""
import sys, os, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import common, bench
from norma_amd import config, hip, synth
import test_gpu_pool as T

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
MODE = sys.argv[2] if len(sys.argv) > 2 else "pool"
name = "distil-large-v3"
cfg = config.preset(name); tk = common.tokens_for(name)
hm = T._varlen_weights(cfg, tk, eot_steps=bench.VARLEN_EOT_STEPS, text_steps=bench.VARLEN_TEXT_STEPS, n_calib=16, max_batch=32, seed=77)
clips = np.stack([synth.synth_pcm(k) for k in range(32)])
pcm = torch.from_numpy(clips).to("cuda:0"); torch.cuda.synchronize()
ns = [synth.N_SAMPLES] * 32
hm.logmel_device_rows(pcm.data_ptr(), ns, synth.N_SAMPLES, 0)
ref = np.stack([hm.get_mel(b) for b in range(32)])
others = [hip.HipWhisper(cfg, device=0, max_batch=96, share_with=hm) for _ in range(2)]
for h in others:
    h.set_tokens(tk, tk.en, tk.transcribe)
    if os.environ.get("STRESS_NO_GRAPHS"):
        h.set_option(hip.NH_OPT_DECODE_GRAPHS, 0)
stop = threading.Event()
bad = []
detail = []

def mel_loop():
    for r in range(ROUNDS):
        hm.logmel_device_rows(pcm.data_ptr(), ns, synth.N_SAMPLES, 0)
        for b in range(32):
            m = hm.get_mel(b)
            if not np.array_equal(m, ref[b]):
                dm = np.abs(m - ref[b]); fr = np.nonzero(dm.max(0))[0]
                bad.append((r, b, float(dm.max()), len(fr), int((dm > 0).sum())))
                if len(detail) < 3:
                    f0 = int(fr[0])
                    bins = np.nonzero(dm[:, f0])[0]
                    # does the wrong column equal another frame's / clip's correct column?
                    match = [(bb, ff) for bb in range(32) for ff in np.nonzero((ref[bb][bins[0]] == m[bins[0], f0]))[0][:3]]
                    detail.append(dict(round=r, clip=b, frames=[int(x) for x in fr[:40]], frame=f0, bins=[int(bins.min()), int(bins.max()), len(bins)],
                                       got=[float(x) for x in m[bins[:6], f0]], want=[float(x) for x in ref[b][bins[:6], f0]],
                                       same_value_elsewhere=match[:6], clipmax_got=float(m.max()), clipmax_want=float(ref[b].max())))
    stop.set()

def pool_traffic(h):
    h.pool_begin(64, 0, False)
    h.logmel_device_rows(pcm.data_ptr(), ns, synth.N_SAMPLES, 64); h.encode_rows(64, 32); h.synchronize()
    owner = [False] * 64
    k = 0
    while not stop.is_set():
        for r in range(64):
            if not owner[r]:
                h.pool_admit(64 + (k % 32), r); owner[r] = True; k += 1
        flags = h.pool_step(16)
        fin = [r for r in range(64) if owner[r] and flags[r] in (1, 2)]
        if fin:
            h.pool_collect(fin)
            for r in fin: owner[r] = False

def lockstep_traffic(h):
    h.logmel_device_rows(pcm.data_ptr(), ns, synth.N_SAMPLES, 0); h.encode(); h.synchronize()
    while not stop.is_set():
        h.decode_greedy()

def poolsteps_traffic(h):    # all 64 rows admitted once and never collected: every step launches the whole kernel set, for ever
    h.pool_begin(64, 0, False)
    h.logmel_device_rows(pcm.data_ptr(), ns, synth.N_SAMPLES, 64); h.encode_rows(64, 32); h.synchronize()
    for r in range(64):
        h.pool_admit(64 + (r % 32), r)
    while not stop.is_set():
        h.pool_step(16)

def d2d_traffic(h):      # admissions only: device-to-device cross-K/V copies + the one-thread admit kernel, no decode step
    h.pool_begin(64, 0, False)
    h.logmel_device_rows(pcm.data_ptr(), ns, synth.N_SAMPLES, 64); h.encode_rows(64, 32); h.synchronize()
    while not stop.is_set():
        h.pool_begin(64, 0, False)
        # staging stays encoded: pool_begin clears have_enc, so re-run the (cheap) bookkeeping through a private flag
        h.logmel_device_rows(pcm.data_ptr(), ns, synth.N_SAMPLES, 64); h.encode_rows(64, 32)
        for r in range(64):
            h.pool_admit(64 + (r % 32), r)
        h.synchronize()

def torch_copy_traffic(h):   # plain device-to-device copies issued through torch, no norma context involved
    a = torch.empty(64, 1500 * 1280, dtype=torch.float16, device="cuda:0"); b = torch.empty_like(a)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        while not stop.is_set():
            for r in range(64):
                b[r].copy_(a[(r * 7) % 64])
            st.synchronize()

def lockstep_rows(rows):
    def f(h):
        for r0 in range(0, rows, 32):
            n = min(32, rows - r0)
            h.logmel_device_rows(pcm.data_ptr(), ns[:n], synth.N_SAMPLES, r0); h.encode_rows(r0, n)
        h.synchronize()
        while not stop.is_set():
            h.decode_greedy()
    return f

if MODE.startswith("lockstep") and MODE != "lockstep":
    globals()["lockstep_n"] = lockstep_rows(int(MODE[len("lockstep"):]))
fn = {"poolsteps": poolsteps_traffic, "pool": pool_traffic, MODE if MODE.startswith("lockstep") and MODE != "lockstep" else "_": globals().get("lockstep_n"), "lockstep": lockstep_traffic, "d2d": d2d_traffic, "tcopy": torch_copy_traffic}.get(MODE)
ths = [threading.Thread(target=mel_loop)] + ([threading.Thread(target=fn, args=(h,)) for h in others] if fn else [])
for t in ths: t.start()
for t in ths: t.join()
print("mode", MODE, "rounds", ROUNDS, "clip-mels compared", ROUNDS * 32, "differing (round, clip, max abs, frames, elements):", len(bad), bad[:10], flush=True)
for dd in detail: print(dd, flush=True)
for h in others: h.close()
hm.close()
