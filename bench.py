#!/usr/bin/env python3
"""bench.py -- audio-sec / wall-sec (xRT) of the MI355X Whisper hot path.

One "step" = one pass of the hot path (log-mel -> encoder -> cross-K/V -> batched greedy decode -> token ids on
the host) over one batch of synthetic 30-second clips, PCM already resident in HBM.

Workloads (`--workload`):
  b32         (default, the headline) BASELINE.json configs[2]: distil-large-v3, fp16 storage / fp32 accumulate,
              32 clips per GPU.  N > 1: every rank runs its own 32 distinct clips -- weak scaling, no data-path
              collective (chunks are independent).
  longform20  BASELINE.json configs[3]: one 10-minute clip (9 600 000 samples) = 20 chunks, split over the ranks with
              norma_amd.shard.partition (8 ranks: 3,3,3,3,2,2,2,2); every rank transcribes its chunks as one batch and
              the results are gathered on all ranks over RCCL inside the timed region -- strong scaling.
  lv3b64      BASELINE.json configs[4]: large-v3 multilingual, 64 chunks split over the ranks (8 per GPU at N = 8),
              language detection + timestamp decoding, results gathered -- strong scaling.

Ranks: `python bench.py --gpus N` starts N fresh child processes by itself (one per GPU; the parent never touches
the GPU, torch or HIP) unless it is already running under a launcher (WORLD_SIZE set, e.g. torch.distributed.run),
in which case --gpus must equal WORLD_SIZE.  Rank 0 prints ONE JSON line (contract in the task statement) with
  roofline      dominant kernel (encoder MFMA GEMM, HIP events around every launch of the last timed step, on the
                stream the kernels run on) + per-phase fractions and their time-weighted mean
  cpu_baseline  the CPU oracle (oracle/, a C restatement of the reference's candle CPU path) timed on a bounded
                sample of the same workload on the box's host cores (rank 0, N = 1, default workload only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (model, chunks in the job or None = batch per GPU, scaling)
    "b32": ("distil-large-v3", None, "weak"),
    "longform20": ("distil-large-v3", 20, "strong"),
    "lv3b64": ("large-v3", 64, "strong"),
}
HBM_PEAK_TBS = 8.0      # MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 (no sparsity)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="b32")
    ap.add_argument("--model", default=None, help="override the workload's model (parity-test sizes on small boxes)")
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (workload b32)")
    ap.add_argument("--max-new-tokens", type=int, default=0,
                    help="0 = reference behaviour (random weights run to the 447-token cap)")
    ap.add_argument("--pipelines", type=int, default=int(os.environ.get("NORMA_BENCH_PIPELINES", "3")),
                    help="batches in flight per GPU (workload b32): each pipeline is its own context + host thread; "
                         "encoders are serialised by a host lock, the latency-bound decode of one batch runs beside the "
                         "MFMA-bound encoder of the next.  1 = one batch at a time")
    ap.add_argument("--no-graphs", action="store_true", help="A/B: launch every decode-step kernel eagerly (NH_OPT_DECODE_GRAPHS = 0)")
    ap.add_argument("--no-ln-fusion", action="store_true", help="A/B: stand-alone decoder LayerNorm kernels (NH_OPT_FUSE_DECODE_LAYERNORM = 0)")
    ap.add_argument("--no-single-extra", action="store_true", help="skip the one-batch-at-a-time measurement in `extra`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-new-tokens", type=int, default=96)
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the ranks rendezvous (gloo), partition the job, exchange placeholder results "
                         "through the real gather and print the JSON line with value null (CPU rehearsal of the N > 1 path)")
    ap.add_argument("--print-tokens-hash", action="store_true", help="add a hash of the gathered token ids to the line")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# parent: start one child per GPU.  Nothing here imports torch or touches HIP -- a process that has initialised the GPU
# must never be the one that forks/execs the ranks.
# ---------------------------------------------------------------------------------------------------------------------
def rank_environments(n, base_env=None, port=None):
    """The environment of every child rank (what torch.distributed.run would have set)."""
    base = dict(os.environ if base_env is None else base_env)
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    envs = []
    for r in range(n):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(e)
    return envs


def spawn_ranks(n, argv):
    envs = rank_environments(n)
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = [subprocess.Popen(cmd, env=e, cwd=ROOT) for e in envs]
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with {code}; stopping the others", file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# algorithmic work, the numerators of the roofline fractions
# ---------------------------------------------------------------------------------------------------------------------
def encoder_flops_per_chunk(cfg):
    d, nm, L = cfg.d_model, cfg.num_mel_bins, cfg.encoder_layers
    return 6.0 * nm * d * 3000 + 6.0 * d * d * 1500 + L * (24.0 * 1500 * d * d + 4.0 * 1500 * 1500 * d)


def cross_kv_flops_per_chunk(cfg):
    return cfg.decoder_layers * 4.0 * 1500 * cfg.d_model ** 2


def decode_step_bytes(cfg, B, t):
    """fp16 bytes one decode step must stream: decoder layer weights + tied embedding + cross K/V + self K/V."""
    d, L, V = cfg.d_model, cfg.decoder_layers, cfg.vocab_size
    return L * 14.0 * d * d * 2 + V * d * 2.0 + B * L * 2.0 * 1500 * d * 2 + B * L * 2.0 * t * d * 2


def mel_bytes_per_chunk(cfg):
    return 480000 * 4.0 + cfg.num_mel_bins * 3000 * (4.0 + 2.0)  # PCM in, f32 mel + fp16 conv image out


def tokens_hash(results):
    import hashlib
    h = hashlib.sha256()
    for r in results:
        h.update((",".join(map(str, r["tokens"])) + ";").encode())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from norma_amd import assets_io, config, shard, synth
    import common

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # rehearsal knobs (one-GPU box): NORMA_BENCH_BACKEND=gloo and NORMA_BENCH_FORCE_DEVICE=0 let several ranks share a card
    backend = "gloo" if args.dry_run else os.environ.get("NORMA_BENCH_BACKEND", "nccl")
    if "NORMA_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["NORMA_BENCH_FORCE_DEVICE"])
    dev = None
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    model_name, job_chunks, scaling = WORKLOADS[args.workload]
    if args.model:
        model_name = args.model
    cfg = config.preset(model_name)
    tk = common.tokens_for(model_name)
    multilingual = args.workload == "lv3b64"
    C = cfg.max_target_positions
    if job_chunks is None:                      # weak: every rank its own batch
        B = args.batch
        first, total_chunks = rank * B, world * B
    else:                                       # strong: a fixed job split over the ranks
        parts = shard.partition(job_chunks, world)
        first, B = parts[rank]
        total_chunks = job_chunks
    gather_dev = dev if backend == "nccl" else None

    if args.dry_run:
        local = [dict(tokens=[tk.sot, tk.transcribe, 1000 + first + i, tk.eot], avg_logprob=-0.5, no_speech_prob=0.0,
                      no_speech_exit=False) for i in range(B)]
        allr = local
        if world > 1:
            dist.barrier()
            allr = shard.gather_results(local, total_chunks, C, device=None) if job_chunks is not None else local
            dist.barrier()
        if rank == 0:
            out = {"metric": f"audio-sec/wall-sec (xRT) {model_name} fp16", "value": None, "unit": "audio-sec/wall-sec",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": scaling,
                   "config": {"workload": args.workload, "chunks_per_rank": [c for _, c in shard.partition(total_chunks, world)]
                              if job_chunks is not None else [B] * world, "chunks": total_chunks},
                   "gathered": len(allr) if job_chunks is not None else None}
            if job_chunks is not None:
                out["gathered_third_tokens"] = [r["tokens"][2] for r in allr]
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    from norma_amd import hip
    if hip.device_count() < 1:
        raise RuntimeError("bench.py needs an MI355X; norma_amd has no CPU fallback")

    t_build = time.time()
    P = max(1, args.pipelines) if job_chunks is None else 1   # the strong workloads run their job once per step
    headline = args.workload == "b32" and args.model is None
    hms = []
    for _ in range(P):
        h = hip.HipWhisper(cfg, device=local_rank, max_batch=max(B, 1))
        h.set_mel_filters(assets_io.mel_filters(cfg.num_mel_bins))
        h.set_tokens(tk, -1 if multilingual else tk.en, tk.transcribe)
        if args.no_graphs:
            h.set_option(hip.NH_OPT_DECODE_GRAPHS, 0)
        if args.no_ln_fusion:
            h.set_option(hip.NH_OPT_FUSE_DECODE_LAYERNORM, 0)
        hms.append(h)
    hm = hms[0]
    want_cpu = (not args.no_cpu_baseline) and headline and world == 1 and rank == 0
    om = None
    if want_cpu:
        from oracle import oracle as O
        om = O.OracleModel(cfg, tk, tk.en, tk.transcribe)
    for name, arr in synth.synth_weights(cfg, seed=0):  # seed-0 N(0, 0.02^2), fp16-representable
        a16 = arr.astype(np.float16)
        for h in hms:
            h.load_tensor(name, a16)
        if om is not None:
            om.set_tensor(name, arr)
    t_build = time.time() - t_build

    # synthetic 16 kHz PCM (chunk k of the job / of the global batch), resident in HBM before the timed region
    clips = np.stack([synth.synth_pcm(first + b) for b in range(B)]) if B else np.zeros((0, synth.N_SAMPLES), np.float32)
    pcm_dev = torch.from_numpy(clips).to(dev) if B else None
    n_samples = [synth.N_SAMPLES] * B
    lang_tokens = [tk.en + i for i in range(99)]
    torch.cuda.synchronize()

    import threading
    enc_lock = threading.Lock()

    def step(max_new, h=None, pipelined=None):
        h = h or hm
        if B == 0:
            return []
        if multilingual:  # config 5: detect_language once per chunk, then timestamped decode with the detected token
            h.logmel_device(pcm_dev.data_ptr(), n_samples, synth.N_SAMPLES)
            h.encode()
            h.detect_language(lang_tokens, want_probs=False)
            return h.decode_greedy(max_new)
        if not (pipelined if pipelined is not None else P > 1):
            return h.transcribe_batch_device(pcm_dev.data_ptr(), n_samples, synth.N_SAMPLES, max_new)
        with enc_lock:  # one encoder at a time on the GPU; decodes of the other pipelines run beside it
            h.logmel_device(pcm_dev.data_ptr(), n_samples, synth.N_SAMPLES)
            h.encode()
            h.synchronize()
        return h.decode_greedy(max_new)

    def job_step(max_new):
        """One pass over the whole job: this rank's chunks, then (strong workloads) the result gather over RCCL."""
        local = step(max_new)
        if job_chunks is not None and world > 1:
            return shard.gather_results(local, total_chunks, C, device=gather_dev)
        return local

    def run_steps(n, max_new, P=P):
        """n steps spread round-robin over the pipelines (each pipeline runs its share sequentially)."""
        if P == 1:
            out = None
            for _ in range(n):
                out = job_step(max_new)
            return out
        last = [None] * P

        def worker_thread(i):
            for _ in range(i, n, P):
                last[i] = step(max_new, hms[i], pipelined=True)
        ths = [threading.Thread(target=worker_thread, args=(i,)) for i in range(P)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return next(r for r in last if r is not None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for h in hms:
            h.synchronize()

    hm.set_profile_gemm(True)  # two hipEventRecord per GEMM launch (~400 per step, < 0.5 % of a step)
    res = run_steps(max(args.warmup, P if args.warmup else 0), args.max_new_tokens)
    barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps, args.max_new_tokens)
    barrier()
    dt = time.perf_counter() - t0
    tm_timed = hm.timings() if B else None    # last step of pipeline 0 inside the timed region (GEMM launch events)
    tm = tm_timed
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    audio_s = total_chunks * 30.0 * args.steps
    value = audio_s / dt

    extra = {}
    single_ms = dt / args.steps * 1e3     # wall time of a step with one batch on the GPU (the H2D comparison below)
    if P > 1 and B:
        # one batch at a time, untimed by the contract: the per-phase times (and their roofline fractions below) are taken
        # from this pass, where no other batch shares the GPU; with several batches in flight a phase's wall time
        # includes the kernels of the other batches interleaved with it
        step(args.max_new_tokens, hm, pipelined=False)
        barrier()
        if not args.no_single_extra:
            n1 = 3
            t1 = time.perf_counter()
            for _ in range(n1):
                step(args.max_new_tokens, hm, pipelined=False)
            barrier()
            single_ms = (time.perf_counter() - t1) / n1 * 1e3
            extra["xrt_one_batch_at_a_time_per_gpu"] = B * 30.0 / (single_ms * 1e-3)
        tm = hm.timings()
    if headline and world == 1:
        # second decode-length protocol: 128 new tokens per clip, untimed by the contract
        if args.max_new_tokens == 0:
            barrier()
            t1 = time.perf_counter()
            r128 = step(128)
            barrier()
            extra["xrt_128_new_tokens_per_gpu"] = B * 30.0 / (time.perf_counter() - t1)
            extra["tokens_128"] = int(np.mean([len(r["tokens"]) for r in r128]))
            extra["decode_ms_128"] = hm.timings()["decode_ms"]
        # PCIe-inclusive rate (never `value`, task statement 4): the same step fed from HOST memory through nh_logmel,
        # i.e. 61 MB of f32 PCM cross PCIe inside the timed region
        hm.logmel_array(clips); hm.encode(); hm.decode_greedy(args.max_new_tokens)
        barrier()
        t2 = time.perf_counter()
        n_h = 2
        for _ in range(n_h):
            hm.logmel_array(clips); hm.encode(); hm.decode_greedy(args.max_new_tokens)
        barrier()
        dth = (time.perf_counter() - t2) / n_h
        extra["xrt_host_pcm_per_gpu"] = B * 30.0 / dth
        extra["h2d_pcm_ms_per_step"] = max(0.0, dth * 1e3 - single_ms)
        hm.transcribe_batch_device(pcm_dev.data_ptr(), n_samples, synth.N_SAMPLES, args.max_new_tokens)  # timings() of a plain step again
        tm = hm.timings()
        if P == 1:
            tm_timed = tm

    if rank == 0:
        # HBM traffic of the dominant kernel per launch: from the committed rocprofv3 PMC passes (FETCH_SIZE and
        # WRITE_SIZE collected in separate runs, FETCH_SIZE doubled per the gfx950 correction); counters cannot be
        # read from inside this process, so this is the last profiled value, not a live one
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")) as f:
                pj = json.load(f)
            rows = [v for k, v in pj.items() if "gemm256_f16_kernel" in k]
            nl = sum(v["launches"] for v in rows)
            traffic = sum(v["hbm_bytes_corrected"] * v["launches"] for v in rows) / nl if nl else None
        except Exception:
            traffic = None
        gemm_tflops = tm_timed["gemm_flops"] / (tm_timed["gemm_ms"] * 1e-3) / 1e12 if tm_timed and tm_timed["gemm_ms"] > 0 else 0.0
        n_tok = int(np.mean([len(r["tokens"]) for r in res])) if res else 0
        # per-phase roofline fractions of the last timed step of rank 0 (algorithmic work / phase time / nominal peak)
        phases = []
        if tm:
            prompt = 3 if (multilingual or tk.en >= 0) else 2
            steps_dec = max(tm["decode_steps"], 1)
            dec_bytes = sum(decode_step_bytes(cfg, B, t) for t in range(steps_dec))
            ph = [("mel", "hbm", B * mel_bytes_per_chunk(cfg) / 1e12, tm["mel_ms"], HBM_PEAK_TBS, "TB/s"),
                  ("encoder", "mfma", B * encoder_flops_per_chunk(cfg) / 1e12, tm["encoder_ms"], MFMA_PEAK_TFLOPS, "TFLOP/s"),
                  ("cross_kv", "mfma", B * cross_kv_flops_per_chunk(cfg) / 1e12, tm["cross_kv_ms"], MFMA_PEAK_TFLOPS, "TFLOP/s"),
                  ("decode", "hbm", dec_bytes / 1e12, tm["decode_ms"], HBM_PEAK_TBS, "TB/s")]
            tw_num = tw_den = 0.0
            for name, bound, work, ms, peak, unit in ph:
                ach = work / (ms * 1e-3) if ms > 0 else 0.0
                phases.append({"phase": name, "bound": bound, "ms": ms, "achieved": ach, "peak": peak, "unit": unit,
                               "frac": ach / peak})
                tw_num += ach / peak * ms
                tw_den += ms
            phases[-1]["us_per_token"] = tm["decode_ms"] * 1e3 / steps_dec
            phases[-1]["bytes_per_step"] = dec_bytes / steps_dec
            time_weighted = tw_num / tw_den if tw_den > 0 else None
        else:
            time_weighted = None
        out = {
            "metric": f"audio-sec/wall-sec (xRT) {model_name} fp16" + (" b32" if args.workload == "b32" else ""),
            "value": value, "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": (f"{model_name} fp16 batch={B} x 30 s clips per GPU" if job_chunks is None else
                                    f"{model_name} fp16, {total_chunks} x 30 s chunks of one job split over {world} GPU(s) "
                                    f"({[c for _, c in shard.partition(total_chunks, world)]}), results gathered over RCCL"
                                    + (", language detection + timestamp decoding" if multilingual else "")) +
                                   ", greedy decode " +
                                   ("to the 447-token cap (seed-0 random weights never emit eot)" if args.max_new_tokens == 0
                                    else f"{args.max_new_tokens} new tokens"),
                       "name": args.workload, "batch_per_gpu": B, "chunks": total_chunks, "clip_seconds": 30,
                       "decode_tokens": n_tok, "parallelism": f"chunk-dp{world}", "batches_in_flight_per_gpu": P},
            "phases_ms": {k: tm[k] for k in ("mel_ms", "encoder_ms", "cross_kv_ms", "decode_ms")} if tm else None,
            "decode_steps": tm["decode_steps"] if tm else 0,
            "roofline": {"bound": "mfma", "kernel": "gemm256_f16_kernel", "achieved": gemm_tflops, "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": gemm_tflops / MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "bytes per launch from profiles/pmc_hbm_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, "
                                         "Infinity-Cache hits included; last profiled value, not live)",
                         "launches": tm_timed["gemm_launches"] if tm_timed else 0,
                         "avg_launch_ms": tm_timed["gemm_ms"] / max(tm_timed["gemm_launches"], 1) if tm_timed else None,
                         "flops_per_step": tm_timed["gemm_flops"] if tm_timed else None,
                         "phases_note": "per-phase times of a step with ONE batch on the GPU" + (" (measured after the timed region; "
                                        f"the timed region keeps {P} batches in flight)" if P > 1 else ""),
                         "phases": phases, "time_weighted_frac": time_weighted},
            "extra": extra, "model_build_s": t_build,
        }
        if args.print_tokens_hash:
            out["tokens_hash"] = tokens_hash(res)
            out["results"] = len(res)
        if om is not None:
            from oracle import oracle as O
            clip = clips[0]
            filt = assets_io.mel_filters(cfg.num_mel_bins)
            c0 = time.perf_counter()
            mel = O.pcm_to_mel(clip, filt)
            xa = om.encoder_forward(mel)
            r = om.decode(xa, use_kv_cache=False, max_new_tokens=args.cpu_new_tokens)  # reference structure: no self-attn KV cache
            cdt = time.perf_counter() - c0
            out["cpu_baseline"] = {
                "value": 30.0 / cdt, "unit": "audio-sec/wall-sec", "cores": O.num_threads(), "kind": "port",
                "sample": f"1 clip (30 s) of the same workload: log-mel + encoder + greedy decode capped at "
                          f"{args.cpu_new_tokens} new tokens, f32, batch 1, no self-attention KV cache "
                          f"({cdt:.1f} s of CPU work); C restatement of the reference's candle CPU path",
                "seconds": cdt, "tokens": len(r["tokens"])}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for h in hms:
        h.close()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, argv)   # no launcher: be the launcher (before anything touches the GPU)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
