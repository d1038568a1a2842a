#!/usr/bin/env python3
"""bench.py -- audio-sec / wall-sec (xRT) of the MI355X Whisper hot path.

One "step" = one pass of the hot path (log-mel -> encoder -> cross-K/V -> batched greedy decode -> token ids on
the host) over one batch of synthetic 30-second clips, PCM already resident in HBM.

Workloads (`--workload`):
  b32         (default, the headline) BASELINE.json configs[2]: distil-large-v3, fp16 storage / fp32 accumulate,
              32 clips per GPU.  N > 1: every rank runs its own 32 distinct clips -- weak scaling, no data-path
              collective (chunks are independent).  A step = one 32-clip batch through log-mel, encoder,
              cross K/V and greedy decode; `--pipelines` x `--decode-groups` batches are in flight per GPU (3 x 2):
              the encoder takes them 32 at a time, two of them share one 64-row decode loop (weights and embedding
              streamed once per token for both), and three such pipelines overlap their decodes with each other's encoders.
  longform20  BASELINE.json configs[3]: one 10-minute clip (9 600 000 samples) = 20 chunks, split over the ranks with
              norma_amd.shard.partition (8 ranks: 3,3,3,3,2,2,2,2); every rank transcribes its chunks as one batch and
              the results are gathered on all ranks over RCCL inside the timed region -- strong scaling.
  lv3b64      BASELINE.json configs[4]: large-v3 multilingual, 64 chunks split over the ranks (8 per GPU at N = 8),
              language detection + timestamp decoding, results gathered -- strong scaling.
  varlen      (not a BASELINE config) distil-large-v3, 64 chunks whose transcripts END AT DIFFERENT STEPS (the audio votes
              text-vs-eot at nine steps): arrival-order batches against length-bucketed batches dealt with
              shard.partition_balanced, wasted row-steps of each reported in `extra`.
`--rank-share r/N` (strong workloads): run, on this one GPU, only the chunks rank r of an N-rank job would get -- the
per-rank time a whole-node run would be bounded by, without the node.

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
    # not a BASELINE config: the reference's decode loop ends per sequence at eot (model.rs:317); this job's 64 chunks end at
    # different steps (audio-decided eot votes), so it measures what a batch that waits for its longest sequence wastes and
    # what dealing the chunks by measured length (shard.partition_balanced / length_buckets) recovers
    "varlen": ("distil-large-v3", 64, "strong"),
}
HBM_PEAK_TBS = 8.0      # MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 (no sparsity)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)   # a multiple of pipelines x decode-groups, so that every decode is a joint one
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="b32")
    ap.add_argument("--model", default=None, help="override the workload's model (parity-test sizes on small boxes)")
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (workload b32)")
    ap.add_argument("--max-new-tokens", type=int, default=0,
                    help="0 = reference behaviour (random weights run to the 447-token cap)")
    ap.add_argument("--pipelines", type=int, default=int(os.environ.get("NORMA_BENCH_PIPELINES", "3")),
                    help="batches in flight per GPU (workload b32): each pipeline is its own context + host thread; "
                         "encoders are serialised by a host lock, the latency-bound decode of one batch runs beside the "
                         "MFMA-bound encoder of the next.  1 = one batch at a time")
    ap.add_argument("--decode-groups", type=int, default=int(os.environ.get("NORMA_BENCH_DECODE_GROUPS", "2")),
                    help="workload b32: encoder batches (of --batch clips each) that share ONE decode loop per pipeline "
                         "(nh_logmel_device_rows / nh_encode_rows, then one nh_decode_greedy over all their rows): the decoder "
                         "weights and the tied embedding are streamed once per token for all of them (default 2: +7 %% over 1 on one box, "
                         "3 no better).  1 = every batch decodes alone (r02)")
    ap.add_argument("--no-graphs", action="store_true", help="A/B: launch every decode-step kernel eagerly (NH_OPT_DECODE_GRAPHS = 0)")
    ap.add_argument("--no-ln-fusion", action="store_true", help="A/B: stand-alone decoder LayerNorm kernels (NH_OPT_FUSE_DECODE_LAYERNORM = 0)")
    ap.add_argument("--private-weights", action="store_true",
                    help="A/B: every pipeline loads its own copy of the weights (r02 behaviour) instead of sharing one (nh_create_shared)")
    ap.add_argument("--no-single-extra", action="store_true", help="skip the one-batch-at-a-time measurement in `extra`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-new-tokens", type=int, default=96)
    ap.add_argument("--rank-share", default=None, metavar="r/N",
                    help="strong workloads, --gpus 1: run only the chunks rank r of an N-rank job would get (no gather); the "
                         "line reports that share's time and the job throughput N such ranks would give")
    ap.add_argument("--balance", action="store_true",
                    help="workload varlen: `value` is the length-bucketed / partition_balanced pass (default: arrival order)")
    ap.add_argument("--pool-contexts", type=int, default=2, help="workload varlen: decode pools in flight per GPU")
    ap.add_argument("--pool-check-every", type=int, default=16, help="workload varlen: decode steps between two looks at the rows' done flags")
    ap.add_argument("--phased", action="store_true",
                    help="A/B: the pipelines alternate between an encoder phase (all of them, one after the other) and a decode phase (all of them together)")
    ap.add_argument("--pool-encoders", type=int, default=2, help="workload varlen: encoder contexts feeding ONE decoding context (0: skip that pass)")
    ap.add_argument("--pool-fed", action="store_true", help="workload varlen: `value` is the pass with one decoding context fed by --pool-encoders encoder contexts")
    ap.add_argument("--pool", action="store_true",
                    help="workload varlen: `value` is the decode-pool pass (norma_amd/pool.py; default: arrival-order lockstep batches)")
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
        if args.rank_share:                     # one rank's share of an N-rank job, alone on this GPU
            if world != 1:
                raise SystemExit("bench.py: --rank-share needs --gpus 1")
            sr, sn = (int(x) for x in args.rank_share.split("/"))
            first, B = shard.partition(job_chunks, sn)[sr]
            total_chunks = B
    gather_dev = dev if backend == "nccl" else None

    if args.workload == "varlen" and not args.dry_run:
        return worker_varlen(args, cfg, tk, world, rank, local_rank, dev, backend, gather_dev)

    if args.dry_run:
        assign = None
        if args.workload == "varlen":   # placeholder decode lengths -> the real partition_balanced + reordering gather
            fake_steps = [20 + (k * 37) % 300 for k in range(job_chunks)]
            assign = shard.partition_balanced(fake_steps, world)
            mine = assign[rank]
        else:
            mine = list(range(first, first + B))
        local = [dict(tokens=[tk.sot, tk.transcribe, 1000 + k, tk.eot], avg_logprob=-0.5, no_speech_prob=0.0,
                      no_speech_exit=False) for k in mine]
        allr = local
        if world > 1:
            dist.barrier()
            allr = shard.gather_results(local, total_chunks, C, device=None, assignment=assign) if job_chunks is not None else local
            dist.barrier()
        if rank == 0:
            out = {"metric": f"audio-sec/wall-sec (xRT) {model_name} fp16", "value": None, "unit": "audio-sec/wall-sec",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": scaling,
                   "config": {"workload": args.workload, "chunks_per_rank": ([len(a) for a in assign] if assign is not None else
                                                                              [c for _, c in shard.partition(total_chunks, world)])
                              if job_chunks is not None else [B] * world, "chunks": total_chunks},
                   "gathered": len(allr) if job_chunks is not None else None}
            if job_chunks is not None:
                out["gathered_third_tokens"] = [r["tokens"][2] for r in allr]
            if assign is not None:
                out["balanced_loads"] = [sum(fake_steps[i] for i in a) for a in assign]
                out["contiguous_loads"] = [sum(fake_steps[s0:s0 + c]) for s0, c in shard.partition(job_chunks, world)]
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    from norma_amd import hip
    if hip.device_count() < 1:
        raise RuntimeError("bench.py needs an MI355X; norma_amd has no CPU fallback")

    t_build = time.time()
    P = max(1, args.pipelines) if job_chunks is None else 1   # the strong workloads run their job once per step
    G = max(1, args.decode_groups) if job_chunks is None else 1   # encoder batches per joint decode
    if B * G > 96:
        raise SystemExit("bench.py: --batch x --decode-groups must be <= 96 rows per context")
    headline = args.workload == "b32" and args.model is None
    hms = []
    for i in range(P):
        shared = hms[0] if (i > 0 and not args.private_weights) else None   # one weight set per device (nh_create_shared)
        h = hip.HipWhisper(cfg, device=local_rank, max_batch=max(B * G, 1), share_with=shared)
        if shared is None:
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
        for h in (hms if args.private_weights else hms[:1]):
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

    def group_step(max_new, h, groups):
        """`groups` encoder batches (B clips each) through log-mel + encoder one after the other, then ONE decode over all
        their rows (lockstep positions): every batch is still one step of the contract."""
        for g in range(groups):
            with enc_lock:
                h.logmel_device_rows(pcm_dev.data_ptr(), n_samples, synth.N_SAMPLES, g * B)
                h.encode_rows(g * B, B)
                h.synchronize()
        if phase_barrier is not None:     # --phased: every pipeline has encoded before any of them decodes, and the other way round
            phase_barrier.wait()
        out = h.decode_greedy(max_new)
        if phase_barrier is not None:
            phase_barrier.wait()
        under_load.extend(out[k * B:(k + 1) * B] for k in range(groups))
        return out[:B]

    phase_barrier = threading.Barrier(P) if args.phased and P > 1 else None
    under_load = []   # every batch result of the pipelined passes: compared with the one-batch-at-a-time result after the timed region

    def run_steps(n, max_new, P=P):
        """n steps spread round-robin over the pipelines (each pipeline runs its share sequentially)."""
        if P == 1 and G == 1:
            out = None
            for _ in range(n):
                out = job_step(max_new)
            return out
        last = [None] * P

        # the n batches are dealt to the pipelines in units of G (one joint decode each; a last, smaller unit if G does not divide n)
        units = [G] * (n // G) + ([n % G] if n % G else [])

        def worker_thread(i):
            for g in units[i::P]:
                if G == 1:
                    last[i] = step(max_new, hms[i], pipelined=True)
                    under_load.append(last[i])
                else:
                    last[i] = group_step(max_new, hms[i], g)
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
    res = run_steps(max(args.warmup, P * G if args.warmup else 0), args.max_new_tokens)
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
    if (P > 1 or G > 1) and B:
        # one batch at a time, untimed by the contract: the per-phase times (and their roofline fractions below) are taken
        # from this pass, where no other batch shares the GPU; with several batches in flight a phase's wall time
        # includes the kernels of the other batches interleaved with it
        alone = step(args.max_new_tokens, hm, pipelined=False)
        barrier()
        # parity under load: every batch decoded while other batches shared the GPU (warm-up and timed passes) against the same
        # batch decoded with the GPU to itself -- tokens, avg_logprob and no_speech_prob, bit for bit
        if under_load and not multilingual:
            import struct
            bits = lambda x: struct.pack("<d", x)     # bit patterns: a NaN equals itself here
            tok_d = sum(1 for batch in under_load for a, b in zip(batch, alone) if a["tokens"] != b["tokens"])
            lp_d = sum(1 for batch in under_load for a, b in zip(batch, alone) if bits(a["avg_logprob"]) != bits(b["avg_logprob"]))
            ns_d = sum(1 for batch in under_load for a, b in zip(batch, alone) if bits(a["no_speech_prob"]) != bits(b["no_speech_prob"]))
            extra["results_under_load"] = {"batches_compared": len(under_load), "sequences_compared": len(under_load) * B,
                                           "differing_from_the_batch_decoded_alone": {"tokens": tok_d, "avg_logprob": lp_d, "no_speech_prob": ns_d}}
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
        if tm and tm is not tm_timed and tm["gemm_ms"] > 0:   # the same launches with nothing else on the GPU (no queueing behind other contexts)
            extra["gemm_tflops_one_batch_at_a_time"] = tm["gemm_flops"] / (tm["gemm_ms"] * 1e-3) / 1e12
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
            "config": {"workload": (f"{model_name} fp16 batch={B} x 30 s clips per GPU" +
                                    (f" (encoder in batches of {B}; {G} such batches share one decode loop of {B * G} rows; "
                                     f"{P} of these pipelines in flight)" if G > 1 else "") if job_chunks is None else
                                    f"{model_name} fp16, {total_chunks} x 30 s chunks of one job split over {world} GPU(s) "
                                    f"({[c for _, c in shard.partition(total_chunks, world)]}), results gathered over RCCL"
                                    + (", language detection + timestamp decoding" if multilingual else "")) +
                                   ", greedy decode " +
                                   ("to the 447-token cap (seed-0 random weights never emit eot)" if args.max_new_tokens == 0
                                    else f"{args.max_new_tokens} new tokens"),
                       "name": args.workload, "batch_per_gpu": B, "chunks": total_chunks, "clip_seconds": 30,
                       "decode_tokens": n_tok, "parallelism": f"chunk-dp{world}", "batches_in_flight_per_gpu": P * G, "pipelines_per_gpu": P,
                       "encoder_batches_per_decode": G,
                       "weight_sets_per_gpu": P if args.private_weights else 1},
            "phases_ms": {k: tm[k] for k in ("mel_ms", "encoder_ms", "cross_kv_ms", "decode_ms")} if tm else None,
            "decode_steps": tm["decode_steps"] if tm else 0,
            "roofline": {"bound": "mfma", "kernel": "gemm256_f16_kernel", "achieved": gemm_tflops, "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": gemm_tflops / MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "bytes per launch from profiles/pmc_hbm_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, "
                                         "Infinity-Cache hits included; last profiled value, not live)",
                         "achieved_note": "algorithmic FLOPs / sum of event-bracketed launch durations of one step INSIDE the timed region; with "
                                          "several batches in flight a launch's bracket includes its wait behind other contexts' kernels "
                                          "(rocprofv3 kernel durations: profiles/r03_default_kernel_stats.csv; alone: extra.gemm_tflops_one_batch_at_a_time)",
                         "launches": tm_timed["gemm_launches"] if tm_timed else 0,
                         "avg_launch_ms": tm_timed["gemm_ms"] / max(tm_timed["gemm_launches"], 1) if tm_timed else None,
                         "flops_per_step": tm_timed["gemm_flops"] if tm_timed else None,
                         "phases_note": "per-phase times of a step with ONE batch on the GPU" + (" (measured after the timed region; "
                                        f"the timed region keeps {P} batches in flight)" if P > 1 else ""),
                         "phases": phases, "time_weighted_frac": time_weighted},
            "extra": extra, "model_build_s": t_build,
        }
        if args.rank_share:
            sr, sn = (int(x) for x in args.rank_share.split("/"))
            out["rank_share"] = {"rank": sr, "of": sn, "chunks": B, "job_chunks": job_chunks, "ms": dt / args.steps * 1e3,
                                 "implied_job_xrt": job_chunks * 30.0 / (dt / args.steps),
                                 "note": "time of ONE rank's share alone on one GPU; rank 0 holds the largest share, so "
                                         "job_chunks x 30 s / this time is the whole-job rate N such ranks would reach "
                                         "(gather excluded: <= 115 KB)"}
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


# ---------------------------------------------------------------------------------------------------------------------
# workload varlen: sequences that end at different steps
# ---------------------------------------------------------------------------------------------------------------------
VARLEN_EOT_STEPS = [40, 55, 70, 85, 100, 120, 150, 200, 280]   # audio-decided (text | eot) votes at these text steps
VARLEN_TEXT_STEPS = 360


def varlen_spec(cfg, tk):
    """tests/common.py:audio_overrides spec: one text pair at every step but nine, where the second token of the pair is
    eot -- about half of the clips still running stop at each of them (which half is decided by the clip's audio)."""
    import numpy as np
    rng = np.random.default_rng(77)
    sup = set(cfg.suppress_tokens)

    def tok():
        while True:
            t = int(rng.integers(300, 40000))
            if t not in sup:
                return t
    pairs, seq = [[tok(), tok(), 0]], [0] * VARLEN_TEXT_STEPS
    for j, e in enumerate(VARLEN_EOT_STEPS, start=1):
        pairs.append([tok(), tk.eot, j])
        seq[e] = j
    return dict(conv_amp=10.0, pos_rms=4.0, peak_logit=14.0, gamma=0.0, segment=20, pairs=pairs, att_ref=[0.0] * cfg.d_model, seq=seq)


def worker_varlen(args, cfg, tk, world, rank, local_rank, dev, backend, gather_dev):
    import numpy as np
    import torch
    import torch.distributed as dist
    from norma_amd import assets_io, hip, shard, synth
    import common
    job, C, BMAX = WORKLOADS["varlen"][1], cfg.max_target_positions, args.batch
    hm = hip.HipWhisper(cfg, device=local_rank, max_batch=BMAX)
    hm.set_mel_filters(assets_io.mel_filters(cfg.num_mel_bins))
    hm.set_tokens(tk, tk.en, tk.transcribe)
    spec = varlen_spec(cfg, tk)
    over0, _ = common.audio_overrides(cfg, tk, spec)
    for name, arr in synth.synth_weights(cfg, 0, over0):
        hm.load_tensor(name, arr.astype(np.float16))
    # calibrate the votes on chunks 0 .. 15 (every rank does the same: the kernels are deterministic, so all ranks end up
    # with bit-identical weights): reference read-out and gain of the last decoder layer's cross-attention (common.py)
    hm.logmel([synth.synth_pcm(k) for k in range(16)]); hm.encode()
    means = [hm.encoder_output(b).mean(0, keepdims=True) for b in range(16)]
    common.audio_calibrate(cfg, means, spec, vote=1.5)
    over, _ = common.audio_overrides(cfg, tk, spec)
    lastp = f"model.decoder.layers.{cfg.decoder_layers - 1}.encoder_attn.out_proj"
    for leaf in (".weight", ".bias"):
        hm.load_tensor(lastp + leaf, over[lastp + leaf].astype(np.float16))
    pcm_all = torch.from_numpy(np.stack([synth.synth_pcm(k) for k in range(job)])).to(dev)   # the whole job, resident in HBM
    stage = torch.empty((BMAX, synth.N_SAMPLES), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def run(assign, batches_of):
        """one pass over the job: this rank's chunks in batches, then the gather"""
        mine, res, batches = assign[rank], {}, batches_of(assign[rank])
        for b in batches:
            idx = torch.tensor(b, dtype=torch.long, device=dev)
            torch.index_select(pcm_all, 0, idx, out=stage[:len(b)])
            torch.cuda.synchronize()
            out = hm.transcribe_batch_device(stage.data_ptr(), [synth.N_SAMPLES] * len(b), synth.N_SAMPLES, 0)
            res.update(zip(b, out))
        local = [res[k] for k in mine]
        if world > 1:
            return shard.gather_results(local, job, C, device=gather_dev, assignment=assign), batches
        return [res[k] for k in range(job)], batches

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(); hm.synchronize()

    def timed(assign, batches_of):
        run(assign, batches_of); barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            allr, batches = run(assign, batches_of)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return allr, batches, dt / args.steps

    contiguous = [list(range(s0, s0 + c)) for s0, c in shard.partition(job, world)]
    arrival = lambda mine: [mine[i:i + BMAX] for i in range(0, len(mine), BMAX)]
    hm.set_profile_gemm(True)
    res_a, _, t_a = timed(contiguous, arrival)
    steps = [len(r["tokens"]) - 3 for r in res_a]              # decode steps every chunk needs (prompt excluded)
    balanced = shard.partition_balanced(steps, world)           # measured lengths -> ranks (LPT), then buckets inside a rank
    res_b, _, t_b = timed(balanced, lambda mine: shard.length_buckets(mine, steps, BMAX))
    identical = [a["tokens"] for a in res_a] == [b["tokens"] for b in res_b]
    # (c) the decode pool (nh_pool_*, norma_amd/pool.py): the same chunks as ONE stream of steps x job clips in arrival order
    # -- nothing is known about their lengths -- through POOL_ROWS decode rows that are refilled as sequences finish
    from norma_amd import pool
    POOL_ROWS = 64
    import threading
    NPOOL = max(1, args.pool_contexts)      # pools in flight on this GPU: one's encoder submission overlaps the others' decode steps
    hmps = [hip.HipWhisper(cfg, device=local_rank, max_batch=POOL_ROWS + BMAX, share_with=hm) for _ in range(NPOOL)]
    stages = [torch.empty((BMAX, synth.N_SAMPLES), dtype=torch.float32, device=dev) for _ in range(NPOOL)]
    for h in hmps:
        h.set_tokens(tk, tk.en, tk.transcribe)
    stream = args.steps * job
    s0, cnt = shard.partition(stream, world)[rank]
    enc_lock = threading.Lock()

    def pool_pass(first, count):
        parts = shard.partition(count, NPOOL)
        out, dps, errs = [None] * NPOOL, [None] * NPOOL, []

        def one(i):
            try:
                torch.cuda.set_device(dev)
                hmp, stg, (p0, pc) = hmps[i], stages[i], parts[i]
                dp = pool.DecodePool(hmp, rows=POOL_ROWS, staging=BMAX, check_every=args.pool_check_every)

                def encode(f, n, row0, must):
                    # one encoder submission at a time on the GPU; the decode steps of the other pools run beside it, and a pool
                    # that still has rows decoding does not wait for the encoder (must = False): it steps on and asks again
                    if not enc_lock.acquire(blocking=must):
                        return False
                    try:
                        idx = torch.tensor([(first + p0 + f + k) % job for k in range(n)], dtype=torch.long, device=dev)
                        torch.index_select(pcm_all, 0, idx, out=stg[:n])
                        torch.cuda.synchronize()
                        hmp.logmel_device_rows(stg.data_ptr(), [synth.N_SAMPLES] * n, synth.N_SAMPLES, row0)
                        hmp.encode_rows(row0, n)
                        hmp.synchronize()
                    finally:
                        enc_lock.release()
                    return True
                out[i] = dp.run(pc, encode) if pc else []
                dps[i] = dp
            except BaseException as e:   # noqa: BLE001 -- re-raised on the main thread
                errs.append(e)
        ths = [threading.Thread(target=one, args=(i,)) for i in range(NPOOL)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        tot = pool.DecodePool(None, rows=POOL_ROWS, staging=BMAX)
        for d in dps:
            tot.row_steps += d.row_steps; tot.steps += d.steps; tot.encodes += d.encodes
        return [r for part in out for r in part], tot
    pool_pass(0, min(job, cnt)); barrier()
    t0 = time.perf_counter()
    res_p, dp = pool_pass(s0, cnt)
    if world > 1:
        shard.gather_results(res_p, stream, C, device=gather_dev)
    barrier()
    t_p = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([t_p], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_p = float(t.item())
    pool_same = all(r["tokens"] == res_a[(s0 + i) % job]["tokens"] and r["avg_logprob"] == res_a[(s0 + i) % job]["avg_logprob"]
                    for i, r in enumerate(res_p))
    # (d) one decoding context fed by encoder contexts (nh_pool_admit_from, pool.FedDecodePool): the pool's stream never runs an encoder
    fed = None
    if args.pool_encoders > 0:
        FROWS = 64
        hf = hip.HipWhisper(cfg, device=local_rank, max_batch=FROWS + 1, share_with=hm)
        hf.set_tokens(tk, tk.en, tk.transcribe)
        hes = [hip.HipWhisper(cfg, device=local_rank, max_batch=BMAX, share_with=hm) for _ in range(args.pool_encoders)]
        for h in hes:
            h.set_tokens(tk, tk.en, tk.transcribe)
        estage = [torch.empty((BMAX, synth.N_SAMPLES), dtype=torch.float32, device=dev) for _ in hes]

        def fed_pass(first, count):
            fp = pool.FedDecodePool(hf, hes, rows=FROWS, batch=BMAX, check_every=args.pool_check_every)

            def encode(i, f, n):
                torch.cuda.set_device(dev)
                c0 = (first + f) % job
                if c0 + n <= job:       # the clips lie next to each other in HBM: no staging copy
                    ptr = pcm_all[c0].data_ptr()
                else:
                    idx = torch.tensor([(first + f + k) % job for k in range(n)], dtype=torch.long, device=dev)
                    torch.index_select(pcm_all, 0, idx, out=estage[i][:n])
                    torch.cuda.synchronize()
                    ptr = estage[i].data_ptr()
                hes[i].logmel_device(ptr, [synth.N_SAMPLES] * n, synth.N_SAMPLES)
                hes[i].encode()
                hes[i].synchronize()
            return fp.run(count, encode), fp
        fed_pass(0, min(job, cnt)); barrier()
        t0 = time.perf_counter()
        res_f, fp = fed_pass(s0, cnt)
        if world > 1:
            shard.gather_results(res_f, stream, C, device=gather_dev)
        barrier()
        t_f = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([t_f], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_f = float(t.item())
        fed_same = all(r["tokens"] == res_a[(s0 + i) % job]["tokens"] and r["avg_logprob"] == res_a[(s0 + i) % job]["avg_logprob"]
                       for i, r in enumerate(res_f))
        fst = torch.tensor([fp.row_steps, fp.steps, fp.encodes, int(fed_same)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(fst, op=dist.ReduceOp.SUM)
        fed = (t_f, [float(v) for v in fst.tolist()], FROWS)
        hf.close()
        for h in hes:
            h.close()
    pool_bad = [(s0 + i, r["tokens"] == res_a[(s0 + i) % job]["tokens"], r["avg_logprob"] - res_a[(s0 + i) % job]["avg_logprob"])
                for i, r in enumerate(res_p) if r["tokens"] != res_a[(s0 + i) % job]["tokens"] or r["avg_logprob"] != res_a[(s0 + i) % job]["avg_logprob"]]
    pool_stats = torch.tensor([dp.row_steps, dp.steps, dp.encodes, int(pool_same)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(pool_stats, op=dist.ReduceOp.SUM)
    pool_stats = [float(v) for v in pool_stats.tolist()]
    tm = hm.timings()
    if rank == 0:
        def summary(assign, batches_of, t):
            run_rs = need_rs = 0
            loads = []
            for r in range(world):
                bs = batches_of(assign[r])
                a, b = shard.wasted_row_steps(bs, steps)
                run_rs += a; need_rs += b
                loads.append(sum(max(steps[i] for i in bt) for bt in bs if bt))
            return {"xrt": job * 30.0 / t, "ms_per_job": t * 1e3, "row_steps_run": run_rs, "row_steps_needed": need_rs,
                    "wasted_row_step_frac": 1.0 - need_rs / max(run_rs, 1), "decode_steps_per_rank": loads}
        sa = summary(contiguous, arrival, t_a)
        sb = summary(balanced, lambda mine: shard.length_buckets(mine, steps, BMAX), t_b)
        need_stream = args.steps * sum(steps)
        sp = {"xrt": stream * 30.0 / t_p, "ms_per_job": t_p * 1e3 / args.steps, "stream_chunks": stream, "pools_per_gpu": NPOOL, "decode_rows": POOL_ROWS,
              "staging_rows": BMAX, "check_every": args.pool_check_every, "row_steps_run": pool_stats[0], "row_steps_needed": need_stream,
              "wasted_row_step_frac": 1.0 - need_stream / max(pool_stats[0], 1.0), "decode_steps_launched": pool_stats[1],
              "encoder_submissions": pool_stats[2], "same_tokens_and_logprobs_as_lockstep": pool_stats[3] == world,
              "mismatches_rank0": pool_bad[:12], "mismatch_indices_rank0": [b[0] for b in pool_bad], "n_mismatches_rank0": len(pool_bad)}
        sf = None
        if fed is not None:
            sf = {"xrt": stream * 30.0 / fed[0], "ms_per_job": fed[0] * 1e3 / args.steps, "stream_chunks": stream, "decode_rows": fed[2],
                  "encoder_contexts": args.pool_encoders, "row_steps_run": fed[1][0], "row_steps_needed": need_stream,
                  "wasted_row_step_frac": 1.0 - need_stream / max(fed[1][0], 1.0), "decode_steps_launched": fed[1][1],
                  "encoder_submissions": fed[1][2], "same_tokens_and_logprobs_as_lockstep": fed[1][3] == world}
        t_val = (fed[0] / args.steps if (args.pool_fed and fed is not None) else t_p / args.steps if args.pool else (t_b if args.balance else t_a))
        gemm_tflops = tm["gemm_flops"] / (tm["gemm_ms"] * 1e-3) / 1e12 if tm["gemm_ms"] > 0 else 0.0
        out = {"metric": "audio-sec/wall-sec (xRT) distil-large-v3 fp16, variable decode lengths", "value": job * 30.0 / t_val,
               "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": args.steps, "warmup": 1, "ms_per_step": t_val * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": f"{job} x 30 s chunks whose transcripts end at different steps (audio-decided eot votes at text "
                                      f"steps {VARLEN_EOT_STEPS}), batches of <= {BMAX}, " +
                                      ("one stream through a 64-row decode pool fed by encoder contexts of the same weight set" if args.pool_fed else
                                       "one stream through a 64-row decode pool (sequences join and leave a running decode)" if args.pool else
                                       "length-bucketed batches dealt with partition_balanced" if args.balance else "arrival-order batches"),
                          "name": "varlen", "chunks": job, "batch_per_gpu": BMAX, "parallelism": f"chunk-dp{world}"},
               "roofline": {"bound": "mfma", "kernel": "gemm256_f16_kernel", "achieved": gemm_tflops, "peak": MFMA_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": gemm_tflops / MFMA_PEAK_TFLOPS, "traffic": None},
               "cpu_baseline": None,
               "extra": {"decode_steps": {"min": min(steps), "median": int(np.median(steps)), "mean": float(np.mean(steps)), "max": max(steps),
                                          "histogram": {str(v): steps.count(v) for v in sorted(set(steps))}},
                         "arrival_order": sa, "length_bucketed": sb, "decode_pool": sp, "decode_pool_fed_by_encoder_contexts": sf,
                         "same_tokens_both_ways": identical}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for h in hmps:
        h.close()
    hm.close()
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
