#!/usr/bin/env python3
"""Benchmark of the MI355X Whisper hot path.

    python bench.py --gpus N --steps K --warmup W [--workload whisper|logmel]

Default workload `whisper` (BASELINE.json's metric "audio-sec transcribed/sec, whisper-large-v3
greedy", configs[3]): 10 h of synthetic 16 kHz audio = 1200 windows of 30 s, dealt round-robin to
the ranks (`dist.window_shard`), every rank holding the PCM of its windows in HBM.  One step =
the whole 10 h once: per batch of windows log-mel -> encoder -> cross K/V -> 128 greedy tokens
(early stopping off: random weights never emit EOS reliably), then the per-window token lists
gathered on every rank - inside the timed region.  Total work is fixed as N grows: strong scaling.
The same line carries the log-mel workload (configs[1]: 1 h of audio, 80 mel bins, fp32, one
full pass per step, per-rank clips) under "logmel" with its own roofline and CPU baseline.

`--workload logmel` runs only the frontend line (weak scaling: every rank owns its own clip).

With `--gpus N > 1` and no RANK in the environment the script starts N ranks itself (one
process per GPU, RCCL) before anything touches a GPU; under torch.distributed.run it uses the
ranks it is given.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense bf16
PROMPT = [50258, 50259, 50359]
# The chip needs about 20 ms of sustained work before a 0.14 ms kernel runs at its steady rate
# (scripts/time_frontend_sustained.py: 151 us per 1 h call over the first 20 calls, 133.5 us from
# the 400th on), so the log-mel legs warm up for this many steps before they time anything.
LOGMEL_WARMUP = 400


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 2 (whisper) / 200 (logmel)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 1 (whisper) / 400 (logmel)")
    ap.add_argument("--workload", default="whisper", choices=("whisper", "logmel"))
    ap.add_argument("--hours", type=float, default=1.0, help="logmel: audio hours per rank and step")
    ap.add_argument("--n-mels", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--windows", type=int, default=1200, help="whisper: 30 s windows in the job (10 h)")
    ap.add_argument("--batch", type=int, default=600, help="whisper: windows per engine call at most")
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--logmel-steps", type=int, default=200, help="steps of the log-mel leg of the whisper line (0 = skip)")
    ap.add_argument("--pipelined", type=int, default=1, help="whisper: also measure two engines per GPU, one phase apart")
    ap.add_argument("--beam-windows", type=int, default=100,
                    help="whisper: windows per call of the beam-search leg (configs[4]; 0 = skip)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 2 if a.workload == "whisper" else 200
    if a.warmup is None:
        a.warmup = 1 if a.workload == "whisper" else LOGMEL_WARMUP
    return a


_T0 = time.time()


def say(msg):
    """Progress on stderr (STT_BENCH_VERBOSE=1): names the stage a run was in when it died."""
    if os.environ.get("STT_BENCH_VERBOSE"):
        sys.stderr.write("[bench %7.2f s] %s\n" % (time.time() - _T0, msg))
        sys.stderr.flush()
    if os.environ.get("STT_BENCH_DUMP_MAPS"):        # the address map, to place a fault address afterwards
        with open("/proc/self/maps") as f, open(os.environ["STT_BENCH_DUMP_MAPS"], "w") as out:
            out.write("# %s\n" % msg)
            out.write(f.read())


# ------------------------------------------------------------------------------ launching
def spawn_ranks(args):
    """`--gpus N` given to a plain `python bench.py`: start N ranks (no GPU call has been made in
    this process), relay their output, return the worst exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable] + sys.argv, env=env))
    return max(p.wait() for p in procs)


def init_dist(backend):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())       # rehearsal on fewer GPUs than ranks
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the backends' native code may greet on stdout (gloo prints its peer count): stdout is kept for
        # the one JSON line, so it points at stderr while the group forms
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
            dist.barrier()
        finally:
            os.dup2(keep, 1)
            os.close(keep)
        world = dist.get_world_size()
    return rank, world, local


def _coll_device():
    import torch.distributed as dist
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=_coll_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return x


def all_ranks_ok(ok, world):
    """True when every rank says ok.  A collective that failed ranks reach too, so that a rank
    whose leg raised (out of memory, ...) does not leave the others waiting in a barrier."""
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_coll_device())
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())
    return bool(ok)


def measured_traffic(csv_name, kernel_prefix):
    """HBM bytes per launch of one kernel from a committed `rocprofv3 --pmc` summary under profiles/
    (FETCH_SIZE and WRITE_SIZE passes, corrected as the file's header says).  The counters cannot be
    read from inside the process, so this is the figure of the profiled run, not of this one; None when
    the file is missing."""
    path = next((q for q in (os.path.join(ROOT, "profiles", r, csv_name) for r in ("r3", "r2"))
                 if os.path.exists(q)), None)
    if path is None:
        return None
    import csv
    with open(path) as f:
        rows = list(csv.DictReader(line for line in f if not line.startswith("#")))
    total = 0
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        hit = [r for r in rows if r["kernel"].startswith(kernel_prefix) and r["counter"] == counter]
        if not hit:
            return None
        total += int(hit[0]["bytes_corrected"])
    return total


def host_threads():
    """Threads the CPU legs use: the share of the host this job is given (the GPU box exports
    OMP_NUM_THREADS = 16 per GPU and asks jobs to stay within it), never more than the host has."""
    return min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))


def host_cpu():
    """What the CPU baseline ran on: the model string, the CPUs the host shows and the threads used."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "host_cpus": os.cpu_count(), "threads_used": host_threads(),
            "note": "the box grants 16 CPUs per GPU (OMP_NUM_THREADS); the baseline uses all of them"}


# ------------------------------------------------------------------------------ log-mel
def bench_logmel(args, rank, world, steps, warmup):
    from speech_to_text_amd import _lib
    from speech_to_text_amd.audio.stft import WhisperSTFT

    lib = _lib.load()
    n = int(round(57_600_000 * args.hours))
    fn = WhisperSTFT(n_mel_channels=args.n_mels)
    dev = torch.device("cuda", torch.cuda.current_device())
    # synthetic PCM uniform(-1, 1) from numpy's default_rng(rank), generated on the host, then made resident
    rng = np.random.default_rng(rank)
    pcm = torch.from_numpy(rng.uniform(-1, 1, n).astype(np.float32)).to(dev)
    frames = lib.stt_logmel_num_frames(n)
    ws_bytes = lib.stt_logmel_workspace_bytes(n)
    mel = torch.empty((frames, args.n_mels), dtype=torch.float32, device=dev)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    plan = fn._plan(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        _lib.check(lib.stt_logmel_run(plan, pcm.data_ptr(), n, mel.data_ptr(), ws.data_ptr(),
                                      ws_bytes, stream))

    for _ in range(warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0            # this rank's K steps; the closing barrier is not timed
    barrier(world)
    dt = max_over_ranks(dt, world)

    alg_bytes = n * 4 + frames * args.n_mels * 4          # each sample read once, each mel written once
    value = world * steps * alg_bytes / dt / 1e9           # GB/s, whole job

    # dominant kernel (logmel_pair), timed with events on the launching stream
    piece_max = ws[:4].view(torch.float32)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        _lib.check(lib.stt_logmel_stage1(plan, pcm.data_ptr(), 0, n, n, 0, frames, mel.data_ptr(),
                                         piece_max.data_ptr(), ws.data_ptr(), ws_bytes, stream))
        b.record()
    torch.cuda.synchronize()
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "logmel_pair<%d>" % args.n_mels, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "traffic_note": "HBM bytes cannot be read from inside the process",
                "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_launch": alg_bytes}
    if args.n_mels == 80 and abs(args.hours - 1.0) < 1e-9:
        t = measured_traffic("logmel_1h_pmc_hbm.csv", "logmel_pair<80")
        if t is not None:
            roofline["traffic"] = t
            roofline["traffic_note"] = ("bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                        "over this workload (profiles/r2/logmel_1h_pmc_hbm.csv); the counters cannot "
                                        "be read from inside the process")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import frontend as oracle
        threads = host_threads()
        sample_s = min(3600, n // 16000)
        host = pcm[: sample_s * 16000].cpu().numpy()
        oracle.whisper_log_mel_torch(host[:1_600_000], args.n_mels, threads=threads)
        best = 1e30
        for _ in range(3):
            t = time.perf_counter()
            ref = oracle.whisper_log_mel_torch(host, args.n_mels, threads=threads)
            best = min(best, time.perf_counter() - t)
        cpu_bytes = host.size * 4 + ref.size * 4
        cpu = {"value": round(cpu_bytes / best / 1e9, 3), "unit": "GB/s", "cores": threads,
               "kind": "port", "sample": "%d s of the same synthetic clip, torch-CPU conv1d "
               "restatement of the reference (oracle/frontend.py), best of 3" % sample_s,
               "seconds": round(best, 3), "host": host_cpu()}
        # parity spot-check of the timed output against the oracle on that sample
        step()
        torch.cuda.synchronize()
        got = mel[: ref.shape[0] - 2].cpu().numpy()
        cpu["max_abs_diff_vs_gpu"] = float(np.abs(got - ref[: got.shape[0]]).max()) \
            if sample_s * 16000 == n else None

    return {
        "metric": "log-mel GB/s (algorithmic bytes: PCM read once + mel written once)",
        "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "log-mel frontend, %.3g h of 16 kHz audio per GPU, %d mel bins, fp32 "
                   "(BASELINE.json configs[1])" % (args.hours, args.n_mels),
                   "audio_seconds_per_step": world * n / 16000.0},
        "audio_sec_per_sec": round(world * steps * (n / 16000.0) / dt, 1),
        "roofline": roofline, "cpu_baseline": cpu,
    }


# ------------------------------------------------------------------------------ whisper
def whisper_flops(dims, batch, new_tokens, prompt_len):
    """Algorithmic FLOPs (2 per multiply-add): encoder, cross-K/V, decode of `batch` windows."""
    d, L, T, M = dims["n_audio_state"], dims["n_audio_layer"], 1500, dims["n_mels"]
    stem = 2 * 3000 * 3 * M * d + 2 * T * 3 * d * d
    layer = 2 * T * (4 * d * d + 8 * d * d) + 4 * T * T * d
    enc = stem + L * layer
    cross = dims["n_text_layer"] * 2 * T * 2 * d * d
    tok = dims["n_text_layer"] * (2 * (4 * d * d + 2 * d * d + 8 * d * d) + 4 * T * d) + 2 * d * dims["n_vocab"]
    return batch * enc, batch * cross, batch * tok * (new_tokens + prompt_len - 1)


def token_step_bytes(dims, batch):
    """Algorithmic HBM bytes of one greedy token step: the decoder's bf16 weights once, the
    cross-attention K/V of every window once (the dominant term); self K/V ignored."""
    d, L, V = dims["n_text_state"], dims["n_text_layer"], dims["n_vocab"]
    return (L * 14 * d * d + V * d) * 2 + batch * L * 2 * 1500 * d * 2


def probe_cross_attention(eng, dims, n_windows, per_window=1):
    """Mean duration (ms) of one cross-attention launch of a token step over the engine's encoded
    batch, HIP events around single launches on this stream (stt_engine_probe_kernel issues the
    launch of a token step; inside stt_generate it sits in a replayed graph), and its algorithmic
    bytes: K and V of every window, bf16, one layer."""
    layers = dims["n_text_layer"]
    code = 0 if per_window == 1 else (0x100 | per_window)
    probes = []
    for it in range(3 * layers):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.probe_kernel(code, it % layers)
        b.record()
        probes.append((a, b))
    torch.cuda.synchronize()
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in probes[layers:]]))
    return k_ms, n_windows * 2 * 1500 * dims["n_text_state"] * 2


def bench_whisper(args, rank, world):
    """BASELINE configs[3].  Returns the headline line (rank 0 keeps it)."""
    import torch.distributed as dist
    from speech_to_text_amd.audio.stft import WhisperSTFT
    from speech_to_text_amd.dist import gather_window_results, window_shard
    from speech_to_text_amd.engine import WhisperEngine, whisper_dims
    from speech_to_text_amd.weights import random_whisper_weights

    dims = whisper_dims(args.model)
    steps, warmup = args.steps, args.warmup
    err = None
    try:
        mine = window_shard(args.windows, rank, world)
        per_call = min(args.batch, max(1, len(mine)))
        batches = [mine[i: i + per_call] for i in range(0, len(mine), per_call)]
        # "peaked" synthetic weights: the decoded ids depend on the audio and on the history, so the
        # check of the tokens against the CPU oracle below can fail
        weights = random_whisper_weights(dims, 0, style="peaked")
        say("weights drawn; building the engine for %d windows per call" % per_call)
        eng = WhisperEngine(dims, weights, max_batch=per_call, max_beams=1,
                            max_text_len=args.new_tokens + len(PROMPT) + 1, max_prompt_len=4)
        fn = WhisperSTFT(n_mel_channels=dims["n_mels"])
        dev = torch.device("cuda", torch.cuda.current_device())
        # this rank's windows, resident in HBM, each seeded by its window index: faint uniform noise
        # under 400 short tone bursts (40-200 ms, 100-7000 Hz, raised-cosine envelopes) at random
        # times, so that the mel of a window varies along time and from window to window and the
        # decoded ids depend on which window they came from (white noise alone gives every window
        # the same flat mel, and a check of the last window against the oracle then says little)
        pcm = torch.empty((len(mine), 480000), dtype=torch.float32, device=dev)
        n_bursts, longest = 400, 3200
        k_axis = torch.arange(longest, device=dev, dtype=torch.float32)[None]
        for i, w in enumerate(mine):
            g = torch.Generator(device=dev).manual_seed(1000 + w)
            x = (torch.rand(480000, generator=g, device=dev) * 2 - 1) * 0.02
            par = torch.rand(n_bursts, 5, generator=g, device=dev)
            start = (par[:, 0] * (480000 - longest)).long()[:, None]
            length = (640 + par[:, 1] * (longest - 640))[:, None]
            freq = (100.0 + 6900.0 * par[:, 2])[:, None]
            amp = (0.1 + 0.4 * par[:, 3])[:, None]
            env = 0.5 - 0.5 * torch.cos(2 * np.pi * (k_axis / length).clamp(max=1.0))
            burst = amp * env * torch.sin(2 * np.pi * freq / 16000.0 * k_axis + 6.2832 * par[:, 4:5])
            x.index_add_(0, (start + k_axis.long()).reshape(-1), burst.reshape(-1))
            pcm[i] = x.clamp_(-1.0, 1.0)
        starts = np.cumsum([0] + [len(b) for b in batches])
        say("engine built, %d windows of PCM resident at %#x" % (len(mine), pcm.data_ptr()))

        def log_mel_pieces(bounds, mel_fn=fn, reduce=True):
            """First half of the log-mel of each run of windows [lo, hi) (30 s = 3000 frames each)
            and the job-wide maximum: the 10 h are ONE clip, so the `max - 8` clamp is taken over
            every window of every rank (utils/audio/stft.py:355-357 clamps over the whole clip):
            one float per rank through all-reduce(MAX)."""
            pieces = []
            for lo, hi in bounds:
                flat = pcm[lo:hi].reshape(-1)
                pieces.append(mel_fn.piece_spectrogram(flat, 0, flat.numel(), 0, (hi - lo) * 3000))
            clip_max = torch.stack([p[1] for p in pieces]).max(dim=0).values
            if world > 1 and reduce:
                if dist.get_backend() == "nccl":
                    dist.all_reduce(clip_max, op=dist.ReduceOp.MAX)
                else:
                    host = clip_max.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.MAX)
                    clip_max = host.to(dev)
            return pieces, clip_max

        def finish_mel(piece, clip_max, n, mel_fn=fn):
            return mel_fn.finish_piece(piece[0], clip_max, piece[2]).view(n, 3000, dims["n_mels"])

        def decode(engine, n, **kw):
            prompt = np.array([PROMPT] * n, dtype=np.int32)
            return engine.generate(prompt, max_new_tokens=args.new_tokens, eos_token=50257, pad_token=50257,
                                   early_stopping=False, **kw)

        def run_batch(lo, hi):
            """One engine call on its own (the extra legs, no collective: the clamp maximum is that
            of the call's windows): log-mel, encoder, tokens."""
            pieces, clip_max = log_mel_pieces([(lo, hi)], reduce=False)
            eng.encode(finish_mel(pieces[0], clip_max, hi - lo))
            return decode(eng, hi - lo)

        def step():
            pieces, clip_max = log_mel_pieces([(int(starts[i]), int(starts[i + 1])) for i in range(len(batches))])
            local = {}
            for i, ws_ in enumerate(batches):
                say("step: call %d of %d (%d windows): encode" % (i + 1, len(batches), len(ws_)))
                eng.encode(finish_mel(pieces[i], clip_max, len(ws_)))
                say("step: call %d: generate" % (i + 1))
                toks, lens, _ = decode(eng, len(ws_))
                for j, w in enumerate(ws_):
                    local[w] = toks[j, : lens[j]]
            return gather_window_results(local)          # KB-scale all_gather_object, part of the step

        for _ in range(warmup):
            step()
    except Exception as e:                  # noqa: BLE001
        err = e
    if not all_ranks_ok(err is None, world):
        raise err if err is not None else RuntimeError("another rank failed to set up the whisper workload")
    barrier(world)
    t0 = time.perf_counter()
    try:
        for _ in range(steps):
            merged = step()
        torch.cuda.synchronize()
    except Exception as e:                  # noqa: BLE001
        err = e
    dt = time.perf_counter() - t0
    barrier(world)
    if not all_ranks_ok(err is None, world):
        raise err if err is not None else RuntimeError("another rank failed in the whisper workload")
    dt = max_over_ranks(dt, world)
    assert sorted(merged) == list(range(args.windows)), "gather lost windows"
    audio_s = steps * args.windows * 30.0
    say("timed steps done: %.2f s per step" % (dt / steps))

    # ---- phases of one full batch on this rank, event-timed on the launching stream
    n_b = len(batches[0])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    ev[0].record()
    pieces, clip_max = log_mel_pieces([(0, n_b)], reduce=False)
    mel = finish_mel(pieces[0], clip_max, n_b)
    ev[1].record()
    eng.encode(mel)
    ev[2].record()
    torch.cuda.synchronize()
    mel_ms, enc_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    t1 = time.perf_counter()
    decode(eng, n_b)
    gen_ms = (time.perf_counter() - t1) * 1e3
    tok_ms = gen_ms / args.new_tokens
    f_enc, f_cross, _ = whisper_flops(dims, n_b, args.new_tokens, len(PROMPT))
    # what the CPU oracle is asked about afterwards: first and last window of this call - their mel, the
    # engine's teacher-forced logits over the prompt, and ids + summed log-probabilities of a short decode
    pick = [0, n_b - 1] if n_b > 1 else [0]
    n_check = min(24, args.new_tokens)
    check = {"mel": mel[pick].cpu(),
             "prompt_logits": eng.decoder_logits(np.array([PROMPT] * n_b, dtype=np.int32))[pick].cpu()}
    ck_tokens, _, ck_scores = eng.generate(np.array([PROMPT] * n_b, dtype=np.int32), max_new_tokens=n_check,
                                           eos_token=50257, pad_token=50257, early_stopping=False)
    check["tokens"], check["scores"] = ck_tokens[pick], ck_scores[pick]

    # ---- dominant kernel: the decoder's cross-attention, one launch per layer, token and engine
    # call.  Timed live at the full call size (the roofline figure) and at every other call size of
    # a step, so that its share of the step is summed over the calls as they really are.
    k_ms, k_bytes = probe_cross_attention(eng, dims, n_b)
    k_gbs = k_bytes / (k_ms * 1e-3) / 1e9
    per_size = {n_b: k_ms}
    for b in batches:
        if len(b) not in per_size:
            pieces, clip_max = log_mel_pieces([(0, len(b))], reduce=False)
            eng.encode(finish_mel(pieces[0], clip_max, len(b)))
            per_size[len(b)] = probe_cross_attention(eng, dims, len(b))[0]
    per_token_launches = dims["n_text_layer"] * args.new_tokens
    kernel_s = sum(per_token_launches * per_size[len(b)] * 1e-3 for b in batches)
    roofline = {"bound": "hbm", "kernel": "attn_decoder_cross_kernel", "achieved": round(k_gbs, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(k_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                "traffic_note": "not measurable in-process; rocprofv3 --pmc pass of this command under profiles/",
                "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_launch": k_bytes,
                "windows_per_launch": n_b, "launches_per_step_and_rank": per_token_launches * len(batches),
                "kernel_ms_by_windows_per_launch": {str(k): round(v, 4) for k, v in sorted(per_size.items())},
                "share_of_step": round(kernel_s / (dt / steps), 3),
                "share_note": "sum over this rank's engine calls of launches x the kernel time measured at "
                              "that call's window count, over the step time"}
    if dims["n_text_state"] == 1280 and dims["n_text_layer"] == 32:
        for windows, name in ((n_b, "decode%d_pmc_hbm.csv" % n_b), (256, "decode256_pmc_hbm.csv")):
            t = measured_traffic(name, "void attn_decoder_cross_kernel<1, 4>")
            if t is not None:
                roofline["traffic"] = int(round(t * n_b / float(windows)))
                roofline["traffic_note"] = (
                    "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes measured %d bytes per %d-window launch "
                    "of this kernel (profiles/, %s)%s; the counters cannot be read from inside this process"
                    % (t, windows, name, "" if windows == n_b else ", scaled by windows per launch / %d" % windows))
                break
    step_bytes = token_step_bytes(dims, n_b)
    line = {
        "metric": "audio-sec transcribed/sec, whisper-%s greedy" % args.model,
        "value": round(audio_s / dt, 1), "unit": "audio-s/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "whisper-%s greedy, %d x 30 s synthetic windows (%.3g h) dealt round-robin to "
                   "%d rank(s), <= %d windows per engine call, %d new tokens per window, token lists gathered "
                   "on every rank (BASELINE.json configs[3])" % (args.model, args.windows, args.windows / 120.0,
                                                                 world, per_call, args.new_tokens),
                   "audio_seconds_per_step": args.windows * 30.0, "windows_per_call": per_call,
                   "calls_per_step_and_rank": len(batches),
                   "note": "log-mel of all of a rank's windows first, clamp maximum over the whole job (one float "
                           "all-reduced over the ranks), then encoder + generation per engine call; synthetic "
                           "'peaked' weights (tokens depend on audio and history)"},
        "audio_hours_per_hour": round(audio_s / dt, 1),
        "roofline": roofline,
        "phases_rank0": {
            "windows": n_b, "logmel_ms": round(mel_ms, 3), "encoder_ms": round(enc_ms, 2),
            "generate_ms": round(gen_ms, 2), "ms_per_token": round(tok_ms, 3),
            "roofline_encoder": {"bound": "mfma", "achieved": round((f_enc + f_cross) / (enc_ms * 1e-3) / 1e12, 1),
                                 "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round((f_enc + f_cross) / (enc_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                 "scope": "whole encoder pass + cross K/V (all kernels), algorithmic FLOPs"},
            "roofline_token_step": {"bound": "hbm", "achieved": round(step_bytes / (tok_ms * 1e-3) / 1e9, 1),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(step_bytes / (tok_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "scope": "whole token step (all kernels, host loop included): decoder "
                                    "weights + cross K/V", "algorithmic_bytes_per_step": step_bytes}},
    }

    # ---- other call sizes on the same engine, this rank alone: 32 windows (the batch BASELINE.json's
    # configs[3] names) and 150 (what each of 8 ranks gets per call from the 1 200 windows)
    for key, nw in (("batch32", 32), ("batch150", 150)):
        nw = min(nw, n_b)
        say("leg %s" % key)
        try:
            run_batch(0, nw)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(2):
                run_batch(0, nw)
            torch.cuda.synchronize()
            tb = (time.perf_counter() - tb) / 2
            line[key] = {"value": round(nw * 30.0 / tb, 1), "unit": "audio-s/s per GPU",
                         "ms_per_call": round(tb * 1e3, 2), "windows": nw,
                         "note": "%d windows per engine call (log-mel + encoder + %d tokens), this rank alone"
                                 % (nw, args.new_tokens)}
        except Exception as e:                  # noqa: BLE001
            line[key] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- windows that end early stop costing attention traffic: half of a full call's windows are
    # given a limit of 40 new tokens (a stand-in for an EOS there: synthetic weights do not emit one
    # on schedule), early stopping on.  The first 40 tokens run with every window live, the rest with
    # half of them.
    try:
        say("leg early_end")
        short = min(40, args.new_tokens)
        caps = np.full(n_b, args.new_tokens, dtype=np.int32)
        caps[::2] = short
        pieces, clip_max = log_mel_pieces([(0, n_b)], reduce=False)
        eng.encode(finish_mel(pieces[0], clip_max, n_b))
        prompt = np.array([PROMPT] * n_b, dtype=np.int32)
        common = dict(max_new_tokens=args.new_tokens, eos_token=50257, pad_token=50257)
        times = {}
        for key, kw in (("all_live", dict(early_stopping=False)),
                        ("half_end_at_40", dict(early_stopping=True, max_new_tokens_per_window=caps))):
            eng.generate(prompt, **common, **kw)
            torch.cuda.synchronize()
            t = time.perf_counter()
            toks, lens, _ = eng.generate(prompt, **common, **kw)
            times[key] = (time.perf_counter() - t) * 1e3
        assert (lens[::2] == short).all() and (lens[1::2] == args.new_tokens).all()
        all_tok = times["all_live"] / args.new_tokens
        rest = max(1, args.new_tokens - short)
        line["early_end"] = {
            "windows": n_b, "ended_at_token_%d" % short: int((caps == short).sum()),
            "generate_ms_all_live": round(times["all_live"], 2), "generate_ms_half_ended": round(times["half_end_at_40"], 2),
            "ms_per_token_all_live": round(all_tok, 3),
            "ms_per_token_half_live": round((times["half_end_at_40"] - short * all_tok) / rest, 3),
            "note": "ended windows are skipped by the cross- and self-attention kernels (their K/V are not read); "
                    "the GEMMs still carry their rows"}
    except Exception as e:                  # noqa: BLE001
        line["early_end"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = whisper_cpu_baseline(args, dims, weights, check)
    # the pipelined leg builds its own two engines (calls of at most 256 windows so that both fit the card)
    del eng
    torch.cuda.empty_cache()
    pb = min(per_call, 256)
    pbatches = [mine[i: i + pb] for i in range(0, len(mine), pb)]
    if args.pipelined and len(pbatches) >= 2:
        try:                                # every rank raises together (all_ranks_ok)
            line["pipelined"] = bench_pipelined(args, dims, weights, fn, log_mel_pieces, finish_mel, decode, pbatches, world)
        except Exception as e:              # noqa: BLE001
            line["pipelined"] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    if args.beam_windows > 0:
        try:
            line["beam5"] = bench_beam(args, dims, weights, fn, pcm, min(args.beam_windows, len(mine)))
        except Exception as e:              # noqa: BLE001
            line["beam5"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return line


def bench_beam(args, dims, weights, fn, pcm, n_w, beams=5):
    """BASELINE configs[4] on this rank alone: beam search with 5 beams and the Whisper timestamp
    rules (prompt without the no-timestamps token), n_w windows per engine call, same number of new
    tokens, early stopping off.  Own engine (5 sequences per window), own roofline: the
    cross-attention kernel reads a window's K/V once for all of its beams."""
    from speech_to_text_amd.engine import TimestampFilter, WhisperEngine
    eng = WhisperEngine(dims, weights, max_batch=n_w, max_beams=beams,
                        max_text_len=args.new_tokens + 3, max_prompt_len=4)
    # ids as the reference's tokenizer has them (openai/whisper-base: timestamps from 50 364)
    remove = [1, 2, 7, 8, 9, 10, 14, 25, 50258, 50360, 50361, 50362, 50363]
    filt = TimestampFilter(remove_tokens=remove, remove_tokens_with_space=[220, 50257] + remove,
                           timestamp_begin_idx=50364, eos_token_idx=50257)
    prompt = np.array([PROMPT] * n_w, dtype=np.int32)

    def call():
        flat = pcm[:n_w].reshape(-1)
        mel = fn(flat)[0].view(n_w, 3000, dims["n_mels"])
        eng.encode(mel)
        return eng.generate(prompt, method="beam", num_beams=beams, num_sentences=1, max_new_tokens=args.new_tokens,
                            eos_token=50257, pad_token=50257, early_stopping=False, logits_filter=filt)
    call()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2):
        toks, lens, _ = call()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / 2
    t1 = time.perf_counter()
    eng.generate(prompt, method="beam", num_beams=beams, num_sentences=1, max_new_tokens=args.new_tokens,
                 eos_token=50257, pad_token=50257, early_stopping=False, logits_filter=filt)
    gen_ms = (time.perf_counter() - t1) * 1e3
    k_ms, k_bytes = probe_cross_attention(eng, dims, n_w, per_window=beams)
    gbs = k_bytes / (k_ms * 1e-3) / 1e9
    out = {"value": round(n_w * 30.0 / t, 1), "unit": "audio-s/s per GPU", "ms_per_call": round(t * 1e3, 2),
           "windows": n_w, "beams": beams, "ms_per_token": round(gen_ms / args.new_tokens, 3),
           "first_token_is_first_timestamp": bool((toks[:, 0, 0] == 50364).all()),
           "roofline": {"bound": "hbm", "kernel": "attn_decoder_cross_kernel (%d queries per window)" % beams,
                        "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "kernel_ms": round(k_ms, 4),
                        "algorithmic_bytes_per_launch": k_bytes, "windows_per_launch": n_w},
           "note_traffic": None,
           "note": "whisper-%s beam search, %d beams, timestamp rules, %d windows per engine call (log-mel + "
                   "encoder + %d tokens), this rank alone (BASELINE.json configs[4])"
                   % (args.model, beams, n_w, args.new_tokens)}
    if dims["n_text_state"] == 1280 and beams == 5:
        t = measured_traffic("beam100_pmc_hbm.csv", "attn_decoder_cross_mfma_kernel")
        if t is not None:
            out["roofline"]["traffic"] = int(round(t * n_w / 100.0))
            out["note_traffic"] = ("separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes measured %d bytes per "
                                   "100-window launch of this kernel (profiles/, beam100_pmc_hbm.csv)" % t)
    del eng
    return out


def whisper_cpu_baseline(args, dims, weights, check):
    """oracle/model.py (torch-CPU float32 restatement of the reference) on the engine's weights
    (rounded to bf16, as the engine holds them).  Timing: the encoder over the windows of
    `check["mel"]` and a bounded number of greedy tokens on the first of them, put together for the
    workload's tokens per window.  Check, on the FIRST and the LAST window of a full engine call:
    (a) the engine's teacher-forced logits over the prompt against the oracle's (relative L2, stated
    tolerance 2e-2 for bf16 storage); (b) the oracle teacher-forced on the ids the GPU decoded: at
    every step the GPU's id must be the oracle's argmax or lose to it by less than the near-tie
    allowance (0.04 x the standard deviation of the oracle's logits row); (c) the GPU's summed
    log-probabilities against the oracle's sum over the same ids."""
    from oracle import model as om
    threads = host_threads()
    torch.set_num_threads(threads)
    odims = om.Dims(**{k: dims[k] for k in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head",
                                            "n_audio_layer", "n_vocab", "n_text_ctx", "n_text_state",
                                            "n_text_head", "n_text_layer")})
    oracle = om.Whisper(odims, {k: v.to(torch.bfloat16).float() for k, v in weights.items()})
    mels = check["mel"]
    n_w = mels.shape[0]
    ids = torch.from_numpy(np.asarray(check["tokens"])).long()
    n_tok = ids.shape[1]
    with torch.no_grad():
        encs, t_enc = [], 1e30
        for i in range(n_w):                 # one window at a time; the first doubles as the warm-up: best of n_w
            t = time.perf_counter()
            encs.append(oracle.encoder(mels[i: i + 1]))
            t_enc = min(t_enc, time.perf_counter() - t)
        enc = torch.cat(encs)
        t = time.perf_counter()
        om.greedy(oracle, enc=enc[:1], prompt=torch.tensor([PROMPT]), max_new_tokens=n_tok, eos=50257,
                  pad=50257, early_stopping=False)
        t_dec = time.perf_counter() - t
        cross = oracle.cross_kv(enc)
        inputs, cache, offset = torch.tensor([PROMPT] * n_w), None, 0
        exact = near = wrong = 0
        worst, rel_l2 = 0.0, None
        want_scores, slack = np.zeros(n_w), np.zeros(n_w)
        for s in range(n_tok):
            logits, cache = oracle.decoder(inputs, cross, cache, offset)
            offset += inputs.shape[1]
            if s == 0:                       # the prompt pass: logits of every prompt position
                got = check["prompt_logits"].double()
                rel_l2 = float((got - logits.double()).norm() / logits.double().norm())
            lp = torch.log_softmax(logits[:, -1], dim=-1)
            for i in range(n_w):
                allow = 0.04 * float(logits[i, -1].std())
                best = int(lp[i].argmax())
                want_scores[i] += float(lp[i, ids[i, s]])
                slack[i] += allow / 2
                if best == int(ids[i, s]):
                    exact += 1
                    continue
                lost = float(lp[i, best] - lp[i, ids[i, s]]) / allow
                worst = max(worst, lost)
                near, wrong = near + (lost < 1.0), wrong + (lost >= 1.0)
            inputs = ids[:, s: s + 1]
    score_gap = np.abs(want_scores - np.asarray(check["scores"], dtype=np.float64))
    per_window = t_enc + t_dec * args.new_tokens / n_tok
    scores_ok = bool((score_gap < slack + 0.02).all())
    return {"value": round(30.0 / per_window, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "seconds": round(t_enc * n_w + t_dec, 2), "host": host_cpu(),
            "sample": "encoder of one window %.1f s (best of %d) + cross K/V and %d greedy tokens of one window %.1f s "
            "measured (torch-CPU fp32, oracle/model.py, the engine's bf16-rounded weights); value = 30 s / "
            "(encoder + tokens scaled to %d)" % (t_enc, n_w, n_tok, t_dec, args.new_tokens),
            "gpu_check": {"windows": "first and last window of a full engine call",
                          "prompt_logits_rel_l2": round(rel_l2, 5), "prompt_logits_tolerance": 2e-2,
                          "steps_checked": n_w * n_tok, "equal_oracle_argmax": exact,
                          "within_near_tie_allowance": near, "wrong": wrong,
                          "worst_loss_over_allowance": round(worst, 3),
                          "score_gap_max": round(float(score_gap.max()), 4), "score_gap_allowed": round(float(slack.max() + 0.02), 4),
                          "distinct_ids": len({int(x) for x in ids.reshape(-1)}),
                          "ok": bool(wrong == 0 and rel_l2 < 2e-2 and scores_ok)}}


def bench_pipelined(args, dims, weights, fn0, log_mel_pieces, finish_mel, decode, batches, world):
    """This rank's batches alternated over two engines built from the same weights, each on its own
    host thread and stream, the second one phase late: while one decodes a batch (HBM bound) the
    other runs the log-mel and the encoder of its next batch (matrix-core bound).  One pass over
    the rank's windows; the time includes the unoverlapped first and last phases."""
    import threading
    from speech_to_text_amd.engine import WhisperEngine
    dev = torch.device("cuda", torch.cuda.current_device())
    errors = []
    engs, fns = [], [fn0, fn0]               # log-mel plans are per stream

    def run_batch(lo, hi, engine, mel_fn, after_encode=None):
        pieces, clip_max = log_mel_pieces([(lo, hi)], mel_fn, reduce=False)
        engine.encode(finish_mel(pieces[0], clip_max, hi - lo, mel_fn))
        if after_encode is not None:
            ev = torch.cuda.Event()
            ev.record()
            after_encode(ev)
        return decode(engine, hi - lo)

    try:
        for r in range(2):
            engs.append(WhisperEngine(dims, weights, max_batch=len(batches[0]), max_beams=1,
                                      max_text_len=args.new_tokens + len(PROMPT) + 1, max_prompt_len=4))
            run_batch(0, len(batches[0]), engs[r], fns[r])       # warm the engine
        torch.cuda.synchronize()
    except Exception as e:                   # noqa: BLE001
        errors.append(e)
    if not all_ranks_ok(not errors, world):
        raise errors[0] if errors else RuntimeError("another rank failed to set up the pipelined leg")
    starts = np.cumsum([0] + [len(b) for b in batches])
    first, box = threading.Event(), {}

    def encoded(ev):
        box["event"] = ev
        first.set()

    def worker(r):
        try:
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(dev)):
                if r == 1:                  # one phase late: start when the first encoder pass is done
                    first.wait()
                    if "event" in box:
                        box["event"].synchronize()
                for i in range(r, len(batches), 2):
                    run_batch(int(starts[i]), int(starts[i + 1]), engs[r], fns[r],
                              after_encode=encoded if i == 0 else None)
        except BaseException as e:
            errors.append(e)
        finally:
            first.set()

    barrier(world)
    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier(world)
    if not all_ranks_ok(not errors, world):
        raise errors[0] if errors else RuntimeError("another rank failed in the pipelined leg")
    dt = max_over_ranks(dt, world)
    return {"value": round(args.windows * 30.0 / dt, 1), "unit": "audio-s/s", "replicas": 2, "seconds": round(dt, 3),
            "note": "one pass over the job, two engines per GPU, encode of one batch overlapped with decode "
            "of another (no token gather)"}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank, world, local = init_dist(args.dist_backend)
    if args.workload == "logmel":
        line = bench_logmel(args, rank, world, args.steps, args.warmup)
    else:
        line = bench_whisper(args, rank, world)
        if args.logmel_steps > 0:
            # the extra leg must never cost the headline line: report its failure instead
            try:
                line["logmel"] = bench_logmel(args, rank, world, args.logmel_steps, LOGMEL_WARMUP)
            except Exception as e:          # noqa: BLE001  (every rank raises together: all_ranks_ok)
                line["logmel"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
