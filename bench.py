#!/usr/bin/env python3
"""Benchmark of the MI355X Whisper hot path.

    python bench.py --gpus N --steps K --warmup W [--workload logmel]

Workload `logmel` (BASELINE.json configs[1]): log-mel frontend over 1 h of synthetic
16 kHz audio (uniform(-1,1) float32, already resident in HBM), 80 mel bins, fp32.
One step = one full pass (both kernels) over the clip.  With N > 1 every rank owns
its own 1 h clip (independent clips, no data-path collective): weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="logmel")
    ap.add_argument("--hours", type=float, default=1.0, help="audio hours per rank and step")
    ap.add_argument("--n-mels", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="large-v3", help="whisper leg / workload: model size")
    ap.add_argument("--batch", type=int, default=256, help="whisper: 30 s windows per step and GPU")
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--replicas", type=int, default=2,
                    help="whisper: 2 adds the pipelined two-engine measurement, 1 skips it")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument("--whisper-steps", type=int, default=2,
                    help="steps of the extra whisper leg reported beside the log-mel workload (0 = skip)")
    return ap.parse_args()


def init_dist(n, backend="nccl"):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())       # rehearsal on fewer GPUs than ranks
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return x


def all_ranks_ok(ok, world):
    """True when every rank says ok.  A collective that failed ranks reach too, so that a rank
    whose leg raised (out of memory, ...) does not leave the others waiting in a barrier."""
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([1 if ok else 0], dtype=torch.int32,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())
    return bool(ok)


# HBM bytes per logmel_main launch from the rocprofv3 PMC passes of this build (FETCH_SIZE x 2,
# the gfx950 correction, + WRITE_SIZE; profiles/r1/logmel_1h_pmc_hbm.csv).  Counters cannot be
# read from inside this process, so only the profiled configuration has a value.
MEASURED_HBM_BYTES = {(57_600_000, 80): 368884285}


def bench_logmel(args, rank, world):
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

    for _ in range(args.warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0            # this rank's K steps; the closing barrier is not timed
    barrier(world)
    dt = max_over_ranks(dt, world)

    alg_bytes = n * 4 + frames * args.n_mels * 4          # each sample read once, each mel written once
    value = world * args.steps * alg_bytes / dt / 1e9      # GB/s, whole job

    # dominant kernel (logmel_main), timed with events on the launching stream
    piece_max = ws[:4].view(torch.float32)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.steps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        _lib.check(lib.stt_logmel_stage1(plan, pcm.data_ptr(), 0, n, n, 0, frames, mel.data_ptr(),
                                         piece_max.data_ptr(), ws.data_ptr(), ws_bytes, stream))
        b.record()
    torch.cuda.synchronize()
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "logmel_main<%d>" % args.n_mels, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": MEASURED_HBM_BYTES.get((n, args.n_mels)), "kernel_ms": round(k_ms, 4),
                "algorithmic_bytes_per_launch": alg_bytes}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import frontend as oracle
        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        sample_s = 3600
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
               "seconds": round(best, 3)}
        # parity spot-check of the timed output against the oracle on that sample
        step()
        torch.cuda.synchronize()
        got = mel[: ref.shape[0] - 2].cpu().numpy()
        cpu["max_abs_diff_vs_gpu"] = float(np.abs(got - ref[: got.shape[0]]).max()) \
            if sample_s * 16000 == n else None

    line = {
        "metric": "log-mel GB/s (algorithmic bytes: PCM read once + mel written once)",
        "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "log-mel frontend, %.3g h of 16 kHz audio per GPU, %d mel bins, fp32 "
                   "(BASELINE.json configs[1])" % (args.hours, args.n_mels),
                   "audio_seconds_per_step": world * n / 16000.0},
        "audio_sec_per_sec": round(world * args.steps * (n / 16000.0) / dt, 1),
        "roofline": roofline, "cpu_baseline": cpu,
    }
    return line


MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense bf16


def whisper_flops(dims, batch, new_tokens, prompt_len):
    """Algorithmic FLOPs (2 per multiply-add) of one step: encoder + cross-K/V + decode."""
    d, L, T, M = dims["n_audio_state"], dims["n_audio_layer"], 1500, dims["n_mels"]
    stem = 2 * 3000 * 3 * M * d + 2 * T * 3 * d * d
    layer = 2 * T * (4 * d * d + 8 * d * d) + 4 * T * T * d
    enc = stem + L * layer
    cross = dims["n_text_layer"] * 2 * T * 2 * d * d
    tok = dims["n_text_layer"] * (2 * (4 * d * d + 2 * d * d + 8 * d * d) + 4 * T * d) + 2 * d * dims["n_vocab"]
    return batch * enc, batch * cross, batch * tok * (new_tokens + prompt_len - 1)


def decode_roofline(dims, batch, ms_per_token):
    """HBM roofline of one token step: the decoder's bf16 weights are read once per step, the
    cross-attention K/V of every window once per step (the dominant term), self K/V ignored."""
    d, L, V = dims["n_text_state"], dims["n_text_layer"], dims["n_vocab"]
    weights = (L * 14 * d * d + V * d) * 2
    cross_kv = batch * L * 2 * 1500 * d * 2
    achieved = (weights + cross_kv) / (ms_per_token * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "scope": "whole token step (all kernels), algorithmic bytes: decoder weights + cross K/V",
            "algorithmic_bytes_per_step": weights + cross_kv}


def bench_whisper(args, rank, world, steps, warmup, with_cpu):
    """Whisper greedy over `batch` synthetic 30 s windows per step: PCM (resident in HBM) ->
    log-mel -> encoder -> cross K/V -> `new_tokens` greedy tokens, early stopping off (random
    weights never emit EOS reliably).  BASELINE.json configs[3] per GPU (configs[2] with
    --model base)."""
    from speech_to_text_amd.audio.stft import WhisperSTFT
    from speech_to_text_amd.engine import WhisperEngine, whisper_dims
    from speech_to_text_amd.weights import random_whisper_weights

    dims = whisper_dims(args.model)
    prompt_row = [50258, 50259, 50359]
    err = None
    try:
        weights = random_whisper_weights(dims, 0)
        eng = WhisperEngine(dims, weights, max_batch=args.batch, max_beams=1,
                            max_text_len=args.new_tokens + len(prompt_row) + 1, max_prompt_len=4)
        fn = WhisperSTFT(n_mel_channels=dims["n_mels"])
        dev = torch.device("cuda", torch.cuda.current_device())
        rng = np.random.default_rng(100 + rank)
        pcm = torch.from_numpy(rng.uniform(-1, 1, args.batch * 480000).astype(np.float32)).to(dev)
        prompt = np.array([prompt_row] * args.batch, dtype=np.int32)

        def step():
            mel = fn(pcm)[0].view(args.batch, 3000, dims["n_mels"])       # 30 s = 3000 frames exactly
            eng.encode(mel)
            return eng.generate(prompt, max_new_tokens=args.new_tokens, eos_token=50257, pad_token=50257,
                                early_stopping=False)

        for _ in range(warmup):
            step()
    except Exception as e:                  # noqa: BLE001
        err = e
    if not all_ranks_ok(err is None, world):
        raise err if err is not None else RuntimeError("another rank failed to set up the whisper leg")
    barrier(world)
    t0 = time.perf_counter()
    try:
        for _ in range(steps):
            toks, lens, _ = step()
        torch.cuda.synchronize()
    except Exception as e:                  # noqa: BLE001
        err = e
    dt = time.perf_counter() - t0
    barrier(world)
    if not all_ranks_ok(err is None, world):
        raise err if err is not None else RuntimeError("another rank failed in the whisper leg")
    dt = max_over_ranks(dt, world)

    # encoder alone (MFMA-bound part), event-timed on the launching stream
    mel = fn(pcm)[0].view(args.batch, 3000, dims["n_mels"])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(); eng.encode(mel); b.record()
    torch.cuda.synchronize()
    enc_ms = a.elapsed_time(b)
    f_enc, f_cross, f_dec = whisper_flops(dims, args.batch, args.new_tokens, len(prompt_row))
    audio_s = world * steps * args.batch * 30.0
    out = {
        "metric": "audio-sec transcribed/sec, whisper-%s greedy" % args.model,
        "value": round(audio_s / dt, 1), "unit": "audio-s/s", "n_gpus": world, "steps": steps,
        "ms_per_step": round(dt / steps * 1e3, 2),
        "config": {"workload": "whisper-%s greedy, %d x 30 s synthetic windows per GPU and step, "
                   "%d new tokens, bf16" % (args.model, args.batch, args.new_tokens)},
        "dtype": "bf16", "data": "synthetic",
        "encoder_ms": round(enc_ms, 2),
        "roofline_encoder": {"bound": "mfma", "achieved": round((f_enc + f_cross) / (enc_ms * 1e-3) / 1e12, 1),
                             "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round((f_enc + f_cross) / (enc_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                             "scope": "whole encoder pass + cross K/V (all kernels), algorithmic FLOPs"},
        "decode_ms_per_token": round((dt / steps * 1e3 - enc_ms) / args.new_tokens, 3),
        "roofline_decode": decode_roofline(dims, args.batch, (dt / steps * 1e3 - enc_ms) / args.new_tokens),
        "tokens_generated_per_window": int(lens[0]),
    }
    if args.batch > 32:
        # the batch size BASELINE.json's configs[3] names (32 windows per GPU), on the same engine;
        # this rank's own time (no collective), scaled by the number of ranks
        try:
            p32 = prompt[:32]

            def step32():
                mel32 = fn(pcm[: 32 * 480000])[0].view(32, 3000, dims["n_mels"])
                eng.encode(mel32)
                eng.generate(p32, max_new_tokens=args.new_tokens, eos_token=50257, pad_token=50257,
                             early_stopping=False)

            step32()
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            for _ in range(2):
                step32()
            torch.cuda.synchronize()
            t32 = (time.perf_counter() - t32) / 2
            out["batch32"] = {"value": round(world * 32 * 30.0 / t32, 1), "unit": "audio-s/s",
                              "ms_per_step": round(t32 * 1e3, 2),
                              "note": "32 windows per GPU and step (BASELINE.json configs[3] batch size), rank 0's time"}
        except Exception as e:              # noqa: BLE001
            out["batch32"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if with_cpu and rank == 0 and world == 1:
        from oracle import model as om
        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        torch.set_num_threads(threads)
        odims = om.Dims(**{k: dims[k] for k in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head",
                                                "n_audio_layer", "n_vocab", "n_text_ctx", "n_text_state",
                                                "n_text_head", "n_text_layer")})
        oracle = om.Whisper(odims, weights)
        mel1 = mel[:1].cpu()
        n_tok = 8
        with torch.no_grad():
            t = time.perf_counter()
            ref = om.greedy(oracle, mel1, torch.tensor([prompt_row]), max_new_tokens=n_tok, eos=50257,
                            pad=50257, early_stopping=False)
            cpu_s = time.perf_counter() - t
        # extrapolate the decode part to new_tokens tokens: report the measured sample as is
        out["cpu_baseline"] = {"value": round(30.0 / cpu_s, 3), "unit": "audio-s/s", "cores": threads,
                               "kind": "port", "seconds": round(cpu_s, 2),
                               "sample": "1 window, encoder + %d greedy tokens (not %d), torch-CPU fp32 "
                               "restatement (oracle/model.py) on the same weights" % (n_tok, args.new_tokens),
                               "first_tokens_match_gpu": bool((ref["tokens"][0].numpy() == toks[0][:n_tok]).all())}
    if args.replicas >= 2:
        try:                                # every rank raises together (all_ranks_ok)
            out["pipelined"] = bench_whisper_pipelined(args, dims, weights, eng, fn, pcm, prompt, world, steps)
        except Exception as e:              # noqa: BLE001
            out["pipelined"] = {"error": "%s: %s" % (type(e).__name__, e)}
    del eng
    return out


def bench_whisper_pipelined(args, dims, weights, eng0, fn0, pcm, prompt, world, steps):
    """The same step on two engines built from the same weights, each on its own host thread and
    stream, the second one phase late: while one decodes a batch (HBM bound) the other runs the
    log-mel and the encoder of its next batch (matrix-core bound).  Every replica does `steps`
    steps; the time includes the unoverlapped first and last phases."""
    import threading
    from speech_to_text_amd.audio.stft import WhisperSTFT
    from speech_to_text_amd.engine import WhisperEngine
    dev = pcm.device
    first = threading.Event()
    box, errors = {}, []
    engs, fns = [eng0], [fn0]
    try:
        engs.append(WhisperEngine(dims, weights, max_batch=args.batch, max_beams=1,
                                  max_text_len=args.new_tokens + 8, max_prompt_len=4))
        fns.append(WhisperSTFT(n_mel_channels=dims["n_mels"]))    # a log-mel plan serves one stream
    except Exception as e:                   # noqa: BLE001
        errors.append(e)

    def one(r, signal=False):
        mel = fns[r](pcm)[0].view(args.batch, 3000, dims["n_mels"])
        engs[r].encode(mel)
        if signal:
            box["event"] = torch.cuda.Event()
            box["event"].record()
            first.set()
        engs[r].generate(prompt, max_new_tokens=args.new_tokens, eos_token=50257, pad_token=50257,
                         early_stopping=False)

    def worker(r, n):
        try:
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(dev)):
                if r == 1:
                    first.wait()
                    if "event" in box:
                        box["event"].synchronize()
                for i in range(n):
                    one(r, signal=(r == 0 and i == 0))
        except BaseException as e:
            errors.append(e)
        finally:
            first.set()

    try:
        if not errors:
            one(1)                           # warm the second engine (the first one is warm)
            torch.cuda.synchronize()
    except Exception as e:                   # noqa: BLE001
        errors.append(e)
    if not all_ranks_ok(not errors, world):
        raise errors[0] if errors else RuntimeError("another rank failed to set up the pipelined leg")
    barrier(world)
    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(r, steps)) for r in range(2)]
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
    return {"value": round(world * 2 * steps * args.batch * 30.0 / dt, 1), "unit": "audio-s/s",
            "replicas": 2, "steps_per_replica": steps, "seconds": round(dt, 3),
            "note": "two engines per GPU, encode of one batch overlapped with decode of the other"}


def main():
    args = parse()
    rank, world, local = init_dist(args.gpus, args.dist_backend)
    if args.workload == "logmel":
        line = bench_logmel(args, rank, world)
        if args.whisper_steps > 0:
            # the extra leg must never cost the headline line: report its failure instead
            try:
                line["whisper"] = bench_whisper(args, rank, world, args.whisper_steps, 1,
                                                not args.no_cpu_baseline)
            except Exception as e:          # noqa: BLE001  (every rank raises together: all_ranks_ok)
                line["whisper"] = {"error": "%s: %s" % (type(e).__name__, e)}
    else:
        w = bench_whisper(args, rank, world, args.steps, args.warmup, not args.no_cpu_baseline)
        line = dict(w, warmup=args.warmup, higher_is_better=True, scaling="weak", vs_baseline=None)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
