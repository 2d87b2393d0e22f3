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
    return ap.parse_args()


def init_dist(n):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return x


def bench_logmel(args, rank, world):
    from speech_to_text_amd import _lib
    from speech_to_text_amd.audio.stft import WhisperSTFT

    lib = _lib.load()
    n = int(round(57_600_000 * args.hours))
    fn = WhisperSTFT(n_mel_channels=args.n_mels)
    dev = torch.device("cuda", torch.cuda.current_device())
    # synthetic PCM generated on the host with the survey's generator, then made resident
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
    barrier(world)
    dt = time.perf_counter() - t0
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
                "traffic": None, "kernel_ms": round(k_ms, 4),
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


def main():
    args = parse()
    rank, world, local = init_dist(args.gpus)
    if args.workload == "logmel":
        line = bench_logmel(args, rank, world)
    else:
        from speech_to_text_amd import bench_model
        line = bench_model.run(args, rank, world, barrier, max_over_ranks)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
