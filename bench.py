#!/usr/bin/env python3
"""Headline benchmark: zeroSpan spectrum/waterfall hot path at BASELINE.json configs[1]
(fftSize 4096, 50 % overlap, hanning, complex64 synthetic IQ), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (IQ -> window -> FFT -> |X| -> fold -> dB -> Cur/Max/Min/Avg
+ waterfall rows) over one HBM-resident batch of `--frames` capture blocks per GPU.  With N > 1 every rank
owns a contiguous time chunk of the run (weak scaling: frames per GPU fixed), and the global Max/Min/Avg/Cur
curves and the waterfall ring come from ONE RCCL all-gather + a local merge kernel per step (distributed.py).  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FFT, FULL, Q, WINDOW, XRES, GAIN = 4096, 32768, 0.5, "hanning", 512, 19.1
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_frame(sample_bytes, hm_w):
    """one read of every IQ sample, one write of the Cur curve, one waterfall row."""
    return FULL * sample_bytes + 4 * N_FFT + 4 * hm_w


def cpu_baseline(seconds=12.0):
    """The float64 numpy oracle (a port of the reference's numpy path) on one host core, on a bounded
    sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ksa_oracle as orc
    win = orc.window_table(WINDOW, N_FFT)
    nfr = 64
    x = orc.synth_iq(FULL * nfr, 20201226 + 2).astype(np.complex64).reshape(nfr, FULL)
    st = orc.ZeroSpanState(N_FFT, XRES, GAIN)
    nwin = len(orc.window_starts(FULL, N_FFT, Q))
    done = 0
    t0 = time.perf_counter()
    while True:
        for fr in x:
            st.push(orc.curscan(fr, N_FFT, Q, win, "AVG"))
        done += nfr
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"value": done * nwin / dt, "unit": "FFT/s", "cores": 1, "kind": "port",
            "sample": "%d frames of 32768 complex samples (%d FFTs) in %.1f s, numpy float64, 1 thread of %d usable"
                      % (done, done * nwin, dt, len(os.sched_getaffinity(0)))}


def _cpu_worker(args):
    seconds, seed = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ksa_oracle as orc
    win = orc.window_table(WINDOW, N_FFT)
    x = orc.synth_iq(FULL * 16, seed).astype(np.complex64).reshape(16, FULL)
    st = orc.ZeroSpanState(N_FFT, XRES, GAIN)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for fr in x:
            st.push(orc.curscan(fr, N_FFT, Q, win, "AVG"))
        done += 16
    return done, time.perf_counter() - t0


def cpu_baseline_multicore(seconds=6.0, max_workers=16):
    """the same numpy port on independent frames, one plain child process per usable core (capped
    at the box's CPU share) -- the non-target multi-core figure.  Children never touch the GPU; any failure or
    timeout just drops the figure."""
    import subprocess
    workers = max(1, min(max_workers, len(os.sched_getaffinity(0))))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(seconds), str(100 + i)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for i in range(workers)]
    res = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=seconds + 60)
            done, dt = out.strip().split()[-2:]
            res.append((int(done), float(dt)))
    except Exception:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        return None
    frames = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    return {"value": frames * 15 / dt, "unit": "FFT/s", "cores": workers, "kind": "port",
            "sample": "%d frames over %d processes in %.1f s, numpy float64" % (frames, workers, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=65536, help="capture blocks per GPU per step (65536 = 16 GiB of complex64 resident in HBM)")
    ap.add_argument("--fmt", choices=("c64", "u8"), default="c64")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1 only: run the multi-GPU merge path (all-gather + merge kernel) on a one-rank group, to price its fixed cost")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): native libraries that print to fd 1 (RCCL's version banner
    # does) are sent to stderr for the whole run, the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
    ksa_dist = importlib.import_module("prgs-sdr-kspecanal_amd.distributed")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch N > 1 through torch.distributed.run" % (args.gpus, world))
    # one rank per GPU; KSA_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the launch path
    backend = os.environ.get("KSA_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    elif args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group(backend, rank=0, world_size=1, device_id=torch.device("cuda", local) if backend == "nccl" else None)

    frames = args.frames
    sb = 8 if args.fmt == "c64" else 2
    fmt = ksa.FMT_C64 if args.fmt == "c64" else ksa.FMT_U8
    # synthetic input: 256 distinct frames generated on the host, tiled in HBM (content does not affect timing)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ksa_oracle as orc   # synthetic source only; nothing of the oracle is timed on the GPU leg
    distinct = min(frames, 256)
    host = orc.synth_iq(FULL * distinct, 20201226 + 2 + rank).astype(np.complex64)
    if args.fmt == "c64":
        tile = torch.view_as_real(torch.from_numpy(host)).reshape(distinct, FULL, 2).cuda()
    else:
        tile = torch.from_numpy(orc.quantize_u8(host * 0.8)).reshape(distinct, FULL * 2).cuda()
    reps = (frames + distinct - 1) // distinct
    iq = tile.repeat(reps, *([1] * (tile.dim() - 1)))[:frames].contiguous()
    del tile

    eng = ksa.SpectrumEngine(N_FFT, full_size=FULL, non_overlap=Q, window=WINDOW, gain=GAIN, xres=XRES,
                             max_frames=frames, device=local, stream=torch.cuda.current_stream().cuda_stream)
    cur_db = torch.empty((frames, N_FFT), dtype=torch.float32, device="cuda")
    hm_rows = torch.empty((frames, eng.hm_width), dtype=torch.float32, device="cuda")
    run = ksa_dist.ShardedZeroSpan(eng, rank, world, always_collective=args.force_collective)

    def step():
        run.step(iq, fmt, frames, cur_db=cur_db, hm_rows=hm_rows)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kern_ms, launches = eng.prof_read()
    eng.prof_enable(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        nwin = eng.num_windows
        total_frames = frames * world * args.steps
        ffts_per_s = total_frames * nwin / dt
        info = eng.kernel_info()
        bpf = algorithmic_bytes_per_frame(sb, eng.hm_width)
        avg_kernel_s = kern_ms / 1e3 / max(1, launches)
        achieved = frames * bpf / avg_kernel_s / 1e9
        out = {
            "metric": "windowed FFTs/sec, fftSize=4096, 50% overlap, hanning (zeroSpan hot path incl. Max/Min/Avg/Cur + waterfall)",
            "value": ffts_per_s, "unit": "FFT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "msamples_per_s": total_frames * FULL / dt / 1e6,
            "config": {"workload": "configs[1]: zeroSpan synthetic 2.4 MS/s IQ, fftSize=4096, 50% overlap, hanning",
                       "input": "complex64" if args.fmt == "c64" else "uint8", "frames_per_gpu_per_step": frames,
                       "samples_per_frame": FULL, "windows_per_frame": nwin, "sharding": "time-chunk",
                       "collective": "RCCL: 1 all-gather of [4N + 128W] floats per rank per step, merged by ksa_merge_gathered_dev" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "ksa::spectrum_kernel<4096,%s>" % ("c64" if args.fmt == "c64" else "u8"),
                         "avg_kernel_ms": avg_kernel_s * 1e3, "launches": launches,
                         "algorithmic_bytes_per_launch": frames * bpf,
                         "threads": info["threads"], "lds_bytes": info["lds_bytes"], "vgprs": info["vgprs"],
                         "grid": min(frames, info["grid"])},
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("frames") == frames and rec.get("fmt") == args.fmt:
                    out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
            except Exception:
                pass
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            multi = cpu_baseline_multicore()
            if multi is not None:
                out["cpu_baseline_multicore"] = multi
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-worker":     # child of cpu_baseline_multicore
        print(*_cpu_worker((float(sys.argv[2]), int(sys.argv[3]))))
    else:
        main()
