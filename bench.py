#!/usr/bin/env python3
"""Benchmark of the spectrum / waterfall hot path on MI355X, one process per GPU.

    python bench.py                                  # headline: BASELINE configs[1], 1 GPU
    python bench.py --config {2,3,4,5} --gpus N --steps K --warmup W

--config picks the BASELINE.json configuration (1-based, as SURVEY.md section 8 numbers them):
  2  zeroSpan, fftSize 4096, 50 % overlap, hanning, complex64 synthetic IQ      (headline, default)
  3  fmScan 88-108 MHz, fftSize 16384 kaiser, 18 steps x 71 windows per pass, stitch + Max/Min/Avg + waterfall
  4  quickFullScan shape (30e6-1.5e9, fftSize 64, 1226 steps per pass), band shard + RCCL gather of the spectra
  5  zeroSpan, fftSize 65536, 75 % overlap (radix-16 first stage + 4096-point kernel), time-chunk shard + RCCL merge of Max/Min/Avg

One "step" = one pass of the whole hot path over one HBM-resident batch per GPU: `--frames` capture blocks
(zeroSpan: IQ -> window -> FFT -> |X| -> fold -> dB -> Cur/Max/Min/Avg + waterfall rows) or `--passes` whole scan
passes (every tuned band's block -> spectrum -> clip/dB, then stitch + Max/Min/Avg + waterfall row per pass).
With N > 1 every rank owns a contiguous time chunk (zeroSpan) or a contiguous range of tuned bands (scan) and ONE
RCCL all-gather per step carries what the ranks must share (distributed.py): the zeroSpan configs scale weakly (every
GPU brings its own time chunk), the scan configs strongly (one fixed range split over the GPUs; halo exchange between
neighbours + one all-gather of the partial waterfall rows).

N > 1 lines prove their own topology: `ranks` (backend, world size, RCCL version and per rank host / device ordinal /
PCI bus id / uuid, gathered from the ranks themselves), `state_identical_across_ranks` (sha256 of every rank's
Cur/Max/Min/Avg + waterfall ring after the timed region) and, for the zeroSpan configs, a `strong` sub-record (the
1-GPU batch split over the ranks) next to the weak headline.  `--inprocess` times the torch-free form instead: ONE process,
one engine per visible GPU, merged by ksa_allreduce_state / ksa_scan_allstitch (peer copies, no RCCL).

Launch: under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or plainly as
`python bench.py --gpus N`: the parent then starts N fresh rank processes itself (before it has touched the GPU),
relays rank 0's JSON line and fails if any rank fails.  Rank 0 prints ONE JSON line on stdout.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3   # vector fp32 peak, same guide (the FFT is not an MFMA shape)
GAIN = 19.1
FS = 2.4e6

# name, geometry (SURVEY.md section 8 sizes), default batch per GPU per step
CONFIGS = {
    2: dict(workload="configs[1]: zeroSpan synthetic 2.4 MS/s complex64 IQ, fftSize=4096, 50% overlap, hanning",
            metric="windowed FFTs/sec, fftSize=4096, 50% overlap, hanning (zeroSpan hot path incl. Max/Min/Avg/Cur + waterfall)",
            mode="zerospan", n=4096, q=0.5, window="hanning", full=32768, xres=512, frames=65536),
    3: dict(workload="configs[2]: scan 88-108 MHz stepped (fmScan), fftSize=16384 kaiser, min/max/avg + waterfall accumulate",
            metric="windowed FFTs/sec, fmScan: fftSize=16384 kaiser, 18 steps x 71 windows per pass (scan hot path incl. stitch + Max/Min/Avg + waterfall)",
            mode="scan", n=16384, q=0.1, window="kaiser", full=131072, xres=512, start=88e6, end=108e6, passes=256),
    4: dict(workload="configs[3]: quickFullScan shape (30e6-1.5e9, fftSize=64), freq-band shard, RCCL gather of spectrum",
            metric="windowed FFTs/sec, quickFullScan: fftSize=64, 1226 steps x 71 windows per pass (scan hot path incl. stitch + Max/Min/Avg + waterfall)",
            mode="scan", n=64, q=0.1, window="ones", full=512, xres=64, start=30e6, end=1.5e9, passes=1024),
    5: dict(workload="configs[4]: fftSize=65536, 75% overlap, synthetic IQ, time-chunk shard with RCCL merge of min/max/avg",
            metric="windowed FFTs/sec, fftSize=65536, 75% overlap, hanning (zeroSpan hot path incl. Max/Min/Avg/Cur + waterfall)",
            mode="zerospan", n=65536, q=0.25, window="hanning", full=524288, xres=512, frames=2048),
}


def _oracle():
    """oracle/ is test infrastructure: bench.py uses it for the synthetic source and as the timed CPU
    baseline leg only -- nothing of it runs inside the GPU-timed region."""
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import ksa_oracle
    return ksa_oracle


def scan_geometry(cfg):
    orc = _oracle()
    end, _ = orc.fixup_scan_range(cfg["start"], cfg["end"], FS)
    steps = len(orc.scan_steps(cfg["start"], end, FS, 0.5))
    groups = int((end - cfg["start"]) / FS)
    return end, steps, groups * cfg["n"]


def algorithmic_bytes(cfg, sample_bytes, nwin):
    """SURVEY.md 8(d), bytes per unit (frame or tuned band): one read of every IQ sample + one write of the unit's
    spectrum (+ one waterfall row per frame in zeroSpan; per pass in scan, amortised over its steps)."""
    b = cfg["full"] * sample_bytes + 4 * cfg["n"]
    if cfg["mode"] == "zerospan":
        b += 4 * min(cfg["n"], cfg["xres"])
    return b


def algorithmic_flops_per_fft(n):
    """5 N log2 N butterflies + ~12 N for window, magnitude, fold and dB."""
    return 5.0 * n * np.log2(n) + 12.0 * n


# ------------------------------------------------------------------------------------------- CPU baseline
def _cpu_unit_fn(cfg):
    """One unit of the workload on the float64 numpy port (the oracle): a zeroSpan frame incl. the accumulate and
    the waterfall row, or one tuned band of a scan incl. clip + dB (the per-pass stitch is timed with it)."""
    orc = _oracle()
    n, q, full = cfg["n"], cfg["q"], cfg["full"]
    win = orc.window_table(cfg["window"], n)
    if cfg["mode"] == "zerospan":
        st = orc.ZeroSpanState(n, cfg["xres"], GAIN)
        return lambda x: st.push(orc.curscan(x, n, q, win, "AVG")), None
    end, steps, _ = scan_geometry(cfg)
    state = orc.ScanState(n, cfg["start"], end, FS, GAIN, (1 / 256) * 0.00001, cfg["xres"])
    return (lambda x: orc.curscan(x, n, q, win, "AVG")), (state, steps)


def _cpu_rounds(cfg, seed, rounds, seconds):
    """`rounds` measurements of >= `seconds` each; returns [(units, dt)]."""
    orc = _oracle()
    full = cfg["full"]
    distinct = max(2, min(32, (8 << 20) // (full * 8)))
    x = orc.synth_iq(full * distinct, seed).astype(np.complex64).reshape(distinct, full)
    unit, scan = _cpu_unit_fn(cfg)
    out = []
    for _ in range(rounds):
        done, t0 = 0, time.perf_counter()
        pending = []
        while True:
            r = unit(x[done % distinct])
            done += 1
            if scan is not None:
                pending.append(r)
                if len(pending) == scan[1]:        # a whole pass captured: stitch + accumulate (K:621-668)
                    scan[0].run_pass(pending)
                    pending = []
            if (done & 3) == 0 and time.perf_counter() - t0 >= seconds:
                break
        out.append((done, time.perf_counter() - t0))
    return out


def cpu_baseline(cfg, nwin, seconds=3.0, rounds=5):
    """the numpy float64 port on ONE host core, median of `rounds` measurements of >= `seconds`."""
    res = _cpu_rounds(cfg, 20201226 + 2, rounds, seconds)
    rates = sorted(u * nwin / dt for u, dt in res)
    unit = "frames" if cfg["mode"] == "zerospan" else "tuned bands"
    return {"value": rates[len(rates) // 2], "unit": "FFT/s", "cores": 1, "kind": "port",
            "sample": "median of %d runs of >= %.1f s (%d..%d %s of %d complex samples each, %d FFTs of %d per unit), numpy float64, "
                      "1 thread of %d usable" % (rounds, seconds, min(u for u, _ in res), max(u for u, _ in res), unit,
                                                 cfg["full"], nwin, cfg["n"], len(os.sched_getaffinity(0)))}


def cpu_baseline_multicore(config_id, nwin, seconds=3.0, rounds=5, max_workers=16):
    """the same port on independent units, one plain child process per usable core -- capped at
    --cpu-workers (default 16: the CPU share and process budget a one-GPU box grants; `--cpu-workers 256` on a box that
    allows it gives the all-cores figure) -- the non-target multi-core figure.  Children never touch the GPU; any
    failure or timeout just drops the figure."""
    workers = max(1, min(max_workers, len(os.sched_getaffinity(0))))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(config_id), str(seconds),
                               str(rounds), str(100 + i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
             for i in range(workers)]
    per_worker = []
    try:
        for pr in procs:
            out, _ = pr.communicate(timeout=seconds * rounds + 120)
            per_worker.append(json.loads(out.strip().splitlines()[-1]))
    except Exception:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        return None
    rates = sorted(sum(w[r][0] for w in per_worker) * nwin / max(w[r][1] for w in per_worker) for r in range(rounds))
    return {"value": rates[len(rates) // 2], "unit": "FFT/s", "cores": workers, "kind": "port",
            "sample": "median of %d runs of >= %.1f s over %d processes (of %d usable cores; --cpu-workers %d: the default 16 is the CPU share and process budget of a one-GPU box, a stated deviation from SURVEY 8d's 'all usable cores'), numpy float64"
                      % (rounds, seconds, workers, len(os.sched_getaffinity(0)), max_workers)}


# ------------------------------------------------------------------------------------------- self launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh rank processes (this parent has not
    initialised the GPU and never does), relay rank 0's JSON line, fail if any rank fails."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), KSA_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad:                              # one rank died: the others would wait in a collective for ever
            rc = bad[0].returncode
            for p in procs:
                if p.poll() is None:
                    p.terminate()            # the exact PIDs started above
            break
        time.sleep(0.2)
    out = procs[0].communicate()[0]
    for p in procs[1:]:
        try:
            p.wait(timeout=60)
        except subprocess.TimeoutExpired:
            p.kill()
    rc = rc or next((p.returncode for p in procs if p.returncode), 0)
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    if rc or not lines:
        sys.stderr.write("bench.py: rank processes failed (rc %s)\n" % rc)
        raise SystemExit(rc or 1)
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()


SECONDARY_BUDGET_S = 150.0      # all side runs together; the driver's limit for the whole bench is 600 s


def secondary_configs(fmt, budget_s=SECONDARY_BUDGET_S):
    """The other BASELINE configurations (C3 fmScan, C4 quickFullScan shape, C5 fftSize 65536 at 75 % overlap) as
    short runs in child processes after the headline's timed region, so that the record the driver keeps carries a
    driver-run number for each of them too: {config: {value, unit, ms_per_step, roofline fractions} | {error}}.  The
    headline's own fields are untouched; `python bench.py --config K` gives the full line of configuration K.  The side
    runs share ONE deadline (a slow or hung one costs the others, never the headline line: what is left when it expires
    is recorded as skipped)."""
    res = {}
    t_end = time.monotonic() + budget_s
    # "2:u8": the headline geometry on the dongle's native uint8 I,Q (the unpack (b - 127.5) / 127.5 runs in the
    # kernel's load stage; algorithmic bytes with s = 2 bytes per sample) -- only when the headline itself ran complex64
    runs = ([("2:u8", 2, "u8")] if fmt == "c64" else []) + [(str(k), k, fmt) for k in (3, 4, 5)]
    for key, k, f in runs:
        left = t_end - time.monotonic()
        if left < 15.0:
            res[key] = {"skipped": "side-run budget of %.0f s used up" % budget_s}
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", str(k), "--fmt", f, "--steps", "10",
                                "--warmup", "2", "--no-cpu", "--no-secondary"], capture_output=True, text=True, timeout=left)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                res[key] = {"error": (r.stderr or "no output")[-300:]}
                continue
            d = json.loads(line[-1])
            rf = d["roofline"]
            res[key] = {"workload": d["config"]["workload"], "input": d["config"]["input"], "value": d["value"], "unit": d["unit"],
                        "ms_per_step": d["ms_per_step"], "msamples_per_s": d["msamples_per_s"], "steps": d["steps"], "kernel": rf["kernel"],
                        "avg_kernel_ms": rf["avg_kernel_ms"], "bound": rf["bound"], "limiter": rf["limiter"],
                        "algorithmic_bytes_per_unit": rf["algorithmic_bytes_per_unit"],
                        "frac": rf["frac"], "frac_step": rf["frac_step"], "flop_frac": rf["flop_frac"],
                        "flop_frac_at_clock": rf["flop_frac_at_clock"], "shader_clock_ghz_live": rf["shader_clock_ghz_live"],
                        "lds_frac": rf["lds_frac"], "valu_issue_frac": rf["valu_issue_frac"],
                        "traffic": rf["traffic"], "traffic_over_algorithmic": rf["traffic_over_algorithmic"]}
        except Exception as ex:      # a failing side run must never take the headline line with it
            res[key] = {"error": repr(ex)[:300]}
    return res


def csrc_sha256():
    """Hash of the kernel sources the library is built from: stamps profiles/pmc_traffic.json so that counter
    traffic measured on an older kernel is never reported for a newer one."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "prgs-sdr-kspecanal_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def pmc_record(key):
    """Counter figures of the separate rocprofv3 --pmc passes (tools/profile_bench.sh -> tools/summarize_profile.py) for
    this bench configuration, or {} when they were taken on other kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
        ent = rec.get("entries", {}).get(key)
        if ent and ent.get("csrc_sha256") == csrc_sha256():
            return ent
    except Exception:
        pass
    return {}


# What binds each configuration: `bound` names the resource, `limiter` says it in a sentence.  The north star
# prices every configuration against HBM (`frac`); the fp32 (`flop_frac`), VALU-issue and LDS-array fractions stand beside it.
# ONE source of truth: every decimal number in the sentence is formatted from a field of the same roofline block
# (tests/test_host_cli.py::test_limiter_quotes_its_own_fields), so the prose cannot go stale against the counters.
BOUND = {2: "hbm", 3: "valu", 4: "valu", 5: "hbm (Z round trip)"}
NOMINAL_CLOCK_GHZ = 2.4    # the clock the 157.3 TFLOP/s fp32 peak is quoted at (MI355X_MICROARCH.md)


def limiter_sentence(config, rf):
    """rf: the roofline block (frac, flop_frac and, when the stored counter passes match the kernel sources, valu_issue_frac /
    lds_frac / traffic_over_algorithmic).  Numbers are quoted with two decimals, straight from those fields."""
    have = all(rf.get(k) is not None for k in ("valu_issue_frac", "lds_frac", "traffic_over_algorithmic"))
    pipes = ("VALU issue %.2f of the issue slots, LDS array %.2f of its cycles, counter traffic %.2fx the algorithmic bytes"
             % (rf["valu_issue_frac"], rf["lds_frac"], rf["traffic_over_algorithmic"])) if have else \
        "counters not available for this source hash (re-run tools/profile_bench.sh)"
    head = {
        2: "priced against HBM as the north star asks (%.2f of its peak, fp32 %.2f of the vector peak); what limits it is VALU issue and the "
           "LDS exchanges running in series at 3 waves per SIMD",
        3: "90 %% window overlap: every sample is transformed ten times, one workgroup of 135 KB of LDS per CU at 2 waves per SIMD; HBM sees "
           "%.2f of its peak, fp32 %.2f of the vector peak",
        4: "90 %% window overlap at N = 64: 16 transforms per wave (8 x 8 plan), more than a third of a wave's time goes into issuing and awaiting "
           "the 8 adjacent-sample 16-byte loads of a round, a fifth into the exchange through LDS (cycle stamps, profiles/r05_c4_stamps_k64.txt); "
           "HBM sees %.2f of its peak, fp32 %.2f of the vector peak",
        5: "two streaming passes over the first-stage scratch Z: HBM sees %.2f of its peak in algorithmic bytes, fp32 %.2f of the vector peak",
    }[config] % (rf["frac"], rf["flop_frac"])
    return head + "; " + pipes


# ------------------------------------------------------------------------------------------- topology / state proofs
def backend_name(backend):
    """What the record calls the collective backend actually in use: torch's "nccl" IS RCCL on ROCm; gloo (the rehearsals
    of the launch path on shared devices) stages through the host."""
    return "RCCL (torch.distributed nccl backend)" if backend == "nccl" else "%s (host-staged; rehearsal backend)" % backend


def device_identity(torch, local):
    """What tells two GPUs apart: ordinal, marketing name, PCI bus id and uuid (whatever this torch build exposes)."""
    pr = torch.cuda.get_device_properties(local)
    ident = {"device": int(local), "name": pr.name, "host": socket.gethostname(), "pid": os.getpid()}
    for attr in ("pci_bus_id", "pci_device_id", "pci_domain_id", "uuid", "gcnArchName"):
        v = getattr(pr, attr, None)
        if v is not None:
            ident[attr] = str(v)
    return ident


def rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as ex:
        return "unavailable (%s)" % type(ex).__name__


def state_digest(st):
    """sha256 over Cur / Max / Min / Avg + the waterfall ring + its position: equal digests = the same bits."""
    h = hashlib.sha256()
    for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg", "fftHM"):
        h.update(np.ascontiguousarray(st[k], dtype=np.float64).tobytes())
    h.update(str(int(st["hm_index"])).encode())
    return h.hexdigest()


def ranks_block(torch, dist, backend, world, local):
    """Every rank reports itself; rank 0 receives the list.  `distinct_devices` counts (host, bus id / uuid / ordinal)."""
    mine = dict(device_identity(torch, local), rank=dist.get_rank(), local_rank=int(os.environ.get("LOCAL_RANK", "0")))
    every = [None] * world
    dist.all_gather_object(every, mine)
    key = lambda r: (r["host"], r.get("uuid") or r.get("pci_bus_id") or r["device"])
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": rccl_version(torch) if backend == "nccl" else None,
            "distinct_devices": len({key(r) for r in every}), "visible_devices_rank0": torch.cuda.device_count(), "per_rank": every}


# ------------------------------------------------------------------------------------------- the bench
def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2, help="BASELINE.json configuration (1-based)")
    ap.add_argument("--frames", type=int, default=None, help="zeroSpan configs: capture blocks per GPU per step")
    ap.add_argument("--passes", type=int, default=None, help="scan configs: whole passes per step")
    ap.add_argument("--fmt", choices=("c64", "u8"), default="c64")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="length of each of the 5 CPU measurements")
    ap.add_argument("--cpu-workers", type=int, default=16,
                    help="processes of the multi-core CPU figure (capped by the usable cores; 16 = the CPU share of a one-GPU box)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline run on one GPU: skip the short runs of the other BASELINE configurations (3, 4, 5)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1, zeroSpan only: run the multi-GPU merge path on a one-rank group, to price its fixed cost")
    ap.add_argument("--inprocess", action="store_true",
                    help="ONE process, --gpus engines placed on the visible devices (engine r on device r %% device_count), merged by "
                         "ksa_allreduce_state / ksa_scan_allstitch: the torch-free multi-GPU form (no RCCL)")
    return ap.parse_args()


def roofline_block(cfg, args, eng, units_per_step, kern_ms, launches, step_s):
    """algorithmic bytes over the spectrum stage's HIP-event time, with the fp32 / LDS / VALU-issue fractions
    and the counter traffic of the stored rocprofv3 passes beside it."""
    n, nwin = cfg["n"], eng.num_windows
    sb = 8 if args.fmt == "c64" else 2
    info = eng.kernel_info()
    bpu = algorithmic_bytes(cfg, sb, nwin)
    avg_kernel_s = kern_ms / 1e3 / max(1, launches)
    alg_bytes = units_per_step * bpu
    achieved = alg_bytes / avg_kernel_s / 1e9
    tflops = units_per_step * nwin * algorithmic_flops_per_fft(n) / avg_kernel_s / 1e12
    if info["path"] == 2:
        kernel = "ksa::dif16_kernel<%s> + ksa::spectrum_kernel<%d,c64> + ksa::dif16_finish_kernel (N = 16*%d)" % (args.fmt, n // 16, n // 16)
    elif info["path"] == 4:
        kernel = "ksa::spectrum_pair_kernel<%d,%s> (two frames per workgroup, packed fp32)" % (n, args.fmt)
    elif info["path"] == 5 and args.fmt == "c64":
        kernel = "ksa::spectrum64_kernel<c64> (8 x 8, adjacent-sample loads)"
    elif info["path"] == 3:
        kernel = "ksa::spectrum32_kernel<%d,%s>" % (n, args.fmt)
    else:
        kernel = "ksa::spectrum_kernel<%d,%s>" % (n, args.fmt)
    rec = pmc_record("%d:%s:%d" % (args.config, args.fmt, units_per_step))
    traffic = rec.get("hbm_bytes_per_launch")
    # shader clock held under the profiled spectrum stages of THIS run (stamp kernels around each stage: ksa_prof_clock)
    clock_live, clock_samples = eng.prof_clock()
    clock = clock_live or rec.get("shader_clock_ghz")
    rf = {"bound": BOUND[args.config], "limiter": None, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": rec.get("source"),
            "traffic_note": ("counter figures (traffic = 2*FETCH_SIZE + WRITE_SIZE per step; lds_frac = SQ_LDS_IDX_ACTIVE / CUs / kernel cycles; "
                             "valu_issue_frac = SQ_INSTS_VALU * 2 clk / SIMDs / kernel cycles; lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / "
                             "SQ_LDS_IDX_ACTIVE) come from the separate rocprofv3 --pmc passes stored in profiles/pmc_traffic.json for this "
                             "exact kernel source (sha256-checked); NOT measured by this run") if rec else None,
            # counter traffic (L2 <-> fabric; Infinity-Cache hits included) per step over the stage's time
            "traffic_gbs": (traffic / avg_kernel_s / 1e9) if traffic else None,
            "traffic_frac": (traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
            "kernel": kernel, "avg_kernel_ms": avg_kernel_s * 1e3, "launches": launches,
            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_unit": bpu,
            # the whole step (spectrum stage + accumulate/stitch + collective) against the same bytes
            "frac_step": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
            # what the chip is busy with: fp32 vector fraction (live), LDS-array and VALU-issue fractions (stored counters)
            "tflops": tflops, "flop_peak": FP32_PEAK_TFLOPS, "flop_frac": tflops / FP32_PEAK_TFLOPS,
            "lds_frac": rec.get("lds_frac"), "lds_conflict_ratio": rec.get("lds_conflict_ratio"),
            "valu_issue_frac": rec.get("valu_issue_frac"), "shader_clock_ghz": rec.get("shader_clock_ghz"),
            # live: median over XCDs and launches of d(s_memtime) / d(s_memrealtime) x 100 MHz around each timed spectrum stage
            "shader_clock_ghz_live": clock_live, "shader_clock_samples": clock_samples,
            "shader_clock_ghz_live_range": getattr(eng, "prof_clock_range", None),
            # the fp32 fraction of what the chip can issue at the clock it actually held (the peak is quoted at 2.4 GHz)
            "flop_frac_at_clock": (tflops / (FP32_PEAK_TFLOPS * clock / NOMINAL_CLOCK_GHZ)) if clock else None,
            "flop_frac_at_clock_source": ("shader_clock_ghz_live" if clock_live else "shader_clock_ghz (stored counter pass)") if clock else None,
            "threads": info["threads"], "lds_bytes": info["lds_bytes"], "vgprs": info["vgprs"], "grid": info["grid"]}
    rf["limiter"] = limiter_sentence(args.config, rf)
    return rf


def main():
    args = parse_args()
    if args.inprocess:
        return main_inprocess(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    # stdout carries exactly ONE line (the JSON): native libraries that print to fd 1 (RCCL's version banner
    # does) are sent to stderr for the whole run, the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
    ksa_dist = importlib.import_module("prgs-sdr-kspecanal_amd.distributed")
    orc = _oracle()   # synthetic source only on this leg

    cfg = dict(CONFIGS[args.config])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
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
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group(backend, rank=0, world_size=1, device_id=torch.device("cuda", local) if backend == "nccl" else None)

    n, full, q = cfg["n"], cfg["full"], cfg["q"]
    fmt = ksa.FMT_C64 if args.fmt == "c64" else ksa.FMT_U8
    stream = torch.cuda.current_stream().cuda_stream
    resident_iq = lambda units: make_resident_iq(torch, orc, args, cfg, units, rank)

    strong_step = None
    if cfg["mode"] == "zerospan":
        frames = args.frames or cfg["frames"]
        units_per_step = frames                       # per rank
        iq = resident_iq(frames)
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=cfg["window"], gain=GAIN, xres=cfg["xres"],
                                 max_frames=frames, device=local, stream=stream)
        cur_db = torch.empty((frames, n), dtype=torch.float32, device="cuda")
        hm_rows = torch.empty((frames, eng.hm_width), dtype=torch.float32, device="cuda")
        run = ksa_dist.ShardedZeroSpan(eng, rank, world, always_collective=args.force_collective)
        step = lambda: run.step(iq, fmt, frames, cur_db=cur_db, hm_rows=hm_rows)
        if world > 1 and frames // world >= 1:
            fs = frames // world                      # the SAME job as one GPU's step, split over the ranks
            strong_step = (fs, lambda: run.step(iq, fmt, fs, cur_db=cur_db, hm_rows=hm_rows))
        sharding = "time-chunk"
        collective = ("%s: 1 all-gather of [4N + 128W] floats per rank per step over %d ranks, merged by ksa_merge_gathered_dev"
                      % (backend_name(backend), world)) if world > 1 else "none"
        xb = (4 * n + 128 * eng.hm_width) * 4
        coll_bytes = {"allgather_send": xb, "allgather_recv": xb * (world - 1)} if world > 1 else {}
        scaling = "weak"                              # every GPU brings its own time chunk: work per GPU fixed
        batch = {"frames_per_gpu_per_step": frames}
    else:
        passes = args.passes or cfg["passes"]
        end, steps, total = scan_geometry(cfg)
        lo, hi = ksa_dist.step_range(steps, rank, world)
        mine = hi - lo
        units_per_step = passes * mine                # tuned bands this rank transforms per step
        iq = resident_iq(max(1, passes * mine))
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=cfg["window"], gain=GAIN, xres=cfg["xres"],
                                 max_frames=max(1, passes * mine), device=local, scan_total_entries=total,
                                 scan_non_overlap=0.5, stream=stream)
        run = ksa_dist.ShardedScan(eng, rank, world)
        step = lambda: run.run_passes(iq, fmt, steps, passes)
        sharding = "freq-band"
        collective = ("%s: halo send/recv of the overlap part of 1 band per pass to the right neighbour + 1 all-gather of the "
                      "partial waterfall rows [min(passes,128)][W] per step over %d ranks; curves stay sharded by stitched range"
                      % (backend_name(backend), world)) if world > 1 else "none"
        scaling = "strong"                            # the scan range is one fixed job split over the GPUs
        batch = {"passes_per_step": passes, "steps_per_pass": steps, "bands_on_rank0": mine, "total_entries": total}

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, steps_):
        """EXACTLY steps_ steps between two barrier + synchronize fences; the slowest rank's time."""
        fence()
        t0 = time.perf_counter()
        for _ in range(steps_):
            fn()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    fence()
    if cfg["mode"] == "scan":
        coll_bytes = run.collective_bytes() if world > 1 else {}
    eng.prof_enable(True)
    dt = timed(step, args.steps)
    kern_ms, launches = eng.prof_read()
    eng.prof_enable(False)

    # ---- after the timed region: does every rank hold the same result?  (scan: after assembling the sharded curves)
    proof = None
    if world > 1:
        st = run.gather_state(steps) if cfg["mode"] == "scan" else eng.state()
        digests = [None] * world
        dist.all_gather_object(digests, state_digest(st))
        proof = {"state_identical_across_ranks": len(set(digests)) == 1, "state_sha256_rank0": digests[0],
                 "ranks_differing_from_rank0": [r for r, dg in enumerate(digests) if dg != digests[0]]}
    strong = None
    if strong_step is not None:
        fs, sfn = strong_step
        for _ in range(args.warmup):
            sfn()
        sdt = timed(sfn, args.steps)
        strong = {"scaling": "strong", "what": "the one-GPU step of %d frames split over the %d ranks (%d each)" % (fs * world, world, fs),
                  "frames_per_gpu_per_step": fs, "value": fs * world * args.steps * eng.num_windows / sdt, "unit": "FFT/s",
                  "ms_per_step": sdt / args.steps * 1e3}
    ranks = ranks_block(torch, dist, backend, world, local) if world > 1 else None

    if rank == 0:
        nwin = eng.num_windows
        if cfg["mode"] == "zerospan":
            units_all = units_per_step * world * args.steps
        else:
            units_all = (args.passes or cfg["passes"]) * steps * args.steps
        ffts_per_s = units_all * nwin / dt
        step_s = dt / args.steps
        out = {
            "metric": cfg["metric"], "value": ffts_per_s, "unit": "FFT/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "msamples_per_s": units_all * full / dt / 1e6,
            "config": dict({"workload": cfg["workload"], "baseline_config": args.config,
                            "input": "complex64" if args.fmt == "c64" else "uint8", "samples_per_unit": full,
                            "windows_per_unit": nwin, "sharding": sharding, "collective": collective,
                            "collective_bytes_per_rank_per_step": coll_bytes}, **batch),
            "roofline": roofline_block(cfg, args, eng, units_per_step, kern_ms, launches, step_s),
        }
        if world > 1:
            out["ranks"] = ranks
            out.update(proof)
            out["value_is_rehearsal"] = ranks["distinct_devices"] < world     # ranks share devices: `value` is NOT a throughput
            out["multi_gpu_note"] = ("%d rank processes over %s; %d distinct device(s) among them%s" % (
                world, ranks["backend"], ranks["distinct_devices"],
                "" if ranks["distinct_devices"] == world else " -- a functional REHEARSAL of the launch path on shared devices, not a scaling measurement"))
            if strong is not None:
                out["strong"] = strong
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(cfg, nwin, args.cpu_seconds)
            multi = cpu_baseline_multicore(args.config, nwin, args.cpu_seconds, max_workers=args.cpu_workers)
            if multi is not None:
                out["cpu_baseline_multicore"] = multi
        if world == 1 and args.config == 2 and not args.no_secondary and not args.force_collective:
            out["secondary_configs"] = secondary_configs(args.fmt)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


def make_resident_iq(torch, orc, args, cfg, units, seed_rank, device=None):
    """`units` capture blocks in HBM, tiled from <= 64 MiB of distinct host-generated blocks (content does not
    affect timing; H2D is outside the timed region)."""
    full = cfg["full"]
    dev = "cuda" if device is None else "cuda:%d" % device
    distinct = max(1, min(units, 256, (64 << 20) // (full * 8)))
    host = orc.synth_iq(full * distinct, 20201226 + args.config + seed_rank).astype(np.complex64)
    if args.fmt == "c64":
        tile = torch.view_as_real(torch.from_numpy(host)).reshape(distinct, full, 2).to(dev)
    else:
        tile = torch.from_numpy(orc.quantize_u8(host * 0.8)).reshape(distinct, full * 2).to(dev)
    reps = (units + distinct - 1) // distinct
    return tile.repeat(reps, *([1] * (tile.dim() - 1)))[:units].contiguous()


def main_inprocess(args):
    """The torch-free multi-GPU form (SURVEY 8b allreduce_state(handles[], n)): ONE process drives --gpus engines, engine
    r on device r % device_count, each on its own stream; per step every engine runs its share (no commit) and the
    library merges them -- ksa_allreduce_state (time chunk: peer copies of the exchange blocks + merge kernel on every
    engine) or ksa_scan_allstitch (band shard: halo peer copies, range stitch, row merge).  torch only provides the
    resident input buffers here.  Timing: host clock around --steps steps, every engine synchronised on both sides."""
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
    ksa_dist = importlib.import_module("prgs-sdr-kspecanal_amd.distributed")
    orc = _oracle()
    cfg = dict(CONFIGS[args.config])
    world, ndev = args.gpus, torch.cuda.device_count()
    n, full, q = cfg["n"], cfg["full"], cfg["q"]
    fmt = ksa.FMT_C64 if args.fmt == "c64" else ksa.FMT_U8
    devs = [r % ndev for r in range(world)]
    engines, inputs, outs = [], [], []
    if cfg["mode"] == "zerospan":
        frames = args.frames or cfg["frames"]
        for r, dv in enumerate(devs):
            inputs.append(make_resident_iq(torch, orc, args, cfg, frames, r, device=dv))
            engines.append(ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=cfg["window"], gain=GAIN, xres=cfg["xres"],
                                              max_frames=frames, device=dv))
        total = frames * world
        hm = [0]

        def step():
            for r, eng in enumerate(engines):
                eng.set_hm_index((hm[0] + r * frames) % 128)
                eng.frames_dev(inputs[r], fmt, frames, first_index=r * frames, total_frames=total, commit=False)
            ksa.allreduce_state(engines, frames, hm[0])
            hm[0] = (hm[0] + total) % 128
        units_all_per_step, merge = frames * world, "ksa_allreduce_state"
        batch = {"frames_per_gpu_per_step": frames}
        scaling = "weak"
    else:
        passes = args.passes or cfg["passes"]
        end, steps, total = scan_geometry(cfg)
        for r, dv in enumerate(devs):
            lo, hi = ksa_dist.step_range(steps, r, world)
            mine = hi - lo
            inputs.append(make_resident_iq(torch, orc, args, cfg, max(1, passes * mine), r, device=dv))
            engines.append(ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=cfg["window"], gain=GAIN, xres=cfg["xres"],
                                              max_frames=max(1, passes * mine), device=dv, scan_total_entries=total, scan_non_overlap=0.5))
            outs.append(torch.empty((passes, max(mine, 1), n), dtype=torch.float32, device="cuda:%d" % dv))

        def step():
            for r, eng in enumerate(engines):
                lo, hi = ksa_dist.step_range(steps, r, world)
                if hi > lo:
                    eng.scan_spectra_dev(inputs[r], fmt, passes * (hi - lo), outs[r])
            ksa.scan_allstitch(engines, [outs[r] if ksa_dist.step_range(steps, r, world)[1] > ksa_dist.step_range(steps, r, world)[0] else None
                                         for r in range(world)], steps, passes)
        units_all_per_step, merge = passes * steps, "ksa_scan_allstitch"
        batch = {"passes_per_step": passes, "steps_per_pass": steps, "total_entries": total}
        scaling = "strong"

    def sync_all():
        for eng in engines:
            eng.synchronize()

    for dv in set(devs):
        torch.cuda.synchronize(dv)
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if cfg["mode"] == "zerospan":
        digests = [state_digest(eng.state()) for eng in engines]
    else:
        curves = ksa.scan_gather_state(engines, steps)
        digests = []
        for eng in engines:
            st = eng.scan_state()
            digests.append(state_digest(dict(curves, fftHM=st["fftHM"], hm_index=st["hm_index"])))
    nwin = engines[0].num_windows
    idents = [dict(device_identity(torch, dv), engine=r) for r, dv in enumerate(devs)]
    distinct = len(set(devs))
    out = {"metric": cfg["metric"], "value": units_all_per_step * args.steps * nwin / dt, "unit": "FFT/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "msamples_per_s": units_all_per_step * args.steps * full / dt / 1e6,
           "config": dict({"workload": cfg["workload"], "baseline_config": args.config, "input": "complex64" if args.fmt == "c64" else "uint8",
                           "driver": "inprocess: one process, %d engines, merged by %s (peer copies, no RCCL)" % (world, merge)}, **batch),
           "ranks": {"backend": "inprocess (hipMemcpyPeerAsync + events)", "world_size": world, "rccl_version": None,
                     "distinct_devices": distinct, "visible_devices_rank0": ndev, "per_rank": idents},
           "state_identical_across_ranks": len(set(digests)) == 1, "state_sha256_rank0": digests[0],
           "value_is_rehearsal": distinct < world,
           "multi_gpu_note": "%d engines on %d distinct device(s)%s" % (
               world, distinct, "" if distinct == world else " -- a functional REHEARSAL on shared devices, not a scaling measurement")}
    for eng in engines:
        eng.close()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    if len(sys.argv) == 6 and sys.argv[1] == "--cpu-worker":     # child of cpu_baseline_multicore
        cid, secs, rnds, seed = int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        print(json.dumps(_cpu_rounds(dict(CONFIGS[cid]), seed, rnds, secs)))
    else:
        main()
