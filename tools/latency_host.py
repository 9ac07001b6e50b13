#!/usr/bin/env python3
"""Host-pointer entry points (the literal drop-in path): frames per second of ksa_frame_c64 / ksa_curscan_c64 /
ksa_frame_u8 including the H2D copy, launches and the synchronise, one block per call."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ksa_oracle as orc
ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
for n, q in ((4096, 0.5), (16384, 0.1), (64, 0.1), (65536, 0.25)):
    full = orc.full_size(n, 2.4e6 if n < 65536 else 1e9)
    x = orc.synth_iq(full, 1).astype(np.complex64)
    raw = orc.quantize_u8(x * 0.8)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning")
    for name, fn, arg in (("frame c64", eng.frame, x), ("frame u8", eng.frame, raw), ("curscan c64", eng.curscan, x)):
        for _ in range(5):
            fn(arg)
        t0 = time.perf_counter(); reps = 200
        for _ in range(reps):
            fn(arg)
        dt = (time.perf_counter() - t0) / reps
        print("N=%-6d q=%-4s %-12s %8.1f us/block  %7.1f MS/s  (%d windows)" % (n, q, name, dt * 1e6, full / dt / 1e6, eng.num_windows))
    eng.close()

# scan passes from host memory (ksa_scan_pass_c64 / _u8): one pass per call, pageable and page-locked capture blocks
for name, n, q, start, end, win in (("fmScan", 16384, 0.1, 88e6, 108e6, "kaiser"), ("quickFullScan", 64, 0.1, 30e6, 1.5e9, "ones")):
    fs = 2.4e6
    end, _ = orc.fixup_scan_range(start, end, fs)
    steps = len(orc.scan_steps(start, end, fs, 0.5))
    total = int((end - start) / fs) * n
    full = orc.full_size(n, fs)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=win, xres=min(n, 512), max_frames=steps, scan_total_entries=total)
    x = np.tile(orc.synth_iq(full * 4, 2).astype(np.complex64).reshape(4, full), (steps // 4 + 1, 1))[:steps].copy()
    pinned = ksa.PinnedBuffer((steps, full), np.complex64)
    pinned.array[:] = x
    raw = np.stack([orc.quantize_u8(x[s] * 0.8) for s in range(min(steps, 4))])
    raw = np.tile(raw, (steps // 4 + 1, 1))[:steps].copy()
    for label, arg in (("c64 pageable", x), ("c64 pinned", pinned.array), ("u8 pageable", raw)):
        for _ in range(3):
            eng.scan_pass(arg)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.scan_pass(arg)
        dt = (time.perf_counter() - t0) / reps
        print("%-13s %-13s %8.1f us/pass  (%d bands of %d samples: %.1f MS/s, %.2f GB/s over PCIe)" %
              (name, label, dt * 1e6, steps, full, steps * full / dt / 1e6, arg.nbytes / dt / 1e9))
    pinned.close()
    eng.close()

# per-frame plot hand-off: the full state + Levels + markers (rounds 1-3) vs ksa_read_view (round 4)
for n, xres in ((4096, 512), (16384, 512), (65536, 512)):
    full = orc.full_size(n, 2.4e6 if n < 65536 else 1e9)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", xres=xres)
    eng.frame(orc.synth_iq(full, 3).astype(np.complex64))
    def old():
        st = eng.state()
        lv = eng.levels(xres, "AVG")
        idx, lvl = eng.highs(xres, "AVG", "cur", min_sep=0.025 * xres, count=5)
        return sum(st[k].size for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg", "fftHM")) * 4 + lv.size * 4 + 2 * len(idx) * 4
    def new():
        lv, idx, lvl, rows, hm_index = eng.view(xres, "AVG", curve="cur", min_sep=0.025 * xres, count=5, hm_rows=1)
        return (lv.size + rows.size + 2 * len(idx)) * 4
    for name, fn in (("state + levels + highs", old), ("ksa_read_view", new)):
        for _ in range(5):
            nbytes = fn()
        t0 = time.perf_counter(); reps = 200
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        print("hand-off N=%-6d xRes=%d %-24s %8.1f us/frame  %8d bytes over PCIe" % (n, xres, name, dt * 1e6, nbytes))
    eng.close()
