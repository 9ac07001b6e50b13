#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (configs 3-5 shapes and the uint8 variant of config 2) on one GPU:
spectrum-stage throughput and achieved algorithmic GB/s (SURVEY 8d byte model).  Not the headline bench --
bench.py is -- this fills the per-config table in DESIGN.md."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ksa_oracle as orc
ksa = importlib.import_module("prgs-sdr-kspecanal_amd")

CASES = [
    # name, N, nonOverlap, window, fullSize, frames, fmt, mode
    ("C2 zeroSpan c64", 4096, 0.5, "hanning", 32768, 16384, "c64", "zerospan"),
    ("C2 zeroSpan u8", 4096, 0.5, "hanning", 32768, 16384, "u8", "zerospan"),
    ("C2 geometry, default 0.1 hop", 4096, 0.1, "hanning", 32768, 4096, "c64", "zerospan"),
    ("C3 fmScan step", 16384, 0.1, "kaiser", 131072, 2304, "c64", "scan"),     # 128 passes x 18 steps
    ("C4 quickFullScan step", 64, 0.1, "ones", 512, 1226 * 64, "c64", "scan"),  # 64 passes x 1226 steps
    ("C5 zeroSpan 1 GS/s", 65536, 0.25, "hanning", 524288, 256, "c64", "zerospan"),
    ("N=1024 zeroSpan", 1024, 0.5, "hanning", 8192, 65536, "c64", "zerospan"),
]


def main():
    rows = []
    for name, n, q, win, full, frames, fmt, mode in CASES:
        distinct = min(frames, max(1, (64 << 20) // (full * 8)))
        host = orc.synth_iq(full * distinct, 20201226 + n).astype(np.complex64)
        if fmt == "c64":
            tile = torch.view_as_real(torch.from_numpy(host)).reshape(distinct, full, 2).cuda()
            code, sb = ksa.FMT_C64, 8
        else:
            tile = torch.from_numpy(orc.quantize_u8(host * 0.8)).reshape(distinct, full * 2).cuda()
            code, sb = ksa.FMT_U8, 2
        reps = (frames + distinct - 1) // distinct
        iq = tile.repeat(reps, *([1] * (tile.dim() - 1)))[:frames].contiguous()
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=win, max_frames=frames,
                                 stream=torch.cuda.current_stream().cuda_stream)
        out = torch.empty((frames, n), dtype=torch.float32, device="cuda")
        rowsbuf = torch.empty((frames, eng.hm_width), dtype=torch.float32, device="cuda")

        def step():
            if mode == "zerospan":
                eng.frames_dev(iq, code, frames, cur_db=out, hm_rows=rowsbuf)
            else:
                eng.curscan_dev(iq, code, frames, out, out_mode=ksa.OUT_DB_CLIP)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        eng.prof_enable(True)
        t0 = time.perf_counter()
        steps = 5
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ms, launches = eng.prof_read()
        kms = ms / max(1, launches)
        nwin = eng.num_windows
        bpf = full * sb + 4 * n + (4 * eng.hm_width if mode == "zerospan" else 0)
        info = eng.kernel_info()
        rows.append({"case": name, "N": n, "windows": nwin, "frames": frames, "fmt": fmt,
                     "spectrum_ms": kms, "step_ms": dt * 1e3, "MFFT_s": frames * nwin / kms / 1e3,
                     "GS_s": frames * full / kms / 1e6, "alg_GB_s": frames * bpf / kms / 1e6,
                     "frac_hbm": frames * bpf / kms / 1e6 / 8000.0, "vgprs": info["vgprs"], "path": info["path"]})
        print(json.dumps(rows[-1]))
        eng.close()
        del iq, out, rowsbuf, tile
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
