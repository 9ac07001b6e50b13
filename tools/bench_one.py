#!/usr/bin/env python3
"""Spectrum-stage timing of one shape: bench_one.py N nonOverlap window fullSize frames [fmt]"""
import importlib, os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ksa_oracle as orc
ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
n, q, win, full, frames = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
distinct = min(frames, max(1, (64 << 20) // (full * 8)))
host = orc.synth_iq(full * distinct, 1 + n).astype(np.complex64)
tile = torch.view_as_real(torch.from_numpy(host)).reshape(distinct, full, 2).cuda()
iq = tile.repeat((frames + distinct - 1) // distinct, 1, 1)[:frames].contiguous()
eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=win, max_frames=frames, stream=torch.cuda.current_stream().cuda_stream)
out = torch.empty((frames, n), dtype=torch.float32, device="cuda")
for _ in range(2): eng.curscan_dev(iq, ksa.FMT_C64, frames, out, out_mode=ksa.OUT_DB)
torch.cuda.synchronize(); eng.prof_enable(True)
for _ in range(5): eng.curscan_dev(iq, ksa.FMT_C64, frames, out, out_mode=ksa.OUT_DB)
ms, k = eng.prof_read()
print("%s N=%d q=%s: %.3f ms  %.2f MFFT/s  vgpr %d" % (os.environ.get("KSA_VARIANT", "main"), n, q, ms / k, frames * eng.num_windows / (ms / k) / 1e3, eng.kernel_info()["vgprs"]))
