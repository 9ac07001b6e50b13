#!/usr/bin/env python3
"""Create / use / destroy many engines of mixed shapes in one process: device memory must come back."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ksa_oracle as orc
ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
rng = np.random.default_rng(1)
free0 = torch.cuda.mem_get_info()[0]
marks = {}
for i in range(300):
    n = int(rng.choice([64, 512, 4096, 16384, 65536]))
    full = 2 * n
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=float(rng.choice([0.1, 0.5])), window="hanning",
                             max_frames=int(rng.integers(1, 64)), xres=64, scan_total_entries=4 * n if i % 3 == 0 else 0)
    x = (rng.standard_normal(full) + 1j * rng.standard_normal(full)).astype(np.complex64)
    y = eng.curscan(x)
    assert np.isfinite(y).all()
    eng.frame(x)
    if i % 2 == 0:
        eng.close()          # the other half is left to __del__
    del eng
    if i in (49, 99, 199, 299):
        torch.cuda.synchronize(); marks[i] = (free0 - torch.cuda.mem_get_info()[0]) / 2**20
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MiB, after %.1f MiB, delta %.1f MiB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
print("held after N engines (MiB):", marks)
assert marks[299] - marks[99] < 64, "device memory grows with the number of engines created"
print("ok")
