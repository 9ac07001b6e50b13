#!/usr/bin/env python3
"""GPU box: what ksa_prof_clock reports for spectrum stages of different lengths (config-2 geometry): median / min / max of
d(s_memtime) / d(s_memrealtime) x 100 MHz over the XCDs and launches, next to the HIP-event time of the stage."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import ksa_oracle as orc
ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
n, full, q = 4096, 32768, 0.5
x = orc.synth_iq(full * 64, 3).astype(np.complex64).reshape(64, full)
tile = torch.view_as_real(torch.from_numpy(x)).to("cuda")
for frames in (256, 1024, 4096, 8192, 16384, 65536):
    iq = tile.repeat((frames + 63) // 64, 1, 1)[:frames].contiguous()
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning", max_frames=frames)
    for _ in range(3):
        eng.frames_dev(iq, ksa.FMT_C64, frames)
    eng.prof_enable(True)
    for _ in range(8):
        eng.frames_dev(iq, ksa.FMT_C64, frames)
    ms, launches = eng.prof_read()
    ghz, samples = eng.prof_clock()
    print("frames %6d  stage %.4f ms  clock median %s GHz  range %s  samples %d" % (frames, ms / launches, ghz, eng.prof_clock_range, samples))
    eng.close()
