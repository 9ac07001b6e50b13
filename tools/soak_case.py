#!/usr/bin/env python3
"""GPU box: one case of tests/test_gpu_random.py::test_random_scan in detail (where the largest deviation from the oracle sits).
    KSA_RANDOM_CASES=500 KSA_RANDOM_SEED=123 python3 tools/soak_case.py s87-N4096"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ksa_oracle as orc
import test_gpu_random as T
ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
want_id = sys.argv[1]
cases = T._scan_cases(T.COUNT // 4 if T.SOAK else 24, T.SEED + 1)
case = [c for c in cases if ("s%d-N%d" % (c[0], c[1])) == want_id][0]
i, n, sq, start, end, fs, passes, window, base_raw, q, xres, dummies = case
print("case", case)
end, _ = orc.fixup_scan_range(start, end, fs)
steps = len(orc.scan_steps(start, end, fs, sq))
full = 2 * n
total = int((end - start) / fs) * n
if total % xres:
    xres = n
ref = orc.ScanState(n, start, end, fs, 19.1, 1e-7, xres, scan_non_overlap=sq, base_is_raw=base_raw)
win = orc.window_table(window, n)
rng = np.random.default_rng(900 + i)
eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=window, gain=19.1, min_amp=1e-7, xres=xres,
                         max_frames=steps, scan_total_entries=ref.total, scan_non_overlap=sq)
eng.scan_set_base_is_raw(base_raw)
for p in range(passes):
    x = (orc.synth_iq(full * steps, 7000 + 10 * i + p) * (0.3 + 0.5 * rng.random())).astype(np.complex64).reshape(steps, full)
    ok = np.ones(steps, dtype=np.uint8)
    if dummies:
        ok[rng.integers(0, steps, size=max(1, steps // 5))] = 0
    lin = [orc.curscan(x[s], n, q, win, "AVG") if ok[s] else None for s in range(steps)]
    ref.run_pass(lin)
    eng.scan_pass_dev(torch.view_as_real(torch.from_numpy(x)).cuda(), ksa.FMT_C64, steps, step_ok=ok)
    # the device's per-band spectra of this pass against the oracle's
    out = torch.empty((steps, n), dtype=torch.float32, device="cuda")
    eng.curscan_dev(torch.view_as_real(torch.from_numpy(x)).cuda(), ksa.FMT_C64, steps, out, out_mode=ksa.OUT_LINEAR)
    got = out.cpu().numpy().astype(np.float64)
    for s in range(steps):
        if ok[s]:
            e = np.abs(got[s] - lin[s]); j = int(np.argmax(e))
            print("pass %d band %d: max |d| %.3g at bin %d (value %.4g, band max %.4g) -> %.3g of the band max" % (p, s, e[j], j, lin[s][j], lin[s].max(), e[j] / lin[s].max()))
st = eng.scan_state()
for k in ("cur", "max", "min", "avg"):
    g = 10 ** (st["Fft." + k.capitalize()] / 10); w = 10 ** (getattr(ref, k) / 10)
    e = np.abs(g - w); j = int(np.argmax(e))
    print("%s: max |d| %.3g at element %d (want %.6g got %.6g, curve max %.4g) normalised %.3g | dB there want %.5f got %.5f" % (
        k, e[j], j, w[j], g[j], w.max(), e[j] / w.max(), getattr(ref, k)[j], st["Fft." + k.capitalize()][j]))
print("window", window, "q", q, "steps", steps, "windows per band", len(eng.starts))
