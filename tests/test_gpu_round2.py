"""GPU parity, round 2: the reference-run vectors of tests/golden/make_golden_r2.py through the C ABI --
BASELINE configs[2] (fmScan, fftSize 16384 kaiser) and configs[3]'s shape (quickFullScan) as whole scans, the dummy
band, Save/AdjSigLvls, the device-side peak markers (plot_highs), per-curve seeding under the bData* toggles, the
multi-pass scan batch, the PSD diagnostic, and bench.py's self-launching multi-rank path.
Tolerances: as test_gpu_parity.py (normalised linear error <= 1e-5; dB within 5e-3 on strong bins)."""
import hashlib
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import ksa_oracle as orc
from conftest import golden, load_pkg, ROOT
from test_gpu_parity import assert_db, assert_lin, GAIN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def _regen_iq(g, count):
    x = orc.synth_iq(count, int(g["seed"])).astype(np.complex64)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["iq_sha256"])
    return x


def _scan_engine(ksa, g, max_frames, **kw):
    n = int(g["fft_size"])
    groups = int((float(g["end_freq"]) - float(g["start_freq"])) / float(g["sampling_rate"]))
    return ksa.SpectrumEngine(n, full_size=int(g["full"]), non_overlap=float(g["non_overlap"]), window=str(g["window"]),
                              gain=float(g["gain"]), min_amp=float(g["min_amp"]), xres=int(g["xres"]),
                              max_frames=max_frames, scan_total_entries=groups * n,
                              scan_non_overlap=float(g["scan_non_overlap"]), **kw)


def _check_sampled(st, g, what):
    """Curves of a full-size scan against the fixture's sampled bins + decimated checksums.  The dB curves are
    compared through assert_db on the sampled bins (all extremes are among them) and on the per-cell maxima."""
    cells = int(g["cells"])
    for k in ("cur", "max", "min", "avg"):
        y = st["Fft." + k.capitalize()]
        assert_db(y[g[k + "_idx"]], g[k + "_at_idx"], what="%s %s sampled" % (what, k))
        assert_db(y.reshape(cells, -1).max(axis=1), g[k + "_decim_max"], what="%s %s cell max" % (what, k))
        # sum of dB values per cell: absolute tolerance scaled by the cell size (fp32 dB noise on weak bins)
        got, want = y.reshape(cells, -1).sum(axis=1), g[k + "_decim_sum"]
        per_bin = np.max(np.abs(got - want)) / (len(y) // cells)
        assert per_bin < 0.05, "%s %s mean dB deviation per cell %.3g" % (what, k, per_bin)


# ------------------------------------------------------------------- the two BASELINE scan configurations, whole
@pytest.mark.parametrize("tag,batched", [("fm_n16384", False), ("fm_n16384", True), ("quickfull_n64", False), ("quickfull_n64", True)])
def test_full_size_scan_vs_reference_golden(ksa, torch_cuda, tag, batched):
    """configs[2]: 18 steps x 71 windows of 16384 (kaiser) -> T=1024 spectrum kernel with OUT_DB_CLIP, window-split
    combine at 18 frames, stitch at total=147456 / hop=8192; configs[3] shape: 1226 steps of 71 x 64.  Pass by pass
    (ksa_scan_pass_dev) and as one two-pass batch (ksa_scan_passes_dev)."""
    torch = torch_cuda
    g = golden("scan_" + tag)
    n, full, passes, steps = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"])
    x = _regen_iq(g, full * steps * passes).reshape(passes, steps, full)
    eng = _scan_engine(ksa, g, steps * passes)
    assert eng.num_windows == 71
    dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    if batched:
        eng.scan_passes_dev(dev, ksa.FMT_C64, steps, passes)
    else:
        for p in range(passes):
            eng.scan_pass_dev(dev[p], ksa.FMT_C64, steps)
    st = eng.scan_state()
    assert st["passes"] == passes and st["hm_index"] == int(g["hm_index"])
    _check_sampled(st, g, tag)
    assert_db(st["fftHM"][:passes], g["hm_rows"][:passes], what=tag + " waterfall rows")
    assert np.allclose(st["fftHM"][passes], g["hm_rows"][passes], rtol=1e-6)    # untouched row = minAmp4Clip
    eng.close()


def test_fmscan_cli_driver_on_reference_stream(ksa):
    """kspecanal.main(["fmScan", ...]) fed with the very stream the reference consumed: the hand-off dict against
    the reference's own curves (configs[2] end to end through the drop-in front end)."""
    K = importlib.import_module("prgs-sdr-kspecanal_amd.kspecanal")
    g = golden("scan_fm_n16384")
    n, full, passes, steps = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"])
    stream = _regen_iq(g, full * steps * passes)

    class Replay:
        valid_gains_db, bandwidth, freq_correction = [19.1], 0, 0
        pos = 0

        def __init__(self):
            self.sample_rate = self.center_freq = 0
            self._gain, self._settle = 0, False

        gain = property(lambda self: self._gain, lambda self, v: (setattr(self, "_gain", v), setattr(self, "_settle", True)))

        def read_samples(self, cnt):
            cnt = int(cnt)
            if self._settle:                      # the 16Ki settle read after a retune (K:301) is discarded
                self._settle = False
                return np.zeros(cnt, dtype=np.complex64)
            out = stream[Replay.pos:Replay.pos + cnt]
            Replay.pos += cnt
            return out

        def close(self):
            pass
    orig = K.open_source
    K.open_source = lambda d: Replay()
    try:
        d = K.main(["fmScan", "window", "kaiser", "prgLoopCnt", str(passes), "bPltLevels", "false", "bPltHeatMap", "false"])
    finally:
        K.open_source = orig
    assert Replay.pos == len(stream)
    assert d["endFreq"] == float(g["end_freq"]) and len(d["Fft.Cur"]) == int(g["total"])
    st = {k: d[k] for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg")}
    _check_sampled(st, g, "fmScan main")
    assert_db(d["fftHM"][:passes], g["hm_rows"][:passes], what="fmScan main waterfall")


def test_scan_dummy_band_vs_reference_golden(ksa, torch_cuda):
    torch = torch_cuda
    g = golden("scan_dummy_n512")
    n, full, passes, steps = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"])
    ok = g["step_ok"]
    x = _regen_iq(g, full * int(ok.sum())).reshape(-1, full)
    blocks = np.zeros((passes, steps, full), dtype=np.complex64)      # failed tunes deliver nothing
    blocks[ok.astype(bool)] = x
    dev = torch.view_as_real(torch.from_numpy(blocks)).cuda()
    for batched in (False, True):
        eng = _scan_engine(ksa, g, steps * passes)
        if batched:
            eng.scan_passes_dev(dev, ksa.FMT_C64, steps, passes, step_ok=ok.reshape(-1))
        else:
            for p in range(passes):
                eng.scan_pass_dev(dev[p], ksa.FMT_C64, steps, step_ok=ok[p])
        st = eng.scan_state()
        for k in ("cur", "max", "min", "avg"):
            assert_db(st["Fft." + k.capitalize()], g[k], what="dummy band %s (batched=%s)" % (k, batched))
        assert_db(st["fftHM"][:passes], g["hm"][:passes], what="dummy band waterfall")
        assert st["hm_index"] == int(g["hm_index"])
        eng.close()


# ----------------------------------------------------------------------------------- Save / AdjSigLvls + markers
def _marker_check(freqs_x, idx, lvl, want, what):
    got_f = freqs_x[idx]
    assert len(idx) == len(want), (what, idx, want)
    assert np.array_equal(got_f, want[:, 0]), (what, got_f, want[:, 0])
    assert np.max(np.abs(lvl - want[:, 1])) < 5e-3, (what, lvl, want[:, 1])


def test_adj_siglvls_zerospan_vs_reference_golden(ksa):
    g = golden("adj_zerospan_n512")
    n, q, full, frames, xres = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"]), int(g["frames"]), int(g["xres"])
    x = _regen_iq(g, full * frames).reshape(frames, full)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=str(g["window"]), gain=float(g["gain"]), xres=xres)
    eng.set_adj(g["adj"])
    for fr in x:
        eng.frame(fr)
    st = eng.state()
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], g[k], what="adj zeroSpan " + k)      # curves stay raw (K:400-411)
    assert_db(st["fftHM"][:frames], g["hm"][:frames], what="adj zeroSpan waterfall")   # rows see cur - Adj (K:480)
    mode = str(g["compress"])
    lv = eng.levels(xres, mode)                       # rows cur, max, min, avg, baseline-adjusted on the device
    for row, name in ((0, "lv_cur"), (1, "lv_max"), (2, "lv_min"), (3, "lv_avg")):
        assert np.max(np.abs(lv[row] - g[name])) < 5e-3, name
    span = g["lv_x"][-1] - g["lv_x"][0]
    sep = float(g["marker_delta"]) * span / (span / (xres - 1))
    idx, lvl = eng.highs(xres, mode, "cur", min_sep=sep, count=int(g["marker_count"]))
    _marker_check(g["lv_x"], idx, lvl, g["markers"], "adj zeroSpan markers")
    eng.close()


def test_adj_siglvls_scan_vs_reference_golden(ksa, torch_cuda):
    torch = torch_cuda
    g = golden("adj_scan_n256")
    n, full, passes, steps, xres = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"]), int(g["xres"])
    x = _regen_iq(g, full * steps * passes).reshape(passes, steps, full)
    eng = _scan_engine(ksa, g, steps)
    eng.set_adj(g["adj"])
    dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    for p in range(passes):
        eng.scan_pass_dev(dev[p], ksa.FMT_C64, steps)
    st = eng.scan_state()
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], g[k], what="adj scan " + k)
    # the dB waterfall rows here are differences of two dB curves (can be ~0): compare them as differences
    assert np.max(np.abs(st["fftHM"][:passes] - g["hm"][:passes])) < 5e-3
    mode = str(g["compress"])
    lv = eng.levels(xres, mode, scan=True)
    for row, name in ((0, "lv_cur"), (1, "lv_max"), (2, "lv_min"), (3, "lv_avg")):
        assert np.max(np.abs(lv[row] - g[name])) < 5e-3, name
    span = g["lv_x"][-1] - g["lv_x"][0]
    sep = float(g["marker_delta"]) * span / (span / (xres - 1))
    idx, lvl = eng.highs(xres, mode, "cur", min_sep=sep, count=int(g["marker_count"]), scan=True)
    _marker_check(g["lv_x"], idx, lvl, g["markers"], "adj scan markers")
    eng.close()


def test_device_plot_highs_vs_reference_golden(ksa, torch_cuda):
    """ksa_read_highs on the reference's own plot_highs cases: the curve is written into the engine's Cur state
    (ksa_state_dev) and marked on the device with cells == points."""
    torch = torch_cuda
    g = golden("plot_highs")
    for c, (delta, count) in enumerate(g["cases"]):
        freqs, lv, want = g["freqs%d" % c], g["levels%d" % c], g["marks%d" % c]
        npts = len(freqs)
        eng = ksa.SpectrumEngine(npts, full_size=npts * 8, non_overlap=0.5, window="ones", xres=npts)
        state, _ = eng.state_dev()
        torch.as_tensor(state, device="cuda")[0].copy_(torch.from_numpy(lv.astype(np.float32)))
        span = freqs[-1] - freqs[0]
        sep = float(delta) * span / (span / (npts - 1))
        idx, lvl = eng.highs(npts, "MAX", "cur", min_sep=sep, count=min(int(count), 64))
        assert len(idx) == len(want), (c, len(idx), len(want))
        assert np.array_equal(freqs[idx], want[:, 0]), c
        assert np.array_equal(lvl, lv.astype(np.float32)[idx].astype(np.float64)), c
        eng.close()


def test_device_highs_nan_and_refusals(ksa, torch_cuda):
    torch = torch_cuda
    n = 64
    eng = ksa.SpectrumEngine(n, full_size=512, non_overlap=0.5, window="ones", xres=n)
    lv = np.linspace(-80, -20, n).astype(np.float32)
    lv[10] = np.nan                                  # numpy's argsort puts NaN last: the walk meets it first
    lv[40] = np.inf
    state, _ = eng.state_dev()
    torch.as_tensor(state, device="cuda")[0].copy_(torch.from_numpy(lv))
    idx, lvl = eng.highs(n, "MAX", "cur", min_sep=3.0, count=4)
    want = orc.plot_highs(np.arange(n, dtype=np.float64), lv.astype(np.float64), 3.0 / (n - 1), 4)
    assert idx.tolist() == [int(f) for f, _ in want] and idx[0] == 10 and idx[1] == 40
    with pytest.raises(ksa.KsaError):
        eng.highs(n, "MAX", "cur", count=65)
    with pytest.raises(ksa.KsaError):
        eng.highs(48, "MAX", "cur")
    eng.close()


# ------------------------------------------------------------------------------------ bData* toggles (K:471-476)
def test_flags_toggle_seeds_each_curve_by_copy(ksa):
    """A curve that was off for the first frames is still None in the reference and is seeded by COPY when its flag
    comes on (data_cumu(None), K:133-134) -- it must not be merged with the zero-initialised device state."""
    n, full = 512, 4096
    x = orc.synth_iq(full * 7, 515).astype(np.complex64).reshape(7, full)
    x[:, :] *= 400.0                                  # tone bins above 0 dB (a zero-seeded Min sticks at 0), noise below
                                                      # (a zero-seeded Max sticks at 0)
    win = orc.window_table("hanning", n)
    ref = orc.ZeroSpanState(n, 128, GAIN, b_max=False, b_min=False, b_avg=False)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=128)
    plan = [(False, False, False)] * 2 + [(True, False, True)] * 2 + [(False, True, False)] + [(True, True, True)] * 2
    for fr, (bmax, bmin, bavg) in zip(x, plan):
        ref.b_max, ref.b_min, ref.b_avg = bmax, bmin, bavg
        ref.push(orc.curscan(fr, n, 0.5, win, "AVG"))
        eng.set_flags(bmax, bmin, bavg)
        eng.frame(fr)
    st = eng.state()
    assert (ref.min > 0).any() and (ref.max < 0).any()
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], getattr(ref, k), what="toggled " + k)
    eng.close()


def test_flags_toggle_negative_levels_batched(ksa, torch_cuda):
    """Same through ksa_frames_dev batches with ordinary negative dB data (a zero-seeded Max would stick at 0 dB)."""
    torch = torch_cuda
    n, full = 1024, 8192
    x = orc.synth_iq(full * 9, 616).astype(np.complex64).reshape(9, full) * 0.01
    win = orc.window_table("hamming", n)
    ref = orc.ZeroSpanState(n, 256, GAIN, b_max=False, b_avg=False)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hamming", gain=GAIN, xres=256, max_frames=4)
    dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    for lo, hi, flags in ((0, 3, (False, True, False)), (3, 7, (True, True, True)), (7, 9, (True, False, True))):
        ref.b_max, ref.b_min, ref.b_avg = flags
        for fr in x[lo:hi]:
            ref.push(orc.curscan(fr, n, 0.5, win, "AVG"))
        eng.set_flags(*flags)
        eng.frames_dev(dev[lo:hi], ksa.FMT_C64, hi - lo)
    st = eng.state()
    assert np.max(ref.max) < 0
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], getattr(ref, k), what="toggled batch " + k)
    eng.close()


# ------------------------------------------------------------------------------------------ multi-pass scan batch
@pytest.mark.parametrize("n,passes,base_raw", [(256, 5, False), (64, 140, False), (512, 3, True)])
def test_scan_pass_batch_equals_pass_by_pass(ksa, torch_cuda, n, passes, base_raw):
    """ksa_scan_passes_dev == `passes` ksa_scan_pass_dev calls, bit for bit (state, ring, ring position); 140 passes
    wrap the 128-row ring."""
    torch = torch_cuda
    fs, start, end = 2.4e6, 150e6, 159.6e6
    steps = len(orc.scan_steps(start, end, fs, 0.5))
    full = orc.full_size(n, fs)
    x = orc.synth_iq(full * steps * passes, 800 + n).astype(np.complex64).reshape(passes, steps, full)
    dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    total = int((end - start) / fs) * n
    res = []
    for batched in (False, True):
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.1, window="hanning", gain=GAIN, xres=64,
                                 max_frames=steps * passes, scan_total_entries=total)
        eng.scan_set_base_is_raw(base_raw)
        # a first single pass, then the batch: the batch must continue an existing state too
        eng.scan_pass_dev(dev[0], ksa.FMT_C64, steps)
        if batched:
            eng.scan_passes_dev(dev[1:], ksa.FMT_C64, steps, passes - 1)
        else:
            for p in range(1, passes):
                eng.scan_pass_dev(dev[p], ksa.FMT_C64, steps)
        res.append(eng.scan_state())
        eng.close()
    a, b = res
    assert a["passes"] == b["passes"] == passes and a["hm_index"] == b["hm_index"] == passes % 128
    for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg", "fftHM"):
        assert np.array_equal(a[k], b[k]), k
    ref = orc.ScanState(n, start, end, fs, GAIN, (1 / 256) * 0.00001, 64, base_is_raw=base_raw)
    win = orc.window_table("hanning", n)
    for p in range(passes):
        ref.run_pass([orc.curscan(x[p, s], n, 0.1, win, "AVG") for s in range(steps)])
    for k in ("cur", "max", "min", "avg"):
        assert_db(b["Fft." + k.capitalize()], getattr(ref, k), what="batched scan " + k)
    assert_db(b["fftHM"], ref.hm, what="batched scan ring")


# ------------------------------------------------------------------------------------------------ PSD diagnostic
def test_psd_crosscheck_diagnostic(ksa):
    """bUsePSD: matplotlib's Welch PSD of the same block next to the GPU spectrum (CPU-only diagnostic; the
    reference's own branch raises TypeError under matplotlib >= 3.8, so this is parity-unpinned): the strongest bin
    agrees and its level, converted to the amplitude convention, is within 0.2 dB."""
    pytest.importorskip("matplotlib")
    K = importlib.import_module("prgs-sdr-kspecanal_amd.kspecanal")
    d = K.main(["zeroSpan", "fftSize", "1024", "window", "hanning", "curScanNonOverlap", "0.5", "prgLoopCnt", "2",
                "centerFreq", "100.3e6", "bUsePSD", "true", "bPltLevels", "false", "bPltHeatMap", "false", "source", "synth"])
    kg, kp, lg, lp = d["psd.check"]
    assert kg == kp and abs(lg - lp) < 0.2
    assert len(d["psd.cur"]) == 1024 and d["Fft.Cur"] is not None


# --------------------------------------------------------------------------------------------- bench.py launching
def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("cfg,extra,units", [(2, ["--frames", "1024"], 1024 * 15), (5, ["--frames", "8"], 8 * 29),
                                             (4, ["--passes", "4"], 4 * 1226 * 71 // 2), (3, ["--passes", "2"], 2 * 18 * 71 // 2)])
def test_bench_self_launch_two_ranks_on_one_gpu(cfg, extra, units):
    """`python bench.py --gpus 2 --config C` outside torch.distributed.run: the parent starts the two ranks itself.
    Rehearsed on ONE GPU with the gloo backend (KSA_BENCH_BACKEND); the collective string names both ranks."""
    d = _bench(["--gpus", "2", "--config", str(cfg), "--steps", "2", "--warmup", "1", "--no-cpu"] + extra,
               env={"KSA_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["baseline_config"] == cfg and "2 ranks" in d["config"]["collective"]
    assert d["scaling"] == ("weak" if cfg in (2, 5) else "strong") and d["value"] > 0 and d["roofline"]["launches"] == 2
    per_step = units * 2                                 # whole job: both ranks
    assert abs(d["value"] - per_step * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-9


@pytest.mark.parametrize("cfg,extra", [(3, ["--passes", "8"]), (4, ["--passes", "16"]), (5, ["--frames", "16"])])
def test_bench_secondary_configs_one_gpu(cfg, extra):
    d = _bench(["--config", str(cfg), "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.3"] + extra)
    assert d["config"]["baseline_config"] == cfg and d["n_gpus"] == 1 and d["unit"] == "FFT/s"
    rf = d["roofline"]
    assert rf["bound"] == {3: "valu", 4: "valu", 5: "hbm (Z round trip)"}[cfg] and rf["limiter"] and 0 < rf["frac"] < 1 and 0 < rf["flop_frac"] < 1 and rf["frac_step"] <= rf["frac"] * 1.001
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


# (the two-frames-per-workgroup kernel against the one-frame kernel: tests/test_gpu_round4.py, without environment switches)
