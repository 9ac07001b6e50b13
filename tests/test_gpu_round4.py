"""GPU parity, round 4: NaN / inf semantics of the waterfall cells and of the within-block fold (np.max / np.min /
np.clip as the reference applies them, K:141-143, K:195, K:100-101, K:110-111), the product library without its
experiment switches, the eight-way quickFullScan golden, engines placed on distinct devices where a node offers them.
All through the C ABI; tolerances as test_gpu_parity.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import ksa_oracle as orc
from conftest import golden, load_pkg, ROOT
from test_gpu_parity import assert_db, assert_lin, GAIN
from test_gpu_round2 import _regen_iq, _scan_engine, _check_sampled, _bench

pytestmark = pytest.mark.gpu
CURVES = ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def _same_specials(got, want, what):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want)), what + ": NaN pattern"
    assert np.array_equal(np.isposinf(got), np.isposinf(want)), what + ": +inf pattern"
    assert np.array_equal(np.isneginf(got), np.isneginf(want)), what + ": -inf pattern"
    fin = np.isfinite(want)
    if fin.any():
        assert np.max(np.abs(got[fin] - want[fin])) < 5e-3, what + ": finite cells"


# ------------------------------------------------------------------------------- waterfall cells: np.max (K:195, K:480)
@pytest.mark.parametrize("n,xres,full", [(4096, 512, 32768), (512, 512, 4096), (1024, 512, 8192), (64, 16, 512),
                                         (4096, 8, 32768), (16384, 16, 32768), (8192, 512, 16384), (65536, 512, 131072),
                                         (65536, 32, 131072)])
def test_waterfall_nan_cells_same_in_every_batch_shape(ksa, torch_cuda, n, xres, full):
    """VERDICT r03 item 2.  A baseline saved from a run that saw a zero magnitude holds -inf (K:469 keeps it); a later
    all-zero frame then gives -inf - (-inf) = NaN at those bins (K:405) and np.max (K:195) makes the whole waterfall cell
    NaN; a normal frame gives +inf there.  Every cell-reduction path of the device -- the shuffle tree of finish_frame
    (g = 4 .. 256), its g = 1 / g = 2 forms, its LDS path (g > 256), rowmax_batch behind the window-split mode,
    dif16_finish_kernel -- must give the same row for the same frame whether it arrives in a batch of 3 (window-split)
    or among many (persistent kernel), equal to the oracle's."""
    torch = torch_cuda
    q, frames_big = 0.5, 600 if n <= 16384 else 40
    rng = np.random.default_rng(n + xres)
    adj = rng.normal(-40.0, 3.0, n)
    hole = rng.choice(n, size=max(3, n // 97), replace=False)
    adj[hole] = -np.inf
    distinct = 5
    x = orc.synth_iq(full * distinct, 77 + n).astype(np.complex64).reshape(distinct, full)
    x[2] = 0                                                    # the all-zero frame
    win = orc.window_table("hanning", n)
    # oracle rows of the distinct frames (a frame's row does not depend on its neighbours)
    want = np.stack([orc.plotcompress(orc.log_no_gain(orc.curscan(x[f], n, q, win, "AVG"), GAIN) - adj, orc.heatmap_width(n, xres), "MAX")
                     for f in range(distinct)])
    assert np.isnan(want[2]).any() and np.isposinf(want[0]).any()
    for frames in (3, frames_big):
        idx = np.arange(frames) % distinct
        dev = torch.view_as_real(torch.from_numpy(x[idx])).cuda()
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning", gain=GAIN, xres=xres, max_frames=frames)
        eng.set_adj(adj, scan=False)
        rows = torch.empty((frames, eng.hm_width), dtype=torch.float32, device="cuda")
        eng.frames_dev(dev, ksa.FMT_C64, frames, hm_rows=rows)
        got = rows.cpu().numpy()
        for f in range(frames):
            _same_specials(got[f], want[idx[f]], "N=%d W=%d batch of %d, frame %d" % (n, xres, frames, f))
        st = eng.state()
        last = min(frames, 128)
        for f in range(frames - last, frames):                  # the ring holds the same rows
            _same_specials(st["fftHM"][f % 128], want[idx[f]], "ring row of frame %d" % f)
        eng.close()


def test_waterfall_nan_against_zerospan_state(ksa, torch_cuda):
    """The same through the frame-by-frame host entry point (ksa_frame_c64) against orc.ZeroSpanState: curves + ring."""
    n, full, xres = 4096, 32768, 512
    rng = np.random.default_rng(5)
    adj = rng.normal(-40.0, 3.0, n)
    adj[rng.choice(n, 40, replace=False)] = -np.inf
    x = orc.synth_iq(full * 4, 99).astype(np.complex64).reshape(4, full)
    x[1] = 0
    win = orc.window_table("hanning", n)
    ref = orc.ZeroSpanState(n, xres, GAIN, adj=adj)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=xres, max_frames=1)
    eng.set_adj(adj, scan=False)
    for f in range(4):
        ref.push(orc.curscan(x[f], n, 0.5, win, "AVG"))
        eng.frame(x[f])
    st = eng.state()
    for f in range(4):
        _same_specials(st["fftHM"][f], ref.hm[f], "ring row %d" % f)
    for k, w in zip(CURVES, (ref.cur, ref.max, ref.min, ref.avg)):
        assert np.array_equal(np.isnan(st[k]), np.isnan(w)) and np.array_equal(np.isneginf(st[k]), np.isneginf(w)), k
    eng.close()


# ------------------------------------------------------------------------------- within-block fold: np.max / np.min (K:141-143)
@pytest.mark.parametrize("n,q,full,frames", [(64, 0.1, 512, 5), (256, 0.5, 2048, 400), (1024, 0.5, 8192, 3), (1024, 0.5, 8192, 2100),
                                             (4096, 0.25, 32768, 3), (4096, 0.5, 32768, 800), (8192, 0.5, 32768, 3),
                                             (16384, 0.1, 65536, 300), (65536, 0.5, 131072, 4)])
@pytest.mark.parametrize("mode", ["MAX", "MIN", "AVG"])
def test_fold_propagates_nan_windows(ksa, torch_cuda, n, q, full, frames, mode):
    """One sample of a frame is NaN -- in frame 1 inside the FIRST window only (the clean windows after it must not wash
    it out), in frame 3 inside the LAST window only: numpy.fft turns a window that covers it into NaN and np.max / np.min
    (K:141-143) keep the bin NaN through the rest of the block's fold.  v_max_f32 / v_min_f32 return the other operand;
    every fold path -- slots of small transforms, the window-split shares, the pair kernel (N = 1024, large batch), the
    32-point kernel, the second stage of the large transform -- must give the oracle's NaN pattern (other frames stay clean)."""
    torch = torch_cuda
    distinct = 4
    x = orc.synth_iq(full * distinct, 1000 + n).astype(np.complex64).reshape(distinct, full)
    x[1, 2] = np.nan
    x[3, full - 1] = np.nan
    idx = np.arange(frames) % distinct
    dev = torch.view_as_real(torch.from_numpy(x[idx])).cuda()
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hamming", cumu_mode=mode, gain=GAIN, xres=64, max_frames=frames)
    out = torch.empty((frames, n), dtype=torch.float32, device="cuda")
    eng.curscan_dev(dev, ksa.FMT_C64, frames, out, out_mode=ksa.OUT_LINEAR)
    got = out.cpu().numpy()
    win = orc.window_table("hamming", n)
    with np.errstate(invalid="ignore", over="ignore"):
        want = [orc.curscan(x[f], n, q, win, mode) for f in range(distinct)]
    assert np.isnan(want[1]).any() and np.isnan(want[3]).any() and not np.isnan(want[0]).any()
    for f in range(frames):
        w = want[idx[f]]
        assert np.array_equal(np.isnan(got[f]), np.isnan(w)), "N=%d %s frame %d of %d: NaN pattern" % (n, mode, f, frames)
        ok = ~np.isnan(w)
        if ok.any():
            assert_lin(got[f][ok], w[ok], what="N=%d %s frame %d" % (n, mode, f))
    eng.close()


# ------------------------------------------------------------------------------- scan: Clip2MinAmp + LogNoGain(infTo = 0)
def test_scan_with_zero_min_amp_maps_inf_to_zero(ksa, torch_cuda):
    """minAmp4Clip = 0 (a legal CLI value): np.clip leaves a zero magnitude alone, 10 log10 gives -inf and the scan's
    LogNoGain replaces +-inf by 0 (infTo = 0: K:641 -> K:110-111; the curves also START at 0 dB then, K:603-604)."""
    torch = torch_cuda
    n, full, fs = 256, 2048, 2.4e6
    start, end = 100e6, 100e6 + 3 * fs
    ref = orc.ScanState(n, start, end, fs, GAIN, 0.0, 64, 0.5)
    steps = len(ref.centers)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, min_amp=0.0, xres=64,
                             max_frames=steps, scan_total_entries=3 * n, scan_non_overlap=0.5)
    st = eng.scan_state()
    assert np.all(st["Fft.Cur"] == 0.0) and np.all(st["Fft.Max"] == 0.0) and np.all(st["Fft.Avg"] == 0.0)
    win = orc.window_table("hanning", n)
    x = orc.synth_iq(full * steps * 2, 31).astype(np.complex64).reshape(2, steps, full)
    x[0, 2] = 0                                                 # a silent band: zero magnitudes
    x[1, 0] = 0
    for p in range(2):
        ref.run_pass([orc.curscan(x[p, s], n, 0.5, win, "AVG") for s in range(steps)])
        eng.scan_pass(x[p])
    st = eng.scan_state()
    for k, w in zip(CURVES, (ref.cur, ref.max, ref.min, ref.avg)):
        assert np.all(np.isfinite(st[k])), k
        assert np.array_equal(st[k] == 0.0, w == 0.0), k + ": bins the reference zeroes"
        assert_db(st[k], w, what="min_amp 0 " + k)
    assert_db(st["fftHM"][:2], ref.hm[:2], what="min_amp 0 waterfall")
    eng.close()


# ------------------------------------------------------------------------------- two frames per workgroup (N = 1024)
@pytest.mark.parametrize("q,fmt,mode", [(0.5, "c64", "AVG"), (0.1, "u8", "MAX"), (0.25, "c64", "MIN")])
def test_pair_kernel_matches_single_frame_kernel(ksa, torch_cuda, q, fmt, mode):
    """spectrum_pair_kernel (N = 1024, batches of >= 2 x CUs x workgroups-per-CU frames) against spectrum_kernel (the
    same frames handed over in batches below the switch-over), with no environment switch: the product library reads
    none.  Frames, state and waterfall rows agree to fp32 rounding (the pair kernel builds its last-pass twiddles from 6
    instead of 15 table entries); a few frames are checked against the oracle as well."""
    torch = torch_cuda
    n, full = 1024, 8192
    probe = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="kaiser", cumu_mode=mode, xres=256, max_frames=1)
    switch_over = 2 * probe.kernel_info()["grid"]    # ksa_api.hip: batches of >= 2 x CUs x workgroups-per-CU frames take the pair kernel
    probe.close()
    frames = switch_over + 513                       # above the switch-over, odd
    distinct = 37
    x = orc.synth_iq(full * distinct, 4321 + n).astype(np.complex64).reshape(distinct, full)
    idx = np.arange(frames) % distinct
    if fmt == "u8":
        dev = torch.from_numpy(orc.quantize_u8((x * 0.7).reshape(-1)).reshape(distinct, 2 * full)[idx]).cuda()
        code = ksa.FMT_U8
    else:
        dev = torch.view_as_real(torch.from_numpy(x[idx])).cuda()
        code = ksa.FMT_C64
    outs = []
    assert 500 < switch_over
    for chunk in (frames, 500):
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="kaiser", cumu_mode=mode, xres=256, max_frames=frames)
        assert eng.kernel_info()["path"] == 4
        db = torch.empty((frames, n), dtype=torch.float32, device="cuda")
        rows = torch.empty((frames, eng.hm_width), dtype=torch.float32, device="cuda")
        for f0 in range(0, frames, chunk):
            cf = min(chunk, frames - f0)
            eng.frames_dev(dev[f0:f0 + cf], code, cf, cur_db=db[f0:f0 + cf], hm_rows=rows[f0:f0 + cf])
        st = eng.state()
        outs.append((db.cpu().numpy(), rows.cpu().numpy(), st))
        eng.close()
    (da, ra, sa), (db_, rb, sb) = outs
    assert_lin(10 ** (da / 10), 10 ** (db_.astype(np.float64) / 10), tol=2e-6, what="pair vs single frames")
    assert not np.array_equal(da, db_), "both runs took the same kernel: the switch-over moved?"
    top = np.max(db_, axis=1, keepdims=True)
    strong = db_ > top - 30
    assert np.max(np.abs(da[strong] - db_[strong])) < 1e-3
    assert np.max(np.abs(ra - rb)) < 1e-3                       # waterfall rows = per-cell maxima
    for k in CURVES:
        assert_db(sa[k], sb[k], what="pair vs single " + k)
    assert np.max(np.abs(sa["fftHM"] - sb["fftHM"])) < 1e-3 and sa["hm_index"] == sb["hm_index"]
    win = orc.window_table("kaiser", n)
    for f in (0, 1, frames - 1):
        src = x[idx[f]] if fmt == "c64" else orc.unpack_u8(orc.quantize_u8(x[idx[f]] * 0.7))
        want = orc.log_no_gain(orc.curscan(src, n, q, win, mode), 19.1)
        assert_db(da[f], want, what="pair frame %d" % f)


def test_library_reads_no_environment_switch(ksa, torch_cuda):
    """a stray experiment variable must not change which kernel a user runs.  With every former
    switch set, an engine reports the same plan as without, and the product library has no getenv import at all."""
    torch = torch_cuda
    base = {}
    for n in (1024, 4096, 16384):
        eng = ksa.SpectrumEngine(n, full_size=8 * n, non_overlap=0.5, window="hanning", max_frames=4)
        base[n] = eng.kernel_info()
        eng.close()
    code = ("import importlib, json, sys; sys.path.insert(0, %r); ksa = importlib.import_module('prgs-sdr-kspecanal_amd'); out = {}\n"
            "for n in (1024, 4096, 16384):\n"
            "    e = ksa.SpectrumEngine(n, full_size=8 * n, non_overlap=0.5, window='hanning', max_frames=4); out[n] = e.kernel_info(); e.close()\n"
            "print(json.dumps(out))") % ROOT
    env = dict(os.environ, KSA_NO_PAIR="1", KSA_PAIR_ALL="1", KSA_PLAN16="1", KSA_NO_REUSE="1", KSA_NO_SPLIT="1", KSA_GRID="7",
               KSA_LDS_PAD_KB="40", KSA_FS_SCRATCH_MB="64", KSA_LIB="/nonexistent/libksa.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = {int(k): v for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items()}
    assert got == base
    nm = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(ROOT, "prgs-sdr-kspecanal_amd", "libksa.so")],
                        capture_output=True, text=True)
    assert nm.returncode == 0 and "getenv" not in nm.stdout


# ------------------------------------------------------------------------------- BASELINE configs[3]: quickFullScan, band shard across 8
def test_band_sharded_quickfullscan_golden_eight_ranks(ksa, torch_cuda):
    """the configuration BASELINE configs[3] literally names -- the 1 226-band quickFullScan pass of the
    reference-run golden through ksa_scan_allstitch with 8 engines (153-154 bands each: the pass-major share, 32-column
    halos from the left neighbour), engine r on GPU r % device_count.  Curves against the golden's sampled bins and
    checksums, waterfall rows against its rows, ring identical on every engine (K:622-668, K:696-697)."""
    from test_gpu_round3 import _own_spectra, dev_of
    torch = torch_cuda
    g = golden("scan_quickfull_n64")
    n, full, passes, steps = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"])
    assert n == 64 and steps == 1226
    x = _regen_iq(g, full * steps * passes).reshape(passes, steps, full)
    x_dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    world = 8
    ranks = [_scan_engine(ksa, g, 154 * passes, device=dev_of(torch, r)) for r in range(world)]
    own, shares = [], []
    for r, eng in enumerate(ranks):
        lo, hi, nhalo, e_lo, e_hi = eng.scan_shard(steps, r, world)
        shares.append(hi - lo)
        assert nhalo == (1 if r else 0)
        own.append(_own_spectra(ksa, torch, eng, x_dev, passes, steps, lo, hi))
    assert sorted(set(shares)) == [153, 154] and sum(shares) == steps
    ksa.scan_allstitch(ranks, own, steps, passes)
    st = ksa.scan_gather_state(ranks, steps)
    rings = [e.scan_state() for e in ranks]
    st.update(fftHM=rings[0]["fftHM"])
    assert all(rg["hm_index"] == int(g["hm_index"]) and rg["passes"] == passes for rg in rings)
    assert all(np.array_equal(rg["fftHM"], rings[0]["fftHM"]) for rg in rings)
    _check_sampled(st, g, "quickFullScan 8 ranks")
    assert_db(rings[0]["fftHM"][:passes], g["hm_rows"][:passes], what="quickFullScan 8 ranks waterfall rows")
    # and bit for bit what ONE engine gives for the same passes
    one = _scan_engine(ksa, g, steps * passes)
    one.scan_passes_dev(x_dev, ksa.FMT_C64, steps, passes)
    want = one.scan_state()
    for k in CURVES:
        assert np.array_equal(st[k], want[k]), k
    assert np.array_equal(rings[0]["fftHM"], want["fftHM"])
    for e in ranks + [one]:
        e.close()


def test_scan_allstitch_refuses_before_touching_any_engine(ksa, torch_cuda):
    """every pointer is validated before the first launch -- a null own_db for a rank that owns bands must
    leave the pass counters and the state of ALL engines as they were, and the caller's current device is handed back."""
    torch = torch_cuda
    n, full = 256, 2048
    mk = lambda: ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, min_amp=1e-7, xres=64,
                                    max_frames=16, scan_total_entries=2 * n)
    a, b = mk(), mk()
    steps, passes = 4, 2
    own = torch.zeros((passes, 2, n), dtype=torch.float32, device="cuda")
    before = [e.scan_state() for e in (a, b)]
    cur = torch.cuda.current_device()
    with pytest.raises(ksa.KsaError, match="own_db_dev"):
        ksa.scan_allstitch([a, b], [own, None], steps, passes)
    after = [e.scan_state() for e in (a, b)]
    for x, y in zip(before, after):
        assert x["passes"] == y["passes"] == 0 and all(np.array_equal(x[k], y[k]) for k in CURVES + ("fftHM",))
    assert torch.cuda.current_device() == cur
    ksa.scan_allstitch([a, b], [own, own], steps, passes)        # and the engines are still usable
    assert a.scan_state()["passes"] == passes
    for e in (a, b):
        e.close()


# ------------------------------------------------------------------------------- the per-frame hand-off
def test_read_view_equals_full_state(ksa, torch_cuda):
    """ksa_read_view (levels + markers + newest ring rows in one call) against the separate full-width reads, zeroSpan and
    scan; ksa_read_hm_rows across the ring's wrap."""
    n, full, xres = 4096, 32768, 512
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=xres, max_frames=1)
    x = orc.synth_iq(full * 131, 12).astype(np.complex64).reshape(131, full)
    for f in range(131):
        eng.frame(x[f])
    st = eng.state()
    lv, idx, lvl, rows, hm_index = eng.view(xres, "AVG", curve="max", min_sep=0.025 * xres, count=5, hm_rows=3)
    assert hm_index == st["hm_index"] == 131 % 128
    assert np.array_equal(lv, eng.levels(xres, "AVG"))
    i2, l2 = eng.highs(xres, "AVG", "max", min_sep=0.025 * xres, count=5)
    assert np.array_equal(idx, i2) and np.array_equal(lvl, l2) and len(idx) == 5
    assert np.array_equal(rows, st["fftHM"][[0, 1, 2]])                       # frames 128, 129, 130 -> ring rows 0, 1, 2
    assert np.array_equal(eng.hm_rows(126, 4), st["fftHM"][[126, 127, 0, 1]])  # wraps
    lv0, idx0, lvl0, rows0, _ = eng.view(xres, "MAX", hm_rows=0)             # no markers, no rows
    assert len(idx0) == 0 and rows0.shape[0] == 0 and np.array_equal(lv0, eng.levels(xres, "MAX"))
    with pytest.raises(ksa.KsaError):
        eng.view(xres, "AVG", scan=True)
    eng.close()


def test_zero_span_handoff_moves_xres_sized_data(ksa, torch_cuda):
    """kspecanal.zero_span's per-frame hand-off (K:477-504): with a decimating pltCompress only the four xRes-point
    curves, the markers and ONE waterfall row cross PCIe per frame (d['handoff.bytes']); the host copy of the ring
    assembled row by row, d['Levels'] and d['Highs'] equal what the full state gives; RAW plots still get full arrays."""
    import importlib
    K = importlib.import_module("prgs-sdr-kspecanal_amd.kspecanal")
    seen = []
    orig = K._handoff

    def spy(d, eng, freqs, scan=False):
        orig(d, eng, freqs, scan)
        st = eng.scan_state() if scan else eng.state()
        seen.append((d["handoff.bytes"], np.array_equal(d["fftHM"], st["fftHM"]), d["fftHMIndex"] == st["hm_index"],
                     dict(d.get("Levels", {})), list(d.get("Highs", [])), eng.levels(d["xRes"], d["pltCompress"], scan=scan) if K._view_on_device(d, len(freqs)) else None))
    K._handoff = spy
    try:
        d = K.main(["zeroSpan", "fftSize", "4096", "window", "hanning", "prgLoopCnt", "5", "centerFreq", "100.3e6", "xRes", "256",
                    "pltCompress", "MAX", "bPltLevels", "false", "bPltHeatMap", "false", "source", "synth"])
        assert len(seen) == 5
        for nbytes, ring_ok, idx_ok, levels, highs, lv in seen:
            assert nbytes <= (4 * 256 + 256 + 2 * 5) * 4 and ring_ok and idx_ok
            assert np.array_equal(levels["cur"], lv[0]) and np.array_equal(levels["max"], lv[1]) and len(highs) == 5
        full_bytes = (4 * 4096 + 128 * 256) * 4
        assert seen[-1][0] * 20 < full_bytes
        assert d["Fft.Cur"].shape == (4096,) and d["fftHM"].shape == (128, 256)      # materialised once at the end
        seen.clear()
        d = K.main(["zeroSpan", "fftSize", "4096", "prgLoopCnt", "2", "xRes", "256", "pltCompress", "RAW", "bPltLevels", "false",
                    "bPltHeatMap", "false", "source", "synth"])
        assert [s[0] for s in seen] == [full_bytes] * 2 and all(s[1] and s[2] for s in seen)
        seen.clear()
        d = K.main(["scan", "startFreq", "100e6", "endFreq", "104.8e6", "fftSize", "1024", "window", "hanning", "prgLoopCnt", "3",
                    "xRes", "128", "bPltLevels", "false", "bPltHeatMap", "false", "source", "synth"])
        assert len(seen) == 3 and all(s[1] and s[2] for s in seen)
        assert all(s[0] <= (4 * 128 + 128 + 2 * 5) * 4 for s in seen)
        assert d["Fft.Avg"].shape == (2 * 1024,)
    finally:
        K._handoff = orig


# ------------------------------------------------------------------------------- first contact with a multi-GPU node
def _node_devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("cfg,extra", [(2, ["--frames", "1024"]), (3, ["--passes", "4"]), (4, ["--passes", "8"]), (5, ["--frames", "16"])])
def test_bench_lines_prove_their_topology_gloo_rehearsal(cfg, extra):
    """`bench.py --gpus 3` over gloo on this GPU: the N > 1 line carries the `ranks` block gathered from the ranks
    themselves, the cross-rank state digest, and (zeroSpan) the strong sub-record -- what a node run will print, here with
    three ranks sharing one device (and saying so)."""
    d = _bench(["--gpus", "3", "--config", str(cfg), "--steps", "2", "--warmup", "1", "--no-cpu"] + extra, env={"KSA_BENCH_BACKEND": "gloo"})
    rk = d["ranks"]
    assert rk["backend"] == "gloo" and rk["world_size"] == 3 and len(rk["per_rank"]) == 3 and rk["rccl_version"] is None
    assert sorted(r["rank"] for r in rk["per_rank"]) == [0, 1, 2] and len({r["pid"] for r in rk["per_rank"]}) == 3
    assert all("name" in r and "host" in r and "device" in r for r in rk["per_rank"])
    assert rk["distinct_devices"] == min(3, _node_devices()) and ("REHEARSAL" in d["multi_gpu_note"]) == (rk["distinct_devices"] < 3)
    assert d["state_identical_across_ranks"] is True and d["ranks_differing_from_rank0"] == [] and len(d["state_sha256_rank0"]) == 64
    if cfg in (2, 5):
        s = d["strong"]
        assert d["scaling"] == "weak" and s["scaling"] == "strong" and s["frames_per_gpu_per_step"] * 3 <= d["config"]["frames_per_gpu_per_step"]
        assert s["value"] > 0 and s["ms_per_step"] > 0
    else:
        assert d["scaling"] == "strong" and "strong" not in d
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "valu", "hbm (Z round trip)") and rf["limiter"]


@pytest.mark.parametrize("cfg,extra", [(2, ["--frames", "1024"]), (5, ["--frames", "16"]), (3, ["--passes", "4"]), (4, ["--passes", "8"])])
def test_bench_inprocess_leg(cfg, extra):
    """`bench.py --inprocess --gpus 4`: one process, four engines (engine r on GPU r % device_count) merged by
    ksa_allreduce_state / ksa_scan_allstitch -- the torch-free multi-GPU form has a bench leg of its own."""
    d = _bench(["--inprocess", "--gpus", "4", "--config", str(cfg), "--steps", "2", "--warmup", "1", "--no-cpu"] + extra)
    assert d["n_gpus"] == 4 and d["value"] > 0 and "inprocess" in d["config"]["driver"]
    assert d["ranks"]["world_size"] == 4 and d["ranks"]["distinct_devices"] == min(4, _node_devices())
    assert d["state_identical_across_ranks"] is True


def test_bench_nccl_over_every_visible_device():
    """On a node: `bench.py --gpus device_count` over RCCL, one rank per GPU.  The record must show device_count distinct
    devices and bit-identical state on every rank.  Skipped on the one-GPU box."""
    nd = _node_devices()
    if nd < 2:
        pytest.skip("needs >= 2 GPUs (RCCL with N > 1 ranks)")
    for cfg, extra in ((2, ["--frames", "4096"]), (3, ["--passes", "16"]), (4, ["--passes", "32"]), (5, ["--frames", "64"])):
        d = _bench(["--gpus", str(nd), "--config", str(cfg), "--steps", "3", "--warmup", "1", "--no-cpu"] + extra)
        rk = d["ranks"]
        assert rk["backend"] == "nccl" and rk["world_size"] == nd and rk["distinct_devices"] == nd and rk["rccl_version"]
        assert d["state_identical_across_ranks"] is True and "REHEARSAL" not in d["multi_gpu_note"]


def _zerospan_nccl_rank(rank, world, port, out_path):
    import importlib
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p_ in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import ksa_oracle as orc_
    ksa = importlib.import_module("prgs-sdr-kspecanal_amd")
    dmod = importlib.import_module("prgs-sdr-kspecanal_amd.distributed")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    n, full, fpr = 1024, 8192, 150
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=128, max_frames=fpr, device=rank,
                             stream=torch.cuda.current_stream().cuda_stream)
    run = dmod.ShardedZeroSpan(eng, rank, world)
    for step in range(2):
        x = orc_.synth_iq(full * fpr * world, 900 + step).astype(np.complex64).reshape(world, fpr, full)
        run.step(torch.view_as_real(torch.from_numpy(x[rank])).cuda(), ksa.FMT_C64, fpr)
    torch.cuda.synchronize()
    st = eng.state()
    np.savez(out_path % rank, hm_index=st["hm_index"], frames=st["frames"], **{k: st[k] for k in CURVES + ("fftHM",)})
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_rccl_merge_path_world_device_count(ksa, torch_cuda, tmp_path):
    """test_rccl_merge_path_single_rank's sibling for a node: world = device_count ranks over RCCL, one GPU each
    (distributed.ShardedZeroSpan: one all-gather of [4N + 128W] per step + ksa_merge_gathered_dev).  Every rank must end
    with the state of ONE engine that ran the whole run, the same bits on every rank.  Skipped on the one-GPU box."""
    import socket
    import torch.multiprocessing as mp
    torch = torch_cuda
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("needs >= 2 GPUs (RCCL with N > 1 ranks)")
    n, full, fpr = 1024, 8192, 150
    one = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=128, max_frames=fpr * world)
    for step in range(2):
        x = orc.synth_iq(full * fpr * world, 900 + step).astype(np.complex64).reshape(world * fpr, full)
        one.frames_dev(torch.view_as_real(torch.from_numpy(x)).cuda(), ksa.FMT_C64, fpr * world)
    want = one.state()
    one.close()
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_zerospan_nccl_rank, args=(world, port, out), nprocs=world, join=True)
    first = np.load(out % 0)
    for r in range(world):
        got = np.load(out % r)
        assert int(got["frames"]) == want["frames"] and int(got["hm_index"]) == want["hm_index"]
        for k in CURVES + ("fftHM",):
            assert_db(got[k], want[k], what="%s on rank %d" % (k, r))
            assert np.array_equal(got[k], first[k]), "rank %d differs from rank 0 in %s" % (r, k)


def test_bench_under_torch_distributed_run_launcher():
    """The driver's exact command shape for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- rehearsed with two gloo ranks on this
    GPU: rank 0 prints exactly one JSON line carrying the N > 1 fields."""
    import socket
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ, KSA_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--frames", "512", "--no-cpu"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["config"]["baseline_config"] == 2
    assert d["ranks"]["world_size"] == 2 and d["state_identical_across_ranks"] is True and d["strong"]["frames_per_gpu_per_step"] == 256
