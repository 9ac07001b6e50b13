"""GPU parity: libksa (through the C ABI) against the oracle and the reference-generated golden
vectors.  Run on the MI355X box with `pytest -m gpu`.

Tolerances (fp32 device arithmetic vs the float64 reference):
  * linear spectra: max|got - want| / max|want| <= 1e-5   (BASELINE.json north_star)
  * dB curves: compared in the linear domain by the same rule (10**(dB/10)), plus an absolute
    5e-3 dB bound on bins within 60 dB of the peak (near-zero bins have unbounded relative error).
"""
import numpy as np
import pytest

import ksa_oracle as orc
from conftest import golden

pytestmark = pytest.mark.gpu

WINDOWS = ("ones", "hanning", "hamming", "kaiser")
MODES = ("AVG", "MAX", "MIN", "RAW")
GAIN = 19.1


def nerr(got, want):
    want = np.asarray(want, dtype=np.float64)
    return float(np.max(np.abs(np.asarray(got, dtype=np.float64) - want)) / np.max(np.abs(want)))


def assert_lin(got, want, tol=1e-5, what="", top=None):
    """Normalised error max|d| / max|X| <= tol.  `top`: the normaliser when `want` is a DERIVED curve that may hold no strong
    bin of its own (the scan's Min curve is a dB-domain mixture of tone and noise bins): the strongest linear value of the
    spectra behind it -- the north star's max|X| -- instead of the curve's own maximum."""
    if top is None:
        e = nerr(got, want)
    else:
        e = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))) / top)
    assert e <= tol, "%s normalised error %.3g > %.3g" % (what, e, tol)


def assert_db(got, want, what="", top=None):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    fin = np.isfinite(want)
    # -inf is log10 of an exactly zero magnitude.  float64 reproduces exact cancellations (quantised uint8 input
    # under a rectangular window has them) that fp32 twiddles turn into ~1e-8 of the strongest bin, so a -inf may
    # face a finite value as long as that value is zero within the linear tolerance.
    lin_got = np.where(np.isneginf(got), 0.0, 10 ** (np.where(np.isneginf(got), 0.0, got) / 10))
    lin_want = np.where(np.isneginf(want), 0.0, 10 ** (np.where(np.isneginf(want), 0.0, want) / 10))
    differ = np.isneginf(got) != np.isneginf(want)
    if differ.any():
        top = np.max(lin_want[np.isfinite(lin_want)])
        assert np.all(np.maximum(lin_got[differ], lin_want[differ]) <= 1e-5 * top), what + " -inf pattern"
    ok = np.isfinite(lin_want) & np.isfinite(lin_got)
    assert np.array_equal(np.isnan(got), np.isnan(want)), what + " NaN pattern"
    assert_lin(lin_got[ok], lin_want[ok], what=what, top=top)
    fin = fin & np.isfinite(got)
    # On top of the north-star tolerance (normalised linear error <= 1e-5, above): bins within 30 dB of the
    # strongest one must agree to 0.005 dB.  fp32 transform noise is ~1e-7 of the strongest bin, i.e. 4e-4 dB
    # at -30 dB (ten-fold margin); at -60 dB the same noise is 0.4 dB, so a dB bound there would only test luck
    # (a 500-case soak run found 0.008 dB at -55 dB in a MIN fold over 1555 kaiser-windowed frames).
    strong = fin & (want > np.max(want[fin]) - 30)
    d = np.max(np.abs(got[strong] - want[strong]))
    assert d <= 5e-3, "%s dB error %.3g" % (what, d)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


# ------------------------------------------------------------------------------------------ curscan
@pytest.mark.parametrize("tag", ["n64_q01", "n512_q01", "n512_q05", "n4096_q05"])
def test_curscan_vs_reference_golden(ksa, tag):
    g = golden("curscan_" + tag)
    n, q, full = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"])
    checked = 0
    for w in WINDOWS:
        for m in MODES:
            key = "%s_%s" % (w, m)
            if key not in g.files:
                continue
            eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=w, cumu_mode=m)
            assert_lin(eng.curscan(g["iq"]), g[key], what=tag + " " + key)
            eng.close()
            checked += 1
    assert checked >= 6


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("q", [0.5, 0.25, 0.1])
def test_curscan_every_plan_size(ksa, n, q):
    full = orc.full_size(n, 2.4e6)
    x = orc.synth_iq(full, 1000 + n).astype(np.complex64)
    for w, m in (("hanning", "AVG"), ("kaiser", "MAX"), ("ones", "MIN")):
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=w, cumu_mode=m)
        assert eng.num_windows == len(orc.window_starts(full, n, q))
        want = orc.curscan(x, n, q, orc.window_table(w, n), m)
        assert_lin(eng.curscan(x), want, what="N=%d q=%s %s %s" % (n, q, w, m))
        eng.close()


@pytest.mark.parametrize("tag", ["n8192_q05", "n16384_q01"])
def test_curscan_large_n_golden(ksa, tag):
    g = golden("curscan_" + tag)
    n, q, full = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"])
    x = orc.synth_iq(full, int(g["seed"])).astype(np.complex64)
    for mode, key, dec in (("AVG", "avg_at_idx", "avg_decim"), ("MAX", "max_at_idx", "max_decim")):
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=str(g["window"]), cumu_mode=mode)
        y = eng.curscan(x)
        peak = float(g["peak"])
        assert np.max(np.abs(y[g["idx"]] - g[key])) / peak <= 1e-5
        red = y.reshape(256, -1).sum(axis=1) if mode == "AVG" else y.reshape(256, -1).max(axis=1)
        assert np.max(np.abs(red - g[dec])) / np.max(g[dec]) <= 1e-5
        eng.close()


@pytest.mark.parametrize("tag", ["n32768_q05", "n65536_q025"])
def test_curscan_four_step_golden(ksa, tag):
    """N > 16384 runs a radix-16 / 32 / 64 DIF stage in front of the single-workgroup kernel (path 2); config 5 geometry
    (65536, 75 % overlap, 29 windows)."""
    g = golden("curscan_" + tag)
    n, q, full = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"])
    x = orc.synth_iq(full, int(g["seed"])).astype(np.complex64)
    for mode, key, dec in (("AVG", "avg_at_idx", "avg_decim"), ("MAX", "max_at_idx", "max_decim")):
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=str(g["window"]), cumu_mode=mode)
        assert eng.kernel_info()["path"] == 2
        y = eng.curscan(x)
        peak = float(g["peak"])
        assert np.max(np.abs(y[g["idx"]] - g[key])) / peak <= 1e-5
        red = y.reshape(256, -1).sum(axis=1) if mode == "AVG" else y.reshape(256, -1).max(axis=1)
        assert np.max(np.abs(red - g[dec])) / np.max(g[dec]) <= 1e-5
        eng.close()


@pytest.mark.parametrize("n,q,fmt", [(32768, 0.5, "c64"), (65536, 0.25, "u8"), (131072, 0.5, "c64"), (262144, 0.3, "u8"), (524288, 0.5, "c64"), (1048576, 0.5, "c64"),
                                     (524288, 0.3, "u8"), (1048576, 0.25, "u8")])
def test_four_step_full_spectrum_vs_oracle(ksa, n, q, fmt):
    full = 2 * n
    x = orc.synth_iq(full, 4000 + (n >> 10))
    if fmt == "u8":
        raw = orc.quantize_u8(x * 0.8)
        xin, arg = orc.unpack_u8(raw), raw
    else:
        xin = arg = x.astype(np.complex64)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning", cumu_mode="AVG")
    want = orc.curscan(xin, n, q, orc.window_table("hanning", n), "AVG")
    assert_lin(eng.curscan(arg), want, what="four-step N=%d" % n)
    eng.close()


def test_four_step_zerospan_state(ksa, torch_cuda):
    """Config 5 shape end to end: frames_dev on the four-step path incl. waterfall rows (g = 128 bins per cell)."""
    torch = torch_cuda
    n, q, full, frames = 65536, 0.25, 524288, 3
    x = orc.synth_iq(full * frames, 555).astype(np.complex64).reshape(frames, full)
    st_ref, db_ref, _ = orc.zerospan_batch(x, n, q, orc.window_table("hanning", n), "AVG", GAIN, 512)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning", gain=GAIN, max_frames=frames)
    assert eng.num_windows == 29
    dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    rows = torch.empty((frames, 512), dtype=torch.float32, device="cuda")
    eng.frames_dev(dev, ksa.FMT_C64, frames, hm_rows=rows)
    st = eng.state()
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], getattr(st_ref, k), what="four-step " + k)
    assert_db(st["fftHM"][:frames], st_ref.hm[:frames], what="four-step waterfall")
    assert_db(rows.cpu().numpy(), st_ref.hm[:frames], what="four-step rows")
    eng.close()


def test_on_bin_tone_known_answer(ksa):
    g = golden("tone_n4096")
    for w in WINDOWS:
        eng = ksa.SpectrumEngine(4096, full_size=32768, non_overlap=0.5, window=w)
        y = eng.curscan(g["iq"])
        assert int(np.argmax(y)) == 2560 and abs(y[2560] - 1.0) < 1e-5
        assert_lin(y, g[w], what="tone " + w)
        eng.close()


@pytest.mark.parametrize("n,q", [(64, 0.1), (512, 0.5), (4096, 0.5), (16384, 0.1)])
def test_curscan_uint8_input(ksa, n, q):
    full = orc.full_size(n, 2.4e6)
    raw = orc.quantize_u8(orc.synth_iq(full, 77 + n) * 0.8)
    want = orc.curscan(orc.unpack_u8(raw), n, q, orc.window_table("hanning", n), "AVG")
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning")
    assert_lin(eng.curscan(raw), want, what="u8 N=%d" % n)
    eng.close()


def test_bad_arguments_raise(ksa):
    with pytest.raises(ksa.KsaError):
        ksa.SpectrumEngine(1000)                       # not a power of two
    with pytest.raises(ksa.KsaError):
        ksa.SpectrumEngine(4096, window="blackman")
    with pytest.raises(ksa.KsaError):
        ksa.SpectrumEngine(4096, cumu_mode="MEDIAN")
    eng = ksa.SpectrumEngine(512, non_overlap=0.5)
    with pytest.raises(ksa.KsaError):
        eng.curscan(np.zeros(100, dtype=np.complex64))  # ragged block
    eng.close()


# ------------------------------------------------------------------------------------- zeroSpan state
@pytest.mark.parametrize("tag", ["n512", "n4096", "n64", "hm_n512"])
def test_zerospan_frames_vs_reference_golden(ksa, tag):
    g = golden("zerospan_" + tag)
    n, q, full, frames = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"]), int(g["frames"])
    x = g["iq"].reshape(frames, full)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=str(g["window"]),
                             gain=float(g["gain"]), xres=int(g["xres"]))
    for f in range(frames):
        eng.frame(x[f])
    st = eng.state()
    assert st["frames"] == frames and st["hm_index"] == frames % 128
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], g[k], what=tag + " " + k)
    if "hm" in g.files:
        want = g["hm"]
        assert st["fftHM"].shape == want.shape
        assert_db(st["fftHM"][:frames], want[:frames], what=tag + " hm")
        assert np.all(st["fftHM"][frames:] == 0)     # untouched rows keep np.zeros (K:456)
    eng.close()


@pytest.mark.parametrize("n,q,fmt", [(4096, 0.5, "c64"), (512, 0.1, "c64"), (64, 0.1, "u8"), (2048, 0.5, "u8")])
def test_zerospan_batched_device_path(ksa, torch_cuda, n, q, fmt):
    """frames_dev over a batch == the oracle's sequential loop, incl. a second batch on top (EMA carry)."""
    torch = torch_cuda
    full = orc.full_size(n, 2.4e6)
    f1, f2 = 37, 150
    x = orc.synth_iq(full * (f1 + f2), 5 + n).astype(np.complex64).reshape(f1 + f2, full)
    if fmt == "u8":
        raw = orc.quantize_u8(x.reshape(-1) * 0.8).reshape(f1 + f2, 2 * full)
        host, xin = raw, orc.unpack_u8(raw.reshape(-1)).reshape(f1 + f2, full)
        dev = torch.from_numpy(raw).cuda()
        code = ksa.FMT_U8
    else:
        xin = x
        dev = torch.view_as_real(torch.from_numpy(x)).cuda()
        code = ksa.FMT_C64
    win = orc.window_table("hanning", n)
    st_ref, db_ref, _ = orc.zerospan_batch(xin, n, q, win, "AVG", GAIN, 512)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning", gain=GAIN, max_frames=f2)
    cur_db = torch.empty((f2, n), dtype=torch.float32, device="cuda")
    rows = torch.empty((f2, eng.hm_width), dtype=torch.float32, device="cuda")
    eng.frames_dev(dev[:f1], code, f1, cur_db=cur_db, hm_rows=rows)
    eng.frames_dev(dev[f1:], code, f2, cur_db=cur_db, hm_rows=rows)
    torch.cuda.synchronize()
    st = eng.state()
    assert st["frames"] == f1 + f2 and st["hm_index"] == (f1 + f2) % 128
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], getattr(st_ref, k), what="N=%d %s" % (n, k))
    assert_db(cur_db.cpu().numpy(), db_ref[f1:], what="per-frame dB")
    assert_db(st["fftHM"], st_ref.hm, what="waterfall ring")
    want_rows = np.array([orc.plotcompress(r, eng.hm_width, "MAX") for r in db_ref[f1:]])
    assert_db(rows.cpu().numpy(), want_rows, what="per-frame rows")
    eng.close()


def test_zero_magnitude_keeps_minus_inf(ksa):
    """K:469: zeroSpan keeps log10(0) = -inf; it poisons Min and Avg by IEEE rules."""
    n, full = 512, 4096
    frames = [orc.synth_iq(full, 11), np.zeros(full), orc.synth_iq(full, 12)]
    win = orc.window_table("hanning", n)
    st_ref, _, _ = orc.zerospan_batch(np.array(frames), n, 0.5, win, "AVG", GAIN, 512)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN)
    for fr in frames:
        eng.frame(fr.astype(np.complex64))
    st = eng.state()
    assert np.all(np.isneginf(st_ref.min)) and np.all(np.isneginf(st["Fft.Min"]))
    assert np.all(np.isneginf(st_ref.avg)) and np.all(np.isneginf(st["Fft.Avg"]))
    assert_db(st["Fft.Max"], st_ref.max, what="max")
    assert_db(st["Fft.Cur"], st_ref.cur, what="cur")
    eng.close()


def test_zerospan_play_spectra(ksa):
    """zeroSpanPlay (config 1): saved linear spectra drive the accumulate + waterfall only."""
    g = golden("zerospan_save_n512")
    n, frames = int(g["fft_size"]), int(g["frames"])
    eng = ksa.SpectrumEngine(n, non_overlap=0.5, window="hanning", gain=float(g["header"][2]))
    for f in range(frames):
        eng.frame_spectrum(g["spectra"][f])
    st = eng.state()
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], g["play_" + k], what="play " + k)
    # and the spectra themselves are what curscan produces for the recorded IQ
    x = g["iq"].reshape(frames, int(g["full"]))
    assert_lin(eng.curscan(x[2]), g["spectra"][2], what="saved spectrum")
    eng.close()


def test_adj_siglvls_and_flags(ksa):
    n, full = 512, 4096
    x = orc.synth_iq(full * 3, 21).astype(np.complex64).reshape(3, full)
    adj = np.linspace(-3, 3, n)
    win = orc.window_table("kaiser", n)
    st_ref = orc.ZeroSpanState(n, 128, GAIN, adj=adj, b_min=False)
    for fr in x:
        st_ref.push(orc.curscan(fr, n, 0.5, win, "AVG"))
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="kaiser", gain=GAIN, xres=128)
    eng.set_adj(adj)
    eng.set_flags(b_max=True, b_min=False, b_avg=True)
    for fr in x:
        eng.frame(fr)
    st = eng.state()
    assert_db(st["fftHM"][:3], st_ref.hm[:3], what="adj waterfall")
    assert_db(st["Fft.Avg"], st_ref.avg, what="avg")
    assert np.all(st["Fft.Min"] == 0)      # never written when bDataMin is off (reference leaves None)
    eng.close()


# ------------------------------------------------------------------------------------------------ scan
@pytest.mark.parametrize("tag", ["3band_n512", "frac_n256", "quick_n64", "baseraw_n256"])
def test_scan_vs_reference_golden(ksa, torch_cuda, tag):
    torch = torch_cuda
    g = golden("scan_" + tag)
    n, full = int(g["fft_size"]), int(g["full"])
    passes, steps = int(g["passes"]), int(g["steps"])
    groups = int((float(g["end_freq"]) - float(g["start_freq"])) / float(g["sampling_rate"]))
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=float(g["non_overlap"]), window=str(g["window"]),
                             gain=float(g["gain"]), min_amp=float(g["min_amp"]), xres=int(g["xres"]),
                             max_frames=steps, scan_total_entries=groups * n,
                             scan_non_overlap=float(g["scan_non_overlap"]))
    eng.scan_set_base_is_raw(bool(g["base_is_raw"]))
    x = torch.view_as_real(torch.from_numpy(g["iq"].reshape(passes, steps, full))).cuda()
    for p in range(passes):
        eng.scan_pass_dev(x[p], ksa.FMT_C64, steps)
    st = eng.scan_state()
    assert st["passes"] == passes and st["hm_index"] == int(g["hm_index"])
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], g[k], what=tag + " " + k)
    want_hm = g["hm"]
    assert_db(st["fftHM"][:passes], want_hm[:passes], what=tag + " hm")
    assert np.allclose(st["fftHM"][passes:], want_hm[passes:], rtol=1e-6)   # untouched rows = minAmp4Clip
    eng.close()


def test_scan_dummy_band(ksa, torch_cuda):
    """A band whose tune failed is replaced by ones(fftSize) (K:637-639)."""
    torch = torch_cuda
    n, full, fs = 256, 2048, 2.4e6
    start, end = 100e6, 104.8e6
    steps = len(orc.scan_steps(start, end, fs, 0.5))
    x = orc.synth_iq(full * steps, 31).astype(np.complex64).reshape(steps, full)
    win = orc.window_table("hanning", n)
    ok = np.ones(steps, dtype=np.uint8)
    ok[1] = 0
    ref = orc.ScanState(n, start, end, fs, GAIN, 1e-7, 64)
    ref.run_pass([orc.curscan(x[s], n, 0.1, win, "AVG") if ok[s] else None for s in range(steps)])
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.1, window="hanning", gain=GAIN, min_amp=1e-7,
                             xres=64, max_frames=steps, scan_total_entries=ref.total)
    eng.scan_pass_dev(torch.view_as_real(torch.from_numpy(x)).cuda(), ksa.FMT_C64, steps, step_ok=ok)
    st = eng.scan_state()
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], getattr(ref, k), what="dummy " + k)
    eng.close()


# ------------------------------------------------------------------------- size-independent properties
def test_full_size_batch_properties(ksa, torch_cuda):
    """Config-2 shape at bench scale: replicated frames must give identical rows, Max == Min == Cur,
    and scaling the input by 4 moves every dB value by 10*log10(4)."""
    torch = torch_cuda
    n, full, frames = 4096, 32768, 2048
    base = orc.synth_iq(full, 99).astype(np.complex64)
    one = torch.view_as_real(torch.from_numpy(base)).cuda()
    dev = one.unsqueeze(0).expand(frames, full, 2).contiguous()
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, max_frames=frames)
    cur = torch.empty((frames, n), dtype=torch.float32, device="cuda")
    eng.frames_dev(dev, ksa.FMT_C64, frames, cur_db=cur)
    torch.cuda.synchronize()
    assert bool((cur == cur[0]).all())
    st = eng.state()
    assert np.array_equal(st["Fft.Max"], st["Fft.Min"]) and np.array_equal(st["Fft.Max"], st["Fft.Cur"])
    assert np.max(np.abs(st["Fft.Avg"] - st["Fft.Cur"])) < 1e-4
    want = orc.log_no_gain(orc.curscan(base, n, 0.5, orc.window_table("hanning", n)), GAIN)
    assert_db(st["Fft.Cur"], want, what="bench-size frame")
    eng.reset()
    eng.frames_dev(dev * 4.0, ksa.FMT_C64, frames, cur_db=cur)
    st4 = eng.state()
    assert np.max(np.abs(st4["Fft.Cur"] - st["Fft.Cur"] - 10 * np.log10(4.0))) < 1e-4
    eng.close()


def test_sharded_scan_driver_world1(ksa, torch_cuda):
    """distributed.ShardedScan on one rank: per-step clipped dB spectra -> scan_stitch_dev == scan_pass_dev."""
    torch = torch_cuda
    dmod = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    g = golden("scan_quick_n64")
    n, full = int(g["fft_size"]), int(g["full"])
    passes, steps = int(g["passes"]), int(g["steps"])
    groups = int((float(g["end_freq"]) - float(g["start_freq"])) / float(g["sampling_rate"]))
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=float(g["non_overlap"]), window=str(g["window"]),
                             gain=float(g["gain"]), min_amp=float(g["min_amp"]), xres=int(g["xres"]),
                             max_frames=steps, scan_total_entries=groups * n)
    x = torch.view_as_real(torch.from_numpy(g["iq"].reshape(passes, steps, full))).cuda()
    run = dmod.ShardedScan(eng)
    for p in range(passes):
        run.run_pass(x[p], ksa.FMT_C64, steps)
    st = eng.scan_state()
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], g[k], what="sharded scan " + k)
    eng.close()


def test_rccl_merge_path_single_rank(ksa, torch_cuda):
    """The multi-GPU merge of distributed.ShardedZeroSpan (a view of library memory as the send buffer of an RCCL
    all-gather, ksa_merge_gathered_dev) driven on a one-rank nccl group must equal the plain path."""
    import os
    import torch.distributed as dist
    torch = torch_cuda
    dmod = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, full, frames = 1024, 8192, 300
        x = orc.synth_iq(full * 64, 808).astype(np.complex64).reshape(64, full)
        dev = torch.view_as_real(torch.from_numpy(x)).cuda().repeat(5, 1, 1)[:frames].contiguous()
        states = []
        for coll in (False, True):
            eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", max_frames=frames,
                                     stream=torch.cuda.current_stream().cuda_stream)
            run = dmod.ShardedZeroSpan(eng, 0, 1, always_collective=coll)
            run.step(dev, ksa.FMT_C64, frames)
            run.step(dev[:77], ksa.FMT_C64, 77)
            torch.cuda.synchronize()
            states.append(eng.state())
            eng.close()
        a, b = states
        assert a["frames"] == b["frames"] == 377 and a["hm_index"] == b["hm_index"] == 377 % 128
        for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg", "fftHM"):
            assert np.array_equal(a[k], b[k]), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fpr,idx0", [(3, 40, 0), (2, 100, 77), (4, 150, 5), (8, 16, 120)])
def test_gathered_merge_kernel_emulated_ranks(ksa, torch_cuda, world, fpr, idx0):
    """ksa_merge_gathered_dev with the ranks of a sharded run emulated on one GPU: every "rank" (its own engine)
    processes its time chunk uncommitted, the exchange blocks are stacked as an all-gather would deliver them,
    and each rank's merge must equal (a) a single engine running the whole run and (b) the torch restatement
    of the kernel.  Two steps, so that stale ring rows and the has-previous path are covered."""
    torch = torch_cuda
    dmod = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    n, full, xres = 1024, 4096, 128
    total = world * fpr
    x = orc.synth_iq(full * total * 2, 31 + world).astype(np.complex64).reshape(2, total, full)
    dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    mk = lambda mf: ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=xres,
                                       max_frames=mf, stream=torch.cuda.current_stream().cuda_stream)
    one = mk(total)
    one.set_hm_index(idx0)
    ranks = [mk(fpr) for _ in range(world)]
    hm_index = idx0
    for step in range(2):
        one.frames_dev(dev[step], ksa.FMT_C64, total)
        for r, eng in enumerate(ranks):
            eng.set_hm_index((hm_index + r * fpr) % 128)
            eng.frames_dev(dev[step, r * fpr:(r + 1) * fpr], ksa.FMT_C64, fpr, first_index=r * fpr, total_frames=total,
                           commit=False)
        gathered = torch.stack([torch.as_tensor(eng.exchange(), device="cuda").clone() for eng in ranks])
        ref_partial, ref_ring, defined = dmod.merge_gathered_reference(gathered, n, xres, hm_index, fpr)
        for eng in ranks:
            eng.merge_gathered(gathered, world, fpr, hm_index)
        torch.cuda.synchronize()
        hm_index = (hm_index + total) % 128
        want = one.state()
        for r, eng in enumerate(ranks):
            got = eng.state()
            assert got["frames"] == want["frames"] and got["hm_index"] == want["hm_index"] == hm_index
            # (tolerance, not equality: small chunks run in window-split mode, which folds in another order)
            for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg", "fftHM"):
                assert_db(got[k], want[k], what="%s rank %d step %d" % (k, r, step))
            part = torch.as_tensor(eng.partial(), device="cuda").cpu().numpy()
            assert np.array_equal(part, ref_partial.cpu().numpy()), ("partial vs torch restatement", r, step)
            rows = np.flatnonzero(defined.cpu().numpy())
            assert np.array_equal(got["fftHM"][rows], ref_ring.cpu().numpy()[rows].astype(np.float64))
    # every rank holds the same bits
    a = ranks[0].state()
    for eng in ranks[1:]:
        b = eng.state()
        assert all(np.array_equal(a[k], b[k]) for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg", "fftHM"))
    for eng in ranks + [one]:
        eng.close()


def test_merge_gathered_refusals(ksa, torch_cuda):
    torch = torch_cuda
    eng = ksa.SpectrumEngine(256, full_size=1024, non_overlap=0.5, max_frames=8)
    buf = torch.zeros((2, 4 * 256 + 128 * eng.hm_width), dtype=torch.float32, device="cuda")
    with pytest.raises(ksa.KsaError):
        eng.merge_gathered(buf, 2, 4, 0)                      # nothing pending
    x = torch.zeros((4, 1024, 2), dtype=torch.float32, device="cuda")
    eng.frames_dev(x, ksa.FMT_C64, 4, first_index=0, total_frames=8, commit=False)
    for bad in ((0, 4, 0), (2, 3, 0), (2, 4, 128), (2, 4, -1)):
        with pytest.raises(ksa.KsaError):
            eng.merge_gathered(buf, *bad)
    eng.close()


@pytest.mark.parametrize("n,xres", [(128, 64), (64, 64), (1024, 512), (4096, 16), (16384, 32), (16384, 8192), (256, 4), (65536, 64), (65536, 16), (32768, 32768), (131072, 2048), (524288, 512), (1048576, 64), (1048576, 4096)])
def test_waterfall_cell_paths(ksa, torch_cuda, n, xres):
    """Every waterfall reduction path of the output stage: g = N/W of 1, 2, 4..256 (shuffles) and > 256 (LDS),
    on the single-workgroup kernels and behind the radix-16 / 32 / 64 first stages."""
    torch = torch_cuda
    full, frames = 2 * n, 3
    x = orc.synth_iq(full * frames, 900 + n + xres).astype(np.complex64).reshape(frames, full)
    st_ref, db_ref, _ = orc.zerospan_batch(x, n, 0.5, orc.window_table("hamming", n), "AVG", GAIN, xres)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hamming", gain=GAIN, xres=xres, max_frames=frames)
    assert eng.hm_width == min(n, xres)
    rows = torch.empty((frames, eng.hm_width), dtype=torch.float32, device="cuda")
    eng.frames_dev(torch.view_as_real(torch.from_numpy(x)).cuda(), ksa.FMT_C64, frames, hm_rows=rows)
    st = eng.state()
    assert_db(rows.cpu().numpy(), st_ref.hm[:frames], what="rows N=%d W=%d" % (n, xres))
    assert_db(st["fftHM"][:frames], st_ref.hm[:frames], what="ring N=%d W=%d" % (n, xres))
    eng.close()


def test_frame_stride_and_batch_limits(ksa, torch_cuda):
    """Frames may overlap or be spaced in the device buffer (frame_stride != fullSize); batches beyond
    max_frames, misaligned outputs and empty batches are refused."""
    torch = torch_cuda
    n, full = 1024, 8192
    x = orc.synth_iq(full * 6, 321).astype(np.complex64)
    dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    win = orc.window_table("hanning", n)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", max_frames=8)
    for stride, frames in ((full // 2, 8), (full + 1024, 4), (3, 5)):
        out = torch.empty((frames, n), dtype=torch.float32, device="cuda")
        eng.curscan_dev(dev, ksa.FMT_C64, frames, out, frame_stride=stride)
        want = np.array([orc.curscan(x[f * stride:f * stride + full], n, 0.5, win, "AVG") for f in range(frames)])
        assert_lin(out.cpu().numpy(), want, what="stride %d" % stride)
    out = torch.empty((9, n), dtype=torch.float32, device="cuda")
    with pytest.raises(ksa.KsaError):
        eng.curscan_dev(dev, ksa.FMT_C64, 9, out)                      # > max_frames
    with pytest.raises(ksa.KsaError):
        eng.curscan_dev(dev, ksa.FMT_C64, 0, out)                      # empty batch
    with pytest.raises(ksa.KsaError):
        eng.curscan_dev(dev, ksa.FMT_C64, 2, out.view(-1)[1:])         # output not 16-byte aligned
    with pytest.raises(ksa.KsaError):
        eng.frames_dev(dev, ksa.FMT_C64, 4, first_index=6, total_frames=8)   # batch runs past the run
    with pytest.raises(ksa.KsaError):
        eng.commit(4)                                                  # nothing pending
    eng.close()


@pytest.mark.parametrize("frames_per_rank,ranks", [(40, 2), (200, 3), (5, 4)])
def test_time_chunk_shards_equal_one_run(ksa, torch_cuda, frames_per_rank, ranks):
    """SURVEY 8(e) on the device: `ranks` engines each process their time chunk with the run's global frame
    indices (ksa_frames_dev first_index/total_frames, commit = 0), the partial blocks are merged with the MAX /
    SUM algebra the RCCL path uses, every engine commits -- and all of them must equal one engine that saw the
    whole run in order (and the float64 oracle)."""
    torch = torch_cuda
    dmod = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    n, full = 512, 4096
    total = frames_per_rank * ranks
    x = orc.synth_iq(full * total, 1234 + total).astype(np.complex64).reshape(total, full)
    dev = torch.view_as_real(torch.from_numpy(x)).cuda()
    mk = lambda: ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=128,
                                    max_frames=total)
    one = mk()
    one.frames_dev(dev[:7], ksa.FMT_C64, 7)          # some history first, so that has_prev is exercised
    one.frames_dev(dev, ksa.FMT_C64, total)
    want = one.state()
    engines = [mk() for _ in range(ranks)]
    parts = []
    for r, eng in enumerate(engines):
        eng.frames_dev(dev[:7], ksa.FMT_C64, 7)
        eng.set_hm_index((7 + r * frames_per_rank) % 128)
        eng.frames_dev(dev[r * frames_per_rank:(r + 1) * frames_per_rank], ksa.FMT_C64, frames_per_rank,
                       first_index=r * frames_per_rank, total_frames=total, commit=False)
        parts.append(torch.as_tensor(eng.partial(), device="cuda"))
    merged = torch.stack(parts)
    mx = merged[:, 0:3].max(dim=0).values            # what all_reduce(MAX) leaves on every rank
    sm = merged[:, 3].sum(dim=0)                     # all_reduce(SUM)
    rings = []
    for eng, p in zip(engines, parts):
        p[0:3] = mx
        p[3] = sm
        eng.commit(total)
        rings.append(torch.as_tensor(eng.state_dev()[1], device="cuda").clone())
    own = dmod.ring_owner(7, frames_per_rank, ranks)
    for eng in engines:
        st = eng.state()
        assert st["frames"] == want["frames"] == 7 + total
        for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg"):
            # same terms; the summation order differs (global weights per shard, window-split folds of small batches)
            assert np.max(np.abs(st[k] - want[k])) < 2e-5, k
    ring = want["fftHM"]
    for slot in range(128):
        if own[slot] >= 0:
            assert np.max(np.abs(rings[int(own[slot])][slot].cpu().numpy().astype(np.float64) - ring[slot])) < 2e-5, slot
    ref, _, _ = orc.zerospan_batch(np.concatenate([x[:7], x]), n, 0.5, orc.window_table("hanning", n), "AVG", GAIN, 128)
    for k in ("cur", "max", "min", "avg"):
        assert_db(want["Fft." + k.capitalize()], getattr(ref, k), what="one-run " + k)
    for eng in engines + [one]:
        eng.close()


def test_device_levels_decimation(ksa):
    """data_plotcompress on the device (K:205-221, with the Fft.Adj baseline of K:400-411) vs the oracle's
    _data_plotcompress restatement on the full-size curves."""
    n, full = 2048, 16384
    x = orc.synth_iq(full * 4, 2468).astype(np.complex64).reshape(4, full)
    adj = np.sin(np.arange(n) / 50.0)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=256)
    eng.set_adj(adj)
    for fr in x:
        eng.frame(fr)
    st = eng.state()
    for mode in ("AVG", "MAX"):
        got = eng.levels(256, mode)
        for row, key in enumerate(("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg")):
            want = orc.plotcompress(st[key] - adj, 256, mode)
            assert np.max(np.abs(got[row] - want)) < 2e-5, (mode, key)
    got = eng.levels(256, "MIN")
    assert np.max(np.abs(got[0] - (st["Fft.Cur"] - adj).reshape(256, -1).min(axis=1))) < 2e-5
    with pytest.raises(ksa.KsaError):
        eng.levels(300)
    eng.close()
