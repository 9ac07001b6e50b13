"""The oracle (oracle/ksa_oracle.py) against vectors produced by executing the reference
(tests/golden/make_golden.py).  float64 vs float64: bit-exact unless stated."""
import hashlib

import numpy as np
import pytest

import ksa_oracle as orc
from conftest import golden

WINDOWS = ("ones", "hanning", "hamming", "kaiser")
MODES = ("AVG", "MAX", "MIN", "RAW")


@pytest.mark.parametrize("tag", ["n64_q01", "n512_q01", "n512_q05", "n4096_q05"])
def test_curscan_matches_reference(tag):
    g = golden("curscan_" + tag)
    n, q, full = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"])
    x = g["iq"]
    assert x.dtype == np.complex64 and len(x) == full
    checked = 0
    for w in WINDOWS:
        for m in MODES:
            key = "%s_%s" % (w, m)
            if key not in g.files:
                continue
            got = orc.curscan(x, n, q, orc.window_table(w, n), m)
            assert np.array_equal(got, g[key]), key
            checked += 1
    assert checked >= 6


def test_window_geometry_counts():
    # window counts probed on the reference: 15 / 71 / 71 / 29
    assert len(orc.window_starts(32768, 4096, 0.5)) == 15
    assert len(orc.window_starts(131072, 16384, 0.1)) == 71
    assert len(orc.window_starts(512, 64, 0.1)) == 71
    assert len(orc.window_starts(524288, 65536, 0.25)) == 29
    hops = np.diff(orc.window_starts(512, 64, 0.1))
    assert set(hops.tolist()) == {6, 7}
    hops = np.diff(orc.window_starts(131072, 16384, 0.1))
    assert set(hops.tolist()) == {1638, 1639}
    assert orc.full_size(4096, 2.4e6) == 32768 and orc.full_size(65536, 1e9) == 524288
    assert orc.full_size(2 ** 19, 2.4e6) == 2 ** 20


@pytest.mark.parametrize("tag", ["n512", "n4096", "n64", "hm_n512"])
def test_zerospan_state_matches_reference(tag):
    g = golden("zerospan_" + tag)
    n, q, full, frames = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"]), int(g["frames"])
    win = orc.window_table(str(g["window"]), n)
    x = g["iq"].reshape(frames, full)
    st, db, lin = orc.zerospan_batch(x, n, q, win, "AVG", float(g["gain"]), int(g["xres"]))
    assert np.array_equal(st.cur, g["cur"])
    assert np.array_equal(st.max, g["max"])
    assert np.array_equal(st.min, g["min"])
    assert np.array_equal(st.avg, g["avg"])
    if "hm" in g.files:
        assert np.array_equal(st.hm, g["hm"])
        assert st.hm_index == frames % 128


@pytest.mark.parametrize("tag", ["3band_n512", "frac_n256", "quick_n64", "baseraw_n256"])
def test_scan_state_matches_reference(tag):
    g = golden("scan_" + tag)
    n, full = int(g["fft_size"]), int(g["full"])
    passes, steps = int(g["passes"]), int(g["steps"])
    win = orc.window_table(str(g["window"]), n)
    st = orc.ScanState(n, float(g["start_freq"]), float(g["end_freq"]), float(g["sampling_rate"]),
                       float(g["gain"]), float(g["min_amp"]), int(g["xres"]),
                       float(g["scan_non_overlap"]), base_is_raw=bool(g["base_is_raw"]))
    assert len(st.centers) == steps
    x = g["iq"].reshape(passes, steps, full)
    for p in range(passes):
        st.run_pass([orc.curscan(x[p, s], n, float(g["non_overlap"]), win, "AVG") for s in range(steps)])
    for k in ("cur", "max", "min", "avg", "hm"):
        assert np.array_equal(getattr(st, k), g[k]), k
    # the reference advances fftHMIndex after every pass (K:732)
    assert st.hm_index == int(g["hm_index"])


def test_fmscan_and_quickfullscan_geometry():
    # SURVEY 3.2 [probed]: 88-108 MHz -> 109.6 MHz, 9 groups, 18 steps; 30e6-1.5e9 -> 1501.2 MHz, 613, 1226
    end, _ = orc.fixup_scan_range(88e6, 108e6, 2.4e6)
    assert end == 109.6e6 and len(orc.scan_steps(88e6, end, 2.4e6, 0.5)) == 18
    end, _ = orc.fixup_scan_range(30e6, 1.5e9, 2.4e6)
    assert abs(end - 1501.2e6) < 1 and len(orc.scan_steps(30e6, end, 2.4e6, 0.5)) == 1226
    assert int((end - 30e6) / 2.4e6) * 64 == 39232


def test_zerospan_save_stream_and_play():
    g = golden("zerospan_save_n512")
    n, frames, full = int(g["fft_size"]), int(g["frames"]), int(g["full"])
    x = g["iq"].reshape(frames, full)
    win = orc.window_table("hanning", n)
    for f in range(frames):
        assert np.array_equal(orc.curscan(x[f], n, 0.5, win, "AVG"), g["spectra"][f])
    # play = accumulate saved spectra only (K:547-564 feeding K:464-484)
    st = orc.ZeroSpanState(n, 512, float(g["header"][2]))
    for f in range(frames):
        st.push(np.copy(g["spectra"][f]))
    for k in ("cur", "max", "min", "avg"):
        assert np.array_equal(getattr(st, k), g["play_" + k]), k


def test_on_bin_tone_known_answer():
    # A=0.5 on bin k/N=0.125, N=4096: shifted bin 2560 reads 2A = 1.0 under every window
    g = golden("tone_n4096")
    n = 4096
    for w in WINDOWS:
        y = orc.curscan(g["iq"], n, 0.5, orc.window_table(w, n), "AVG")
        assert np.array_equal(y, g[w])
        assert int(np.argmax(y)) == 2560
        assert abs(y[2560] - 1.0) < 1e-6
    lvl = orc.log_no_gain(np.array([orc.curscan(g["iq"], n, 0.5, orc.window_table("hanning", n))[2560]]), 19.1)
    assert abs(lvl[0] - (-19.1)) < 1e-5        # 10*log10(2A) - gain with 2A = 1


@pytest.mark.parametrize("tag", ["n8192_q05", "n16384_q01", "n32768_q05", "n65536_q025"])
def test_large_n_sampled(tag):
    g = golden("curscan_" + tag)
    n, q, full = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"])
    x = orc.synth_iq(full, int(g["seed"])).astype(np.complex64)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["iq_sha256"])
    win = orc.window_table(str(g["window"]), n)
    y = orc.curscan(x, n, q, win, "AVG")
    ym = orc.curscan(x, n, q, win, "MAX")
    idx = g["idx"]
    assert np.array_equal(y[idx], g["avg_at_idx"])
    assert np.array_equal(ym[idx], g["max_at_idx"])
    assert np.array_equal(y.reshape(256, -1).sum(axis=1), g["avg_decim"])
    assert np.array_equal(ym.reshape(256, -1).max(axis=1), g["max_decim"])


def test_pieces():
    g = golden("pieces")
    a, b = g["a"], g["b"]
    for mode in MODES:
        assert np.array_equal(orc.data_cumu(mode, np.copy(a), 8, 40, b, 4, 36), g["cumu_" + mode])
    assert np.array_equal(orc.plotcompress(np.copy(a), 16, "MAX"), g["compress_MAX"])
    assert np.array_equal(orc.plotcompress(np.copy(a), 16, "AVG"), g["compress_AVG"])
    v = np.abs(a); v[3] = 0.0
    assert np.array_equal(orc.log_no_gain(np.copy(v), 19.1), g["lognogain"])
    assert np.isneginf(g["lognogain"][3])
    assert np.array_equal(orc.log_no_gain(np.copy(v), 19.1, inf_to=0), g["lognogain_inf0"])
    assert np.array_equal(orc.clip2minamp(np.copy(v) * 1e-7, (1 / 256) * 0.00001), g["clip"])


def test_avg_closed_form_weights():
    # AVG over n+1 items == x0/2^n + sum x_k / 2^(n-k+1); used by the device kernels
    rng = np.random.default_rng(7)
    xs = rng.random((9, 33))
    acc = None
    for x in xs:
        acc = orc.data_cumu("AVG", acc, 0, 33, x, 0, 33)
    n = len(xs) - 1
    w = np.array([2.0 ** -n] + [2.0 ** -(n - k + 1) for k in range(1, n + 1)])
    s = np.zeros(33)
    for k in range(len(xs)):
        s += w[k] * xs[k]
    assert np.array_equal(acc, s)


def test_u8_roundtrip_convention():
    x = orc.synth_iq(4096, 3) * 0.9
    b = orc.quantize_u8(x)
    y = orc.unpack_u8(b)
    assert np.max(np.abs(y - x)) <= (0.5 / 127.5) * np.sqrt(2) + 1e-12
    assert orc.unpack_u8(np.array([0, 255], dtype=np.uint8))[0] == complex(-1.0, 1.0)


# ---------------------------------------------------------------------------------------------------------
# round 2: reference-run vectors for the full-size scans, the dummy band, Save/AdjSigLvls and plot_highs
# (tests/golden/make_golden_r2.py)
def _regen_iq(g, count):
    x = orc.synth_iq(count, int(g["seed"])).astype(np.complex64)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["iq_sha256"])
    return x


def _scan_state(g, **kw):
    return orc.ScanState(int(g["fft_size"]), float(g["start_freq"]), float(g["end_freq"]), float(g["sampling_rate"]),
                         float(g["gain"]), float(g["min_amp"]), int(g["xres"]), float(g["scan_non_overlap"]), **kw)


@pytest.mark.parametrize("tag", ["fm_n16384", "quickfull_n64"])
def test_full_size_scan_matches_reference(tag):
    """BASELINE configs[2] (fmScan, 18 x 71 windows of 16384, kaiser) and configs[3]'s shape (quickFullScan,
    1226 steps of 71 x 64), two passes each: sampled bins + decimated checksums of all four curves, the
    waterfall rows of both passes."""
    g = golden("scan_" + tag)
    n, full, passes, steps = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"])
    assert steps == (18 if n == 16384 else 1226)
    x = _regen_iq(g, full * steps * passes).reshape(passes, steps, full)
    win = orc.window_table(str(g["window"]), n)
    st = _scan_state(g)
    assert st.total == int(g["total"])
    for p in range(passes):
        st.run_pass([orc.curscan(x[p, s], n, float(g["non_overlap"]), win, "AVG") for s in range(steps)])
    cells = int(g["cells"])
    for k in ("cur", "max", "min", "avg"):
        y = getattr(st, k)
        assert np.array_equal(y[g[k + "_idx"]], g[k + "_at_idx"]), k
        assert np.array_equal(y.reshape(cells, -1).sum(axis=1), g[k + "_decim_sum"]), k
        assert np.array_equal(y.reshape(cells, -1).max(axis=1), g[k + "_decim_max"]), k
    assert np.array_equal(st.hm[:passes + 1], g["hm_rows"])
    assert st.hm_index == int(g["hm_index"])


def test_scan_dummy_band_matches_reference():
    """A tune that fails (K:296-306) makes that band flat ones (K:637-639): steps 2 and 5 on passes 0 and 2."""
    g = golden("scan_dummy_n512")
    n, full, passes, steps = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"])
    ok = g["step_ok"]
    x = _regen_iq(g, full * int(ok.sum())).reshape(-1, full)
    win = orc.window_table(str(g["window"]), n)
    st = _scan_state(g)
    used = 0
    for p in range(passes):
        spectra = []
        for s in range(steps):
            if ok[p, s]:
                spectra.append(orc.curscan(x[used], n, float(g["non_overlap"]), win, "AVG"))
                used += 1
            else:
                spectra.append(None)
        st.run_pass(spectra)
    assert used == len(x)
    for k in ("cur", "max", "min", "avg", "hm"):
        assert np.array_equal(getattr(st, k), g[k]), k
    assert st.hm_index == int(g["hm_index"])


def test_adj_siglvls_zerospan_matches_reference():
    """AdjSigLvls (K:400-411): the curves stay raw, the waterfall row, the Levels curves and the markers see
    curve - Fft.Adj."""
    g = golden("adj_zerospan_n512")
    n, q, full, frames, xres = int(g["fft_size"]), float(g["non_overlap"]), int(g["full"]), int(g["frames"]), int(g["xres"])
    x = _regen_iq(g, full * frames).reshape(frames, full)
    adj = g["adj"]
    st = orc.ZeroSpanState(n, xres, float(g["gain"]), adj=adj)
    win = orc.window_table(str(g["window"]), n)
    for fr in x:
        st.push(orc.curscan(fr, n, q, win, "AVG"))
    for k in ("cur", "max", "min", "avg", "hm"):
        assert np.array_equal(getattr(st, k), g[k]), k
    freqs = np.fft.fftshift(np.fft.fftfreq(n, 1 / 2.4e6) + (float(g["start_freq"]) + float(g["end_freq"])) / 2)
    mode = str(g["compress"])
    mx, mn, av, cu = orc.adj_siglvls(st, adj)
    for name, y in (("lv_max", mx), ("lv_min", mn), ("lv_avg", av), ("lv_cur", cu)):
        xs, ys = orc.data_plotcompress(freqs, y, xres, mode)
        assert np.array_equal(xs, g["lv_x"]) and np.array_equal(ys, g[name]), name
    xs, ys = orc.data_plotcompress(freqs, cu, xres, mode)       # the last curve plotted is Cur (K:499-504)
    marks = orc.plot_highs(xs, ys, float(g["marker_delta"]), int(g["marker_count"]))
    assert np.array_equal(np.array(marks), g["markers"])


def test_adj_siglvls_scan_matches_reference():
    g = golden("adj_scan_n256")
    n, full, passes, steps, xres = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"]), int(g["xres"])
    x = _regen_iq(g, full * steps * passes).reshape(passes, steps, full)
    adj = g["adj"]
    st = _scan_state(g, adj=adj)
    win = orc.window_table(str(g["window"]), n)
    for p in range(passes):
        st.run_pass([orc.curscan(x[p, s], n, float(g["non_overlap"]), win, "AVG") for s in range(steps)])
    for k in ("cur", "max", "min", "avg", "hm"):
        assert np.array_equal(getattr(st, k), g[k]), k
    span = st.num_groups * float(g["sampling_rate"])
    freqs = np.fft.fftshift(np.fft.fftfreq(st.total, 1 / span) + float(g["start_freq"]) + span / 2)   # K:609
    mode = str(g["compress"])
    mx, mn, av, cu = orc.adj_siglvls(st, adj)
    for name, y in (("lv_max", mx), ("lv_min", mn), ("lv_avg", av), ("lv_cur", cu)):
        xs, ys = orc.data_plotcompress(freqs, y, xres, mode)
        assert np.array_equal(xs, g["lv_x"]) and np.array_equal(ys, g[name]), name
    xs, ys = orc.data_plotcompress(freqs, cu, xres, mode)
    marks = orc.plot_highs(xs, ys, float(g["marker_delta"]), int(g["marker_count"]))
    assert np.array_equal(np.array(marks), g["markers"])


def test_plot_highs_matches_reference():
    g = golden("plot_highs")
    for c, (delta, count) in enumerate(g["cases"]):
        marks = orc.plot_highs(g["freqs%d" % c], g["levels%d" % c], float(delta), int(count))
        assert np.array_equal(np.array(marks).reshape(-1, 2), g["marks%d" % c]), c
    # case 4: 16 points, delta 0, 20 markers asked -> 15 marked: the lowest point is never visited (K:258)
    assert len(g["marks4"]) == 15
