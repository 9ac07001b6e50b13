"""Seeded random sweep over the configuration space (sizes, fractional overlaps, windows, fold modes, sample
formats, batch lengths, xRes, non-standard fullSize): engine vs oracle through the batched device path."""
import os

import numpy as np
import pytest

import ksa_oracle as orc
from test_gpu_parity import assert_db, assert_lin

pytestmark = pytest.mark.gpu

SIZES = [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]
OVERLAPS = [0.07, 0.1, 0.25, 0.3333, 0.5, 0.75, 0.9, 1.0]
WINDOWS = ["ones", "hanning", "hamming", "kaiser"]
MODES = ["AVG", "MAX", "MIN", "RAW"]


# soak runs: KSA_RANDOM_CASES=600 KSA_RANDOM_SEED=7 python -m pytest tests/test_gpu_random.py -m gpu -q
COUNT = int(os.environ.get("KSA_RANDOM_CASES", "40"))
SEED = int(os.environ.get("KSA_RANDOM_SEED", "20201226"))
SOAK = COUNT > 40


def _cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(rng.choice(SIZES))
        q = float(rng.choice(OVERLAPS))
        mult = int(rng.choice([1, 2, 3, 8])) if n >= 8192 else int(rng.choice([1, 2, 5, 8, 11]))
        extra = int(rng.integers(0, n)) if rng.random() < 0.5 else 0       # ragged tail that must be dropped (K:389-390)
        full = n * mult + extra
        frames = int(rng.choice([1, 2, 3, 7, 33])) if n < 8192 else int(rng.choice([1, 2, 3]))
        if SOAK and n <= 2048 and mult <= 5 and rng.random() < 0.3:
            frames = int(rng.choice([777, 1555, 3100]))                    # more frames than resident workgroups
        xres = int(2 ** rng.integers(1, 10))
        out.append((i, n, q, str(rng.choice(WINDOWS)), str(rng.choice(MODES)), full, frames,
                    "u8" if rng.random() < 0.3 else "c64", min(xres, n)))
    return out


@pytest.mark.parametrize("case", _cases(COUNT, SEED), ids=lambda c: "r%d-N%d-q%s-%s-%s-%s" % (c[0], c[1], c[2], c[3], c[4], c[7]))
def test_random_configuration(ksa, case):
    import torch
    i, n, q, window, mode, full, frames, fmt, xres = case
    x = orc.synth_iq(full * frames, 5000 + i) * 0.6
    if fmt == "u8":
        raw = orc.quantize_u8(x).reshape(frames, 2 * full)
        xin = orc.unpack_u8(raw.reshape(-1)).reshape(frames, full)
        dev, code = torch.from_numpy(raw).cuda(), ksa.FMT_U8
    else:
        xin = x.astype(np.complex64).reshape(frames, full)
        dev, code = torch.view_as_real(torch.from_numpy(xin)).cuda(), ksa.FMT_C64
    win = orc.window_table(window, n)
    st_ref, db_ref, lin_ref = orc.zerospan_batch(xin, n, q, win, mode, 19.1, xres)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=window, cumu_mode=mode, xres=xres, max_frames=frames)
    assert eng.num_windows == len(orc.window_starts(full, n, q))
    lin = torch.empty((frames, n), dtype=torch.float32, device="cuda")
    eng.curscan_dev(dev, code, frames, lin)
    assert_lin(lin.cpu().numpy(), lin_ref, what="linear")
    rows_dev = torch.empty((frames, n), dtype=torch.float32, device="cuda")
    eng.frames_dev(dev, code, frames, cur_db=rows_dev)
    st = eng.state()
    for k in ("cur", "max", "min"):
        assert_db(st["Fft." + k.capitalize()], getattr(st_ref, k), what=k)
    # Row A10 on EVERY bin, ill-conditioned ones included: the oracle's accumulate (data_cumu, K:470-476) run over the
    # device's own per-frame dB rows must give the device's Cur / Max / Min exactly and its Avg to float32 rounding --
    # this separates the accumulate arithmetic from the fp32 transform noise that the comparison below has to allow for.
    rows = rows_dev.cpu().numpy().astype(np.float64)
    with np.errstate(invalid="ignore"):
        ema = rows[0].copy()
        for f in range(1, frames):
            ema = (ema + rows[f]) / 2                                  # K:137-139
    assert np.array_equal(st["Fft.Cur"], rows[-1]) and np.array_equal(st["Fft.Max"], rows.max(axis=0))
    assert np.array_equal(st["Fft.Min"], rows.min(axis=0))
    assert np.array_equal(np.isnan(st["Fft.Avg"]), np.isnan(ema)) and np.array_equal(np.isneginf(st["Fft.Avg"]), np.isneginf(ema))
    fin = np.isfinite(ema)
    assert np.max(np.abs(st["Fft.Avg"][fin] - ema[fin]), initial=0.0) <= 2e-4 * max(1.0, np.max(np.abs(ema[fin]), initial=0.0)), "avg vs EMA of the device rows"
    # Avg is an EMA of dB values (K:137-139): a frame in which a bin all but cancels (float64 ~1e-16, fp32 ~1e-8 of
    # the strongest bin) moves its average by tens of dB.  Such bins are compared only where every frame's value is
    # within 50 dB of the batch maximum, i.e. well above the fp32 noise floor.
    finite = np.isfinite(db_ref)
    well = np.all(finite & (db_ref > np.max(db_ref[finite]) - 50), axis=0) if finite.any() else np.zeros(n, bool)
    if well.all():
        assert_db(st["Fft.Avg"], st_ref.avg, what="avg")
    elif well.any():
        assert_db(st["Fft.Avg"][well], st_ref.avg[well], what="avg (well-conditioned bins)")
    # every bin, ill-conditioned ones included: the NaN pattern is the reference's, and an EMA of the frames' dB
    # values lies between their minimum and maximum (a -inf frame poisons Min and Avg alike, K:469)
    avg, mx, mn = st["Fft.Avg"], st["Fft.Max"], st["Fft.Min"]
    assert np.array_equal(np.isnan(avg), np.isnan(st_ref.avg)), "avg NaN pattern"
    ok = ~np.isnan(avg)
    assert not np.isposinf(avg[ok]).any()
    assert np.all(avg[ok] <= mx[ok] + 1e-3) and np.all(avg[ok] >= mn[ok] - 1e-3), "avg outside [min, max]"
    assert np.array_equal(np.isneginf(avg), np.isneginf(mn)), "-inf pattern of avg and min differ"
    rows = min(frames, 128)
    assert_db(st["fftHM"][:rows], st_ref.hm[:rows], what="waterfall")
    # the host-pointer drop-in agrees with the batched path
    assert_lin(eng.curscan(xin[0] if fmt == "c64" else raw[0]), lin_ref[0], what="host curscan")
    eng.close()


# ------------------------------------------------------------------------------------------------ scan stitch
def _scan_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(rng.choice([64, 128, 256, 512, 1024, 4096]))
        sq = float(rng.choice([0.5, 0.25, 0.75, 1.0, 0.125]))          # scanRangeNonOverlap with an integer hop
        fs = 2.4e6
        groups = int(rng.integers(1, 9 if n <= 1024 else 4))
        frac = float(rng.choice([0.0, 0.3, 0.77]))                     # span not a multiple of the sampling rate
        start = 88e6 + float(rng.integers(0, 10)) * 1e5
        end = start + (groups + frac) * fs
        passes = int(rng.choice([1, 2, 3, 5]))
        out.append((i, n, sq, start, end, fs, passes, str(rng.choice(WINDOWS)), bool(rng.random() < 0.3),
                    float(rng.choice([0.1, 0.5])), int(2 ** rng.integers(3, 10)), bool(rng.random() < 0.4)))
    return out


@pytest.mark.parametrize("case", _scan_cases(24 if not SOAK else COUNT // 4, SEED + 1),
                         ids=lambda c: "s%d-N%d-sq%s-p%d" % (c[0], c[1], c[2], c[6]))
def test_random_scan(ksa, case):
    """Scan passes (K:568-698: clip, dB, RAW/AVG stitch, Max/Min/Avg, waterfall row per pass) with random geometry,
    dummy bands (K:637-639) and bScanRangeBaseDataIsRaw (K:651-662) against the oracle's ScanState."""
    import torch
    i, n, sq, start, end, fs, passes, window, base_raw, q, xres, dummies = case
    end, _ = orc.fixup_scan_range(start, end, fs)          # K:701-709: the span becomes a whole number of bands
    steps = len(orc.scan_steps(start, end, fs, sq))
    if steps < 1:
        pytest.skip("empty scan range")
    full = 2 * n
    total = int((end - start) / fs) * n                    # K:599-600
    if total % xres:
        xres = n                                           # the reference's reshape (K:188-200) needs a divisor
    ref = orc.ScanState(n, start, end, fs, 19.1, 1e-7, xres, scan_non_overlap=sq, base_is_raw=base_raw)
    assert ref.total == total
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
        ref.run_pass([orc.curscan(x[s], n, q, win, "AVG") if ok[s] else None for s in range(steps)])
        eng.scan_pass_dev(torch.view_as_real(torch.from_numpy(x)).cuda(), ksa.FMT_C64, steps, step_ok=ok)
    st = eng.scan_state()
    assert st["passes"] == passes and st["hm_index"] == ref.hm_index
    # The four curves are dB-domain mixtures of the bands' bins (K:643-668): Min in particular may hold no strong bin at all, so
    # the linear comparison is normalised by the strongest value behind them -- the maximum of the oracle's Max curve -- the
    # north star's max|X| (a 500-case soak with seed 123 found 1.1e-5 of the Min curve's OWN maximum at a bin 30 dB below the
    # spectra's peak: 3e-8 of that peak, i.e. plain fp32 transform noise).
    top = float(np.max(10 ** (ref.max[np.isfinite(ref.max)] / 10)))
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], getattr(ref, k), what="scan " + k, top=top)
    assert_db(st["fftHM"][:min(passes, 128)], ref.hm[:min(passes, 128)], what="scan waterfall")
    eng.close()
