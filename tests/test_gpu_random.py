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
    eng.frames_dev(dev, code, frames)
    st = eng.state()
    for k in ("cur", "max", "min"):
        assert_db(st["Fft." + k.capitalize()], getattr(st_ref, k), what=k)
    # Avg is an EMA of dB values (K:137-139): a frame in which a bin all but cancels (float64 ~1e-16, fp32 ~1e-8 of
    # the strongest bin) moves its average by tens of dB.  Such bins are compared only where every frame's value is
    # within 50 dB of the batch maximum, i.e. well above the fp32 noise floor.
    finite = np.isfinite(db_ref)
    well = np.all(finite & (db_ref > np.max(db_ref[finite]) - 50), axis=0) if finite.any() else np.zeros(n, bool)
    if well.all():
        assert_db(st["Fft.Avg"], st_ref.avg, what="avg")
    elif well.any():
        assert_db(st["Fft.Avg"][well], st_ref.avg[well], what="avg (well-conditioned bins)")
    rows = min(frames, 128)
    assert_db(st["fftHM"][:rows], st_ref.hm[:rows], what="waterfall")
    # the host-pointer drop-in agrees with the batched path
    assert_lin(eng.curscan(xin[0] if fmt == "c64" else raw[0]), lin_ref[0], what="host curscan")
    eng.close()
