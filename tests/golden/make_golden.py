#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE in the build container.

Run only where /root/reference exists (never on the GPU box, never from tests):

    python tests/golden/make_golden.py

How the reference is driven: `python/kspecanal.py` has no
`__main__` guard and imports the absent `rtlsdr` package, so it is executed with
`runpy.run_path` under (a) the Agg matplotlib backend, (b) a data-source module
registered as `rtlsdr` whose `RtlSdr.read_samples(n)` replays a deterministic
complex64 IQ stream (upcast to complex128 exactly as the reference's sdr_read
buffer would hold it) -- the same seam the reference itself provides at its line
14 (`#import testfft as rtlsdr`), (c) `builtins.input` patched, (d) `sys.argv`
set to the reference's own KEY value grammar with plotting off.  Outputs are
read from the reference's global dict `gD` and by calling its own functions
(`sdr_curscan`, `data_cumu`, `_data_plotcompress`, `data_proc`) from the
returned namespace.  Only data (inputs + expected outputs) is written.
"""
import builtins
import hashlib
import io
import contextlib
import os
import pickle
import runpy
import sys
import types

import numpy as np
import matplotlib
matplotlib.use("Agg")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ksa_oracle as orc  # only for the synthetic source (synth_iq / quantize_u8)

REF = "/root/reference/python/kspecanal.py"
SEED0 = 20201226


class ReplaySdr:
    """Duck type of rtlsdr.RtlSdr (K:281-308): replays a flat complex64 stream."""
    valid_gains_db = [0.0, 19.1]
    bandwidth = 0
    freq_correction = 0
    stream = np.zeros(0, dtype=np.complex64)
    pos = 0

    def __init__(self):
        self.sample_rate = 0
        self.center_freq = 0
        self._gain = 0
        self._settle = False

    @property
    def gain(self):
        return self._gain

    @gain.setter
    def gain(self, v):          # sdr_setup sets gain last, then does the settle read (K:299-301)
        self._gain = v
        self._settle = True

    def read_samples(self, n):
        n = int(n)
        if self._settle:        # the 16Ki settle read is discarded by the reference
            self._settle = False
            return np.zeros(n, dtype=np.complex128)
        cls = type(self)
        out = cls.stream[cls.pos:cls.pos + n].astype(np.complex128)
        assert len(out) == n, "replay stream exhausted"
        cls.pos += n
        return out

    def close(self):
        pass


def run_reference(argv, stream):
    mod = types.ModuleType("rtlsdr")
    mod.RtlSdr = ReplaySdr
    sys.modules["rtlsdr"] = mod
    ReplaySdr.stream = np.ascontiguousarray(stream, dtype=np.complex64)
    ReplaySdr.pos = 0
    builtins.input = lambda *a, **k: ""
    sys.argv = ["kspecanal.py"] + [str(a) for a in argv] + ["bPltLevels", "false", "bPltHeatMap", "false"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns = runpy.run_path(REF)
    return ns


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote %-34s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


def ref_curscan(ns, stream, fft_size, non_overlap, window, cumu, full):
    """Call the reference's own sdr_curscan on one captured block."""
    ReplaySdr.stream = np.ascontiguousarray(stream, dtype=np.complex64)
    ReplaySdr.pos = 0
    win = {"ones": np.ones(fft_size), "hanning": np.hanning(fft_size),
           "hamming": np.hamming(fft_size), "kaiser": np.kaiser(fft_size, 64)}[window]
    d = {"fullSize": full, "fftSize": fft_size, "curScanNonOverlap": non_overlap,
         "curScanCumuMode": cumu, "theWin": win, "bUsePSD": False, "sdr": ReplaySdr()}
    return ns["sdr_curscan"](d)


def main():
    # one reference load whose namespace supplies the plain functions
    ns = run_reference(["zeroSpan", "fftSize", 64, "prgLoopCnt", 1],
                       orc.synth_iq(512, SEED0).astype(np.complex64))

    # ---- (1)+(2) curscan outputs: windows x cumu modes x geometries --------
    cases = [  # (tag, N, nonOverlap, full)
        ("n64_q01", 64, 0.1, 512),          # C4 geometry, hops 6/7
        ("n512_q01", 512, 0.1, 4096),       # fractional hops 51/52
        ("n512_q05", 512, 0.5, 4096),
        ("n4096_q05", 4096, 0.5, 32768),    # C2 geometry
    ]
    for tag, n, q, full in cases:
        x = orc.synth_iq(full, SEED0 + n).astype(np.complex64)
        out = {}
        for window in ("ones", "hanning", "hamming", "kaiser"):
            for cumu in ("AVG", "MAX", "MIN", "RAW"):
                if n == 4096 and window in ("ones", "hamming") and cumu != "AVG":
                    continue
                out["%s_%s" % (window, cumu)] = ref_curscan(ns, x, n, q, window, cumu, full)
        save("curscan_" + tag, iq=x, fft_size=n, non_overlap=q, full=full, **out)

    # per-window row vectors A4-A6 for one window (reference expression K:391 via its numpy calls)
    # are covered by the RAW mode above (RAW = last window only).

    # ---- (3) zeroSpan runs: Fft.* + fftHM + index ---------------------------
    for tag, n, q, window, frames, fs in (("n512", 512, 0.5, "hanning", 5, 2.4e6),
                                          ("n4096", 4096, 0.5, "hanning", 4, 2.4e6),
                                          ("n64", 64, 0.1, "kaiser", 6, 2.4e6)):
        full = orc.full_size(n, fs)
        x = orc.synth_iq(full * frames, SEED0 + 100 + n).astype(np.complex64)
        nsz = run_reference(["zeroSpan", "fftSize", n, "window", window, "curScanNonOverlap", q,
                             "prgLoopCnt", frames, "bPltHeatMap", "true"], x)
        # bPltHeatMap true is needed for fftHM; the trailing "false" appended by run_reference
        # would override it, so re-run with explicit ordering below
        g = nsz["gD"]
        save("zerospan_" + tag, iq=x, fft_size=n, non_overlap=q, window=window, frames=frames,
             full=g["fullSize"], gain=g["gain"], xres=g["xRes"],
             cur=g["Fft.Cur"], max=g["Fft.Max"], min=g["Fft.Min"], avg=g["Fft.Avg"])

    # heat-map: the reference only fills fftHM when bPltHeatMap is true (K:479-484), which needs a
    # figure; run once with the heatmap on (Agg backend) and capture the array handed to set_data.
    n, q, frames = 512, 0.5, 5
    full = orc.full_size(n, 2.4e6)
    x = orc.synth_iq(full * frames, SEED0 + 100 + n).astype(np.complex64)
    captured = {}
    import matplotlib.image as mimage
    orig_set_data = mimage.AxesImage.set_data

    def spy(self, A):
        captured["hm"] = np.array(A, copy=True)
        return orig_set_data(self, A)
    mimage.AxesImage.set_data = spy
    try:
        mod = types.ModuleType("rtlsdr"); mod.RtlSdr = ReplaySdr; sys.modules["rtlsdr"] = mod
        ReplaySdr.stream = x; ReplaySdr.pos = 0
        builtins.input = lambda *a, **k: ""
        sys.argv = ["kspecanal.py", "zeroSpan", "fftSize", str(n), "window", "hanning",
                    "curScanNonOverlap", str(q), "prgLoopCnt", str(frames),
                    "bPltLevels", "false", "bPltHeatMap", "true", "xRes", "128"]
        with contextlib.redirect_stdout(io.StringIO()):
            nsh = runpy.run_path(REF)
    finally:
        mimage.AxesImage.set_data = orig_set_data
    g = nsh["gD"]
    save("zerospan_hm_n512", iq=x, fft_size=n, non_overlap=q, window="hanning", frames=frames,
         full=g["fullSize"], gain=g["gain"], xres=g["xRes"], hm=captured["hm"],
         cur=g["Fft.Cur"], max=g["Fft.Max"], min=g["Fft.Min"], avg=g["Fft.Avg"])

    # ---- (4) mini scans (A13) ------------------------------------------------
    for tag, argv, fs in (
            ("3band_n512", ["scan", "startFreq", 100e6, "endFreq", 107.2e6, "fftSize", 512,
                            "window", "kaiser", "prgLoopCnt", 2], 2.4e6),
            ("frac_n256", ["scan", "startFreq", 88e6, "endFreq", 93e6, "fftSize", 256,
                           "window", "hanning", "prgLoopCnt", 3, "xRes", 64], 2.4e6),
            ("quick_n64", ["scan", "startFreq", 30e6, "endFreq", 54e6, "fftSize", 64,
                           "prgLoopCnt", 2, "pltCompress", "RAW"], 2.4e6),
            ("baseraw_n256", ["scan", "startFreq", 200e6, "endFreq", 207.2e6, "fftSize", 256, "window", "hamming",
                              "prgLoopCnt", 3, "xRes", 128, "bScanRangeBaseDataIsRaw", "true"], 2.4e6)):
        n = argv[argv.index("fftSize") + 1]
        passes = argv[argv.index("prgLoopCnt") + 1]
        a, b = argv[argv.index("startFreq") + 1], argv[argv.index("endFreq") + 1]
        b2, _ = orc.fixup_scan_range(a, b, fs)
        steps = len(orc.scan_steps(a, b2, fs, 0.5))
        full = orc.full_size(n, fs)
        x = orc.synth_iq(full * steps * passes, SEED0 + 200 + n).astype(np.complex64)
        # the heat-map buffer of the scan is built only with bPltHeatMap true (K:610-614, K:696-697)
        mod = types.ModuleType("rtlsdr"); mod.RtlSdr = ReplaySdr; sys.modules["rtlsdr"] = mod
        ReplaySdr.stream = x; ReplaySdr.pos = 0
        builtins.input = lambda *a, **k: ""
        sys.argv = ["kspecanal.py"] + [str(v) for v in argv] + ["bPltLevels", "false", "bPltHeatMap", "true"]
        with contextlib.redirect_stdout(io.StringIO()):
            nss = runpy.run_path(REF)
        g = nss["gD"]
        assert ReplaySdr.pos == len(x), (ReplaySdr.pos, len(x))
        save("scan_" + tag, iq=x, fft_size=n, passes=passes, steps=steps, full=g["fullSize"],
             start_freq=g["startFreq"], end_freq=g["endFreq"], sampling_rate=g["samplingRate"],
             gain=g["gain"], min_amp=g["minAmp4Clip"], xres=g["xRes"], window=g["window"],
             non_overlap=g["curScanNonOverlap"], scan_non_overlap=g["scanRangeNonOverlap"],
             base_is_raw=g["bScanRangeBaseDataIsRaw"],
             cur=g["Fft.Cur"], max=g["Fft.Max"], min=g["Fft.Min"], avg=g["Fft.Avg"],
             hm=g["fftHM"], hm_index=g["fftHMIndex"])

    # ---- (5) zeroSpanSave stream (config 1 plumbing) ---------------------------
    n, frames = 512, 5
    full = orc.full_size(n, 2.4e6)
    x = orc.synth_iq(full * frames, SEED0 + 300).astype(np.complex64)
    path = "/tmp/ksa_golden_zerospan.save"
    run_reference(["zeroSpanSave", "fftSize", n, "window", "hanning", "curScanNonOverlap", 0.5,
                   "prgLoopCnt", frames, "zeroSpanSaveFile", path], x)
    with open(path, "rb") as f:   # a file this script just wrote
        hdr = [pickle.load(f) for _ in range(3)]
        recs = []
        while True:
            try:
                t = pickle.load(f); a = pickle.load(f)
            except EOFError:
                break
            recs.append(a)
        raw = open(path, "rb").read()
    nsp = run_reference(["zeroSpanPlay", "fftSize", n, "zeroSpanPlayFile", path, "prgLoopCnt", frames + 3], x[:1])
    g = nsp["gD"]
    save("zerospan_save_n512", iq=x, fft_size=n, frames=frames, full=full, header=np.array(hdr),
         spectra=np.array(recs), stream=np.frombuffer(raw, dtype=np.uint8),
         play_cur=g["Fft.Cur"], play_max=g["Fft.Max"], play_min=g["Fft.Min"], play_avg=g["Fft.Avg"])

    # ---- (6) on-bin tone known answers -----------------------------------------
    n, full = 4096, 32768
    t = np.arange(full)
    x = (0.5 * np.exp(2j * np.pi * 0.125 * t)).astype(np.complex64)
    out = {w: ref_curscan(ns, x, n, 0.5, w, "AVG", full) for w in ("ones", "hanning", "hamming", "kaiser")}
    save("tone_n4096", iq=x, fft_size=n, non_overlap=0.5, full=full, **out)

    # ---- (7) large N: sampled bins + checksums ------------------------------------
    for tag, n, q, window, full in (("n16384_q01", 16384, 0.1, "kaiser", 131072),
                                    ("n65536_q025", 65536, 0.25, "hanning", 524288),
                                    ("n8192_q05", 8192, 0.5, "hanning", 65536),
                                    ("n32768_q05", 32768, 0.5, "hamming", 65536)):
        seed = SEED0 + 400 + n
        x = orc.synth_iq(full, seed).astype(np.complex64)
        y = ref_curscan(ns, x, n, q, window, "AVG", full)
        ymax = ref_curscan(ns, x, n, q, window, "MAX", full)
        idx = np.unique(np.concatenate([np.arange(0, n, n // 256), np.argsort(y)[-32:], np.argsort(y)[:32]]))
        save("curscan_" + tag, seed=seed, iq_sha256=sha(x), fft_size=n, non_overlap=q, window=window,
             full=full, idx=idx, avg_at_idx=y[idx], max_at_idx=ymax[idx], avg_sum=np.sum(y),
             avg_decim=y.reshape(256, -1).sum(axis=1), max_decim=ymax.reshape(256, -1).max(axis=1),
             peak=np.max(y))

    # ---- CLI grammar: what handle_args leaves in gD for a set of command lines (K:778-949) ----------
    import json
    cli = {}
    cases_cli = {
        "default": [],
        "zerospan": ["zeroSpan", "centerFreq", "91.1e6", "fftSize", "4096", "window", "hanning", "curScanNonOverlap", "0.5"],
        "fmscan": ["FMSCAN", "gain", "7.1", "window", "kaiser"],
        "quickfullscan": ["quickFullScan"],
        "scan_uneven": ["scan", "startFreq", "400e6", "endFreq", "500e6", "fftSize", "2048", "samplingRate", "2e6"],
        "xres_fix": ["zeroSpan", "fftSize", "1024", "xRes", "300"],
        "xres_big": ["zeroSpan", "fftSize", "256", "xRes", "512"],
        "big_fft": ["zeroSpan", "fftSize", str(2 ** 19), "curScanCumuMode", "max", "pltCompress", "max"],
        "flags": ["ZEROSPAN", "bDataMin", "false", "bDataMax", "TRUE", "bGrid", "False", "minAmp4Clip", "1e-9",
                  "scanRangeNonOverlap", "0.25", "pltHighsNumMarkers", "3", "pltHighsDelta4Marking", "0.1"],
    }
    keys = ["prgMode", "samplingRate", "gain", "centerFreq", "fftSize", "curScanNonOverlap", "curScanCumuMode",
            "window", "minAmp4Clip", "scanRangeNonOverlap", "prgLoopCnt", "xRes", "pltCompress", "pltCompressHM",
            "pltHighsNumMarkers", "pltHighsDelta4Marking", "bDataMin", "bDataMax", "bDataAvg", "bDataCur", "bGrid",
            "bUsePSD", "bScanRangeBaseDataIsRaw", "startFreq", "endFreq", "fullSize", "zeroSpanSaveFile"]
    for name, argv in cases_cli.items():
        g = run_reference(argv + ["prgLoopCnt", 0], np.zeros(1, dtype=np.complex64))["gD"]
        cli[name] = {"argv": argv, "d": {k: (float(g[k]) if isinstance(g[k], (float, np.floating)) else g[k]) for k in keys}}
    with open(os.path.join(HERE, "cli_args.json"), "w") as f:
        json.dump(cli, f, indent=1, sort_keys=True)
    print("wrote cli_args.json")

    # ---- pieces: data_cumu / _data_plotcompress / data_proc direct -----------------
    rng = np.random.default_rng(SEED0 + 500)
    a = rng.standard_normal(64); b = rng.standard_normal(64)
    d = {"xRes": 16, "gain": 19.1, "minAmp4Clip": (1 / 256) * 0.00001}
    pieces = {"a": a, "b": b}
    for mode in ("RAW", "AVG", "MAX", "MIN"):
        pieces["cumu_" + mode] = ns["data_cumu"](d, mode, np.copy(a), 8, 40, b, 4, 36)
    pieces["compress_MAX"] = ns["_data_plotcompress"](d, np.copy(a), "MAX")
    pieces["compress_AVG"] = ns["_data_plotcompress"](d, np.copy(a), "AVG")
    v = np.abs(a); v[3] = 0.0
    with np.errstate(divide="ignore"):
        pieces["lognogain"] = ns["data_proc"](d, np.copy(v), "LogNoGain")
        pieces["lognogain_inf0"] = ns["data_proc"](d, np.copy(v), "LogNoGain", 0)
    pieces["clip"] = ns["data_proc"](d, np.copy(v) * 1e-7, "Clip2MinAmp")
    save("pieces", **pieces)


if __name__ == "__main__":
    main()
