#!/usr/bin/env python3
"""kSpecAnal front end on the MI355X engine: the reference's zeroSpan / scan command line and plotting
hand-off (python/kspecanal.py, cited "K:") with the numpy.fft hot path replaced by libksa.

    python -m prgs-sdr-kspecanal_amd.kspecanal zeroSpan centerFreq 91.1e6 fftSize 4096 window hanning source synth
    python kspecanal.py fmScan source synth prgLoopCnt 4 bPltLevels false bPltHeatMap false

What is kept from the reference: the case-insensitive `KEY value` grammar and every key (K:813-911), the
mode words and aliases (K:816, K:912-921), the defaults (K:41-59), the dict `d` as the only state carrier
with the same key names, the module-level `sdr_curscan(d)` seam that playback rebinds (K:531, K:543), and the
arrays handed to matplotlib: d['Fft.Cur'|'Fft.Max'|'Fft.Min'|'Fft.Avg'], the freqs axis, and the
[128, W] waterfall buffer (K:470-481, K:729).  New keys are additive: `source` (rtlsdr|synth|file:<path>),
`device`, `iqFormat` (c64|u8).  What moved to the GPU: everything from the IQ block to those arrays.
Deliberate differences: playback needs no SDR; in scan mode the Levels plot is
refreshed once per pass (the whole pass is one device call) instead of once per tuned band.
Hand-off traffic: with a decimating pltCompress (AVG|MAX|MIN) a frame / pass brings back the four
xRes-point Levels curves, the peak markers and the ONE new waterfall row (ksa_read_view, d['handoff.bytes']); the
full-width d['Fft.*'] arrays are read from the device when a RAW / CONV plot needs them and once at the end of a run:
DURING such a run they are None / stale (the reference refreshes them after every frame, K:470-476) -- code that reads
d['Fft.*'] between frames calls _materialize(d, d['ksa.engine'], scan) first.
"""
import pickle
import signal
import sys
import time

import numpy as np

from . import engine as _engine
from .engine import SpectrumEngine, KsaError, FMT_C64, FMT_U8
from . import sources

PRGMODES = ("ZEROSPAN", "ZEROSPANSAVE", "ZEROSPANPLAY", "SCAN", "FMSCAN", "QUICKFULLSCAN")
PLTCOMPRESS = ("MAX", "MIN", "AVG", "RAW", "CONV")

# key -> (dict name, parser); the reference's spelling is matched case-insensitively (K:815)
def _arg_boolean(v):
    """K:771-775."""
    return v.upper() == "TRUE"


_bool = _arg_boolean
_KEYS = {
    "CENTERFREQ": ("centerFreq", float), "STARTFREQ": ("startFreq", float), "ENDFREQ": ("endFreq", float),
    "SAMPLINGRATE": ("samplingRate", float), "GAIN": ("gain", float), "MINAMP4CLIP": ("minAmp4Clip", float),
    "CURSCANNONOVERLAP": ("curScanNonOverlap", float), "CURSCANCUMUMODE": ("curScanCumuMode", str.upper),
    "SCANRANGENONOVERLAP": ("scanRangeNonOverlap", float), "FFTSIZE": ("fftSize", int), "XRES": ("xRes", int),
    "BDATAMIN": ("bDataMin", _bool), "BDATAMAX": ("bDataMax", _bool), "BDATAAVG": ("bDataAvg", _bool),
    "BDATACUR": ("bDataCur", _bool), "PLTCOMPRESS": ("pltCompress", str.upper),
    "WINDOW": ("window", lambda v: "WIN." + v.upper()), "BPLTHEATMAP": ("bPltHeatMap", _bool),
    "BPLTLEVELS": ("bPltLevels", _bool), "PRGLOOPCNT": ("prgLoopCnt", int),
    "PLTHIGHSNUMMARKERS": ("pltHighsNumMarkers", int), "PLTHIGHSDELTA4MARKING": ("pltHighsDelta4Marking", float),
    "PLTHIGHSPAUSE": ("pltHighsPause", _bool), "SAVESIGLVLS": ("SaveSigLvls", str), "ADJSIGLVLS": ("AdjSigLvls", str),
    "BGRID": ("bGrid", _bool), "BUSEPSD": ("bUsePSD", _bool),
    "BSCANRANGEBASEDATAISRAW": ("bScanRangeBaseDataIsRaw", _bool),
    "ZEROSPANSAVEFILE": ("zeroSpanSaveFile", str), "ZEROSPANPLAYFILE": ("zeroSpanPlayFile", str),
    # additive keys of this build
    "SOURCE": ("source", str), "DEVICE": ("device", int), "IQFORMAT": ("iqFormat", str.lower),
}


def defaults():
    """K:41-74 and K:783-812."""
    return {
        "prgMode": "FMSCAN", "samplingRate": 2.4e6, "gain": 19.1, "centerFreq": 92e6, "fftSize": 2 ** 14,
        "curScanNonOverlap": 0.1, "curScanCumuMode": "AVG", "window": "WIN.ONES",
        "minAmp4Clip": (1 / 256) * 0.00001, "bPltHeatMap": True, "bPltLevels": True,
        "scanRangeNonOverlap": 0.5, "prgLoopCnt": 8192, "xRes": 512, "pltCompress": "AVG",
        "pltHighsNumMarkers": 5, "pltHighsDelta4Marking": 0.025, "pltHighsPause": False, "pltCompressHM": "MAX",
        "SaveSigLvls": "", "AdjSigLvls": "", "bDataMin": True, "bDataMax": True, "bDataAvg": True, "bDataCur": True,
        "bGrid": True, "bUsePSD": False, "bScanRangeBaseDataIsRaw": False,
        "zeroSpanSaveFile": "/tmp/zerospan.save", "zeroSpanPlayFile": "/tmp/zerospan.save",
        "source": "rtlsdr", "device": 0, "iqFormat": "c64", "cmd.stop": False,
    }


def prg_quit(d, msg=None, tryExit=True):
    """K:967-972."""
    if msg is not None:
        print(msg)
    d["cmd.stop"] = True
    if tryExit:
        sys.exit()


def _fixupfreqs_scanrange(d):
    """K:701-709: stretch endFreq so that the range is a whole number of sampling-rate bands."""
    bands = (d["endFreq"] - d["startFreq"]) / d["samplingRate"]
    if (bands % 1) != 0:
        d["orig.EndFreq"] = d["endFreq"]
        d["endFreq"] = d["startFreq"] + np.ceil(bands) * d["samplingRate"]
        print("WARN:scanRange:Adjusting endFreq: orig [{}] adjusted [{}], so that fullRange is Multiple of samplingRate/freqBand [{}]".format(
            d["orig.EndFreq"], d["endFreq"], d["samplingRate"]))
    d["centerFreq"] = d["startFreq"] + ((d["endFreq"] - d["startFreq"]) / 2)


def handle_args(d, argv=None):
    """Fill `d` with the defaults and the user's KEY value pairs (K:778-949)."""
    for k, v in defaults().items():
        if not (k == "cmd.stop" and k in d):
            d[k] = v
    argv = list(sys.argv[1:] if argv is None else argv)
    i = 0
    while i < len(argv):
        cur = argv[i].upper()
        if cur in PRGMODES:
            d["prgMode"] = cur
        elif cur in _KEYS:
            if i + 1 >= len(argv):
                prg_quit(d, "ERROR:handle_args: Missing value for [{}]".format(cur))
            name, parse = _KEYS[cur]
            i += 1
            d[name] = parse(argv[i])
        else:
            prg_quit(d, "ERROR:handle_args: Unknown argument [{}]".format(cur))
        i += 1
    if d["prgMode"] == "FMSCAN":                                   # K:912-915
        d["prgMode"], d["startFreq"], d["endFreq"] = "SCAN", 88e6, 108e6
    elif d["prgMode"] == "QUICKFULLSCAN":                          # K:916-921
        d["prgMode"], d["startFreq"], d["endFreq"] = "SCAN", 30e6, 1.5e9
        d["fftSize"], d["pltCompress"] = 64, "RAW"
    if d["prgMode"] == "SCAN":
        _fixupfreqs_scanrange(d)
    else:
        d["startFreq"] = d["centerFreq"] - d["samplingRate"] / 2   # K:275-278
        d["endFreq"] = d["centerFreq"] + d["samplingRate"] / 2
    d["fullSize"] = _engine.full_size_for(d["fftSize"], d["samplingRate"])
    for name in ("HAMMING", "HANNING", "KAISER", "ONES"):         # K:932-935
        d["WIN." + name] = _engine.window_table(name, d["fftSize"])
    if d["window"] not in d:
        prg_quit(d, "ERROR:handle_args: Unknown window [{}]".format(d["window"]))
    d["theWin"] = d[d["window"]]
    if d["xRes"] > d["fftSize"]:                                   # K:938-940
        print("WARN:fftSize[{}] < xRes[{}], setting xRes to fftSize".format(d["fftSize"], d["xRes"]))
        d["xRes"] = d["fftSize"]
    elif d["fftSize"] % d["xRes"] != 0:                            # K:941-949
        for div in range(int(d["fftSize"] / 300), 0, -1):
            if d["fftSize"] % div == 0:
                new = d["fftSize"] // div
                print("WARN:fftSize[{}] NotMultipleOf xRes[{}], setting xRes to {}".format(d["fftSize"], d["xRes"], new))
                d["xRes"] = new
                break
    if d["curScanCumuMode"] not in ("AVG", "MAX", "MIN", "RAW"):
        prg_quit(d, "ERROR: Unknown cumuMode [{}], Quiting...".format(d["curScanCumuMode"]))
    return d


def print_info(d):
    """K:953-963."""
    print("INFO: startFreq[{}] centerFreq[{}] endFreq[{}]".format(d["startFreq"], d["centerFreq"], d["endFreq"]))
    print("INFO: samplingRate[{}], gain[{}], bUsePSD[{}]".format(d["samplingRate"], d["gain"], d["bUsePSD"]))
    print("INFO: fullSize[{}], fftSize[{}], curScanCumuMode[{}], window[{}]".format(
        d["fullSize"], d["fftSize"], d["curScanCumuMode"], d["window"]))
    print("INFO: minAmp4Clip[{}], curScanNonOverlap[{}], scanRangeNonOverlap[{}], bScanRangeBaseDataIsRaw[{}]".format(
        d["minAmp4Clip"], d["curScanNonOverlap"], d["scanRangeNonOverlap"], d["bScanRangeBaseDataIsRaw"]))
    print("INFO: prgMode [{}], prgLoopCnt[{}], bPltLevels[{}],  bPltHeatMap[{}]".format(
        d["prgMode"], d["prgLoopCnt"], d["bPltLevels"], d["bPltHeatMap"]))
    print("INFO: xRes [{}], bGrid [{}], pltCompress [{}], pltCompressHM [{}]".format(
        d["xRes"], d["bGrid"], d["pltCompress"], d["pltCompressHM"]))
    print("INFO: source [{}], device [{}], iqFormat [{}]".format(d["source"], d["device"], d["iqFormat"]))


# ------------------------------------------------------------------------------------------ SDR seam
def open_source(d):
    """d['source']: rtlsdr (needs pyrtlsdr) | synth | file:<raw uint8 capture>."""
    src = d["source"]
    if src == "synth":
        return sources.SyntheticSdr()
    if src.startswith("file:"):
        return sources.FileSdr(src[5:], d["samplingRate"], d["centerFreq"])
    try:
        import rtlsdr
    except ImportError:
        prg_quit(d, "ERROR: source rtlsdr needs the pyrtlsdr package; use `source synth` or `source file:<capture>`")
    return rtlsdr.RtlSdr()


_reopen_source = None     # set by main(): how a failed source is re-opened (the reference calls rtlsdr.RtlSdr() again, K:305)


def _calc_startendfreq(centerFreq, samplingRate):
    """K:275-278."""
    return centerFreq - samplingRate / 2, centerFreq + samplingRate / 2


def sdr_info(sdr):
    """K:281-284 (sources without these attributes print what they have)."""
    print("INFO:Sdr:SupportedGains:", getattr(sdr, "valid_gains_db", None))
    print("INFO:Sdr:Bandwidth:", getattr(sdr, "bandwidth", None))
    print("INFO:Sdr:freqCorrection:", getattr(sdr, "freq_correction", None))


def sdr_setup(sdr, fC, fS, gain):
    """K:287-308, same signature: tune, discard 16Ki settle samples; on any failure close and re-open the source and
    report False.  Returns (sdr, bOk)."""
    try:
        sdr.sample_rate = fS
        sdr.center_freq = fC
        sdr.gain = gain
        bOk = True
        sdr.read_samples(16 * 1024)
    except Exception:
        print("WARN:SetupSDR:FAILED: fC[{}] fS[{}] gain[{}]".format(fC, fS, gain))
        try:
            sdr.close()
        except Exception:
            pass
        if _reopen_source is not None:
            sdr = _reopen_source()
        bOk = False
    print("SetupSDR:{}: fC[{}] fS[{}] gain[{}]".format(bOk, fC, fS, gain))
    return sdr, bOk


SDR_READ_UNIT = 2 ** 18   # K:311


def sdr_read(sdr, length, raw=False):
    """K:312-347, same signature (+ raw): one capture block in <= 2^18-sample reads.  Returns complex64, or -- raw=True and a
    source that can deliver bytes -- uint8 I,Q pairs (the unpack then runs on the GPU)."""
    raw = raw and hasattr(sdr, "read_bytes")
    parts, left = [], int(length)
    while left > 0:
        n = min(left, SDR_READ_UNIT)
        if n < SDR_READ_UNIT:
            want = int(2 ** np.ceil(np.log2(n)))      # K:343: the dongle only reads power-of-two sizes
        else:
            want = n
        if raw:
            parts.append(np.asarray(sdr.read_bytes(2 * want), dtype=np.uint8)[:2 * n])
        else:
            parts.append(np.asarray(sdr.read_samples(want))[:n].astype(np.complex64))
        left -= n
    return np.concatenate(parts) if len(parts) > 1 else parts[0]


def get_engine(d, scan_total=0, max_frames=1):
    """One engine per (geometry, mode) -- rebuilt when a GUI toggle or argument changes the key."""
    key = (d["fftSize"], d["fullSize"], d["curScanNonOverlap"], d["curScanCumuMode"], d["window"], d["gain"],
           d["minAmp4Clip"], d["xRes"], scan_total, d["scanRangeNonOverlap"], max_frames, d["device"])
    if d.get("ksa.key") != key:
        if d.get("ksa.engine") is not None:
            d["ksa.engine"].close()
        d["ksa.engine"] = SpectrumEngine(
            d["fftSize"], full_size=d["fullSize"], non_overlap=d["curScanNonOverlap"], window=d["theWin"],
            cumu_mode=d["curScanCumuMode"], gain=d["gain"], min_amp=d["minAmp4Clip"], xres=d["xRes"],
            max_frames=max_frames, device=d["device"], scan_total_entries=scan_total,
            scan_non_overlap=d["scanRangeNonOverlap"])
        d["ksa.key"] = key
        if d.get("Fft.Adj") is not None:
            d["ksa.engine"].set_adj(d["Fft.Adj"], scan=bool(scan_total))
    return d["ksa.engine"]


def psd_crosscheck(d, samples, mag):
    """bUsePSD (K:350, K:374-384, README.rst:523-529) as a CPU-only DIAGNOSTIC next to the GPU result, never in
    place of it: matplotlib's Welch PSD of the same block (mlab.psd, the routine behind plt.psd at K:382: Fs = 2,
    two-sided, scale_by_freq, mean over segments) is converted back to this program's amplitude convention
    (|X| = sqrt(P * Fs * sum(w^2)), then 2*winAdj/N as K:391) and compared at the strongest bin.  The reference's own
    branch passes a float `noverlap` and raises TypeError under matplotlib >= 3.8 (observed with 3.10 in the build
    container), so there is no reference-run vector for it: parity unpinned, diagnostic only.  Stores d['psd.cur']
    (the PSD, fftshifted) and d['psd.check'] = (bin_gpu, bin_psd, level_gpu_dB, level_psd_dB)."""
    from matplotlib import mlab
    n, win = d["fftSize"], np.asarray(d["theWin"], dtype=np.float64)
    x = np.asarray(samples)
    if x.dtype == np.uint8:
        x = (x[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * ((x[1::2].astype(np.float64) - 127.5) / 127.5)
    pxx, _ = mlab.psd(x.astype(np.complex128), NFFT=n, window=win, noverlap=int(n * (1 - d["curScanNonOverlap"])))
    amp = np.sqrt(pxx * 2.0 * np.sum(win ** 2)) * 2.0 * (n / np.sum(win)) / n
    kg, kp = int(np.argmax(mag)), int(np.argmax(amp))
    with np.errstate(divide="ignore"):
        lg, lp = 10 * np.log10(mag[kg]) - d["gain"], 10 * np.log10(amp[kp]) - d["gain"]
    d["psd.cur"], d["psd.check"] = pxx, (kg, kp, float(lg), float(lp))
    print("DBUG:bUsePSD: peak bin gpu[{}] psd[{}] level gpu[{:.3f}] psd[{:.3f}] dB".format(kg, kp, lg, lp))


def sdr_curscan(d):
    """Drop-in for K:351-397: float64[fftSize] linear magnitudes, fftshifted."""
    samples = sdr_read(d["sdr"], d["fullSize"], raw=d.get("iqFormat") == "u8")
    mag = get_engine(d).curscan(samples)
    if d["bUsePSD"]:
        psd_crosscheck(d, samples, mag)
    return mag


_gpu_curscan = sdr_curscan   # zero_span fuses curscan + accumulate on the device while this is still bound


# ---------------------------------------------------------------------------------------- persistence
class _NumpyOnlyUnpickler(pickle.Unpickler):
    """The reference's save files are pickle streams of floats and float64 ndarrays (K:511-525, K:740-742).
    Only those reconstructors are admitted, so a crafted file cannot execute code."""
    _ok = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
           ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
           ("numpy._core.multiarray", "scalar"), ("numpy", "float64"),
           ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self._ok:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("refusing %s.%s in a kspecanal save file" % (module, name))


def _load(f):
    return _NumpyOnlyUnpickler(f).load()


def _save_siglvls(d):
    """K:736-748."""
    if d["SaveSigLvls"] == "":
        return
    try:
        with open(d["SaveSigLvls"], "wb+") as f:
            pickle.dump(d["startFreq"], f)
            pickle.dump(d["endFreq"], f)
            pickle.dump(np.asarray(d["Fft.Avg"], dtype=np.float64), f)
        print("INFO:_save_siglvls: success...", d["SaveSigLvls"])
    except Exception:
        print("WARN:_save_siglvls: Failed...", d["SaveSigLvls"])


def _load_siglvls(d):
    """K:751-768: the saved range must match exactly, otherwise the adjustment is dropped."""
    d["Fft.Adj"] = None
    if d["AdjSigLvls"] == "":
        return
    try:
        with open(d["AdjSigLvls"], "rb") as f:
            start, end, adj = _load(f), _load(f), _load(f)
        if start == d["startFreq"] and end == d["endFreq"]:
            d["Fft.Adj"] = np.asarray(adj, dtype=np.float64)
            print("INFO:_load_siglvls: success...", d["AdjSigLvls"])
        else:
            print("ERRR:_load_siglvls:{}:savedRange[{}-{}] curFreqRange[{}-{}]".format(
                d["AdjSigLvls"], start, end, d["startFreq"], d["endFreq"]))
            d["AdjSigLvls"] = ""
    except Exception:
        print("WARN:_load_siglvls: Failed...", d["AdjSigLvls"])
        d["AdjSigLvls"] = ""


def _adj_siglvls(d, cur):
    """K:400-411: the curves are stored raw and adjusted on the way to the plot."""
    adj = d.get("Fft.Adj")
    if d["AdjSigLvls"] != "" and adj is not None:
        sub = lambda v: None if v is None else v - adj                      # (a curve switched off is still None, K:471-476)
        return sub(d["Fft.Max"]), sub(d["Fft.Min"]), sub(d["Fft.Avg"]), sub(cur)
    return d["Fft.Max"], d["Fft.Min"], d["Fft.Avg"], cur


# ------------------------------------------------------------------------------------------- plotting
def plt_figures(d):
    """K:1077-1115 reduced to what the hand-off needs: a Levels axes and a Heatmap axes.  Skipped entirely
    when both plots are off (headless runs, the GPU box)."""
    d["plt"] = None
    if not (d["bPltLevels"] or d["bPltHeatMap"]):
        return
    import matplotlib.pyplot as plt
    plt.ion()
    fig = plt.figure("kSpecAnal", figsize=(12, 8), constrained_layout=True)
    gs = fig.add_gridspec(nrows=16, ncols=5)
    d["AxLevels"] = fig.add_subplot(gs[:8, :4])
    d["AxFreqs"] = fig.add_subplot(gs[:8, 4])
    d["AxHeatMap"] = fig.add_subplot(gs[8:16, :4])
    d["AxFreqs"].set_xticks([])
    d["AxFreqs"].set_yticks([])
    d["plt"] = plt


def _plotcompress(d, data, mode):
    """K:168-202 for the plot side (host arrays are tiny here: <= totalEntries floats)."""
    if mode == "RAW" or len(data) // d["xRes"] == 0:
        return data
    if mode == "CONV":
        conv = np.kaiser(128, 64)                                  # K:87
        out = np.convolve(data, conv, mode="same")
        avg = np.average(out)
        out[:12] = avg
        out[-12:] = avg
        return out
    t = np.asarray(data).reshape(d["xRes"], len(data) // d["xRes"])
    if mode == "MAX":
        return t.max(axis=1)
    if mode == "MIN":        # unreachable in the reference (K:188 vs K:196); implemented as documented there
        return t.min(axis=1)
    return np.average(t, axis=1)


_data_plotcompress = _plotcompress     # the reference's name (K:168-202)


def data_plotcompress(d, x, y, mode=None):
    """K:205-221."""
    mode = d["pltCompress"] if mode is None else mode
    if mode == "RAW":
        return x, y
    if mode == "CONV":
        return x, _plotcompress(d, y, mode)
    return _plotcompress(d, x, "AVG"), _plotcompress(d, y, mode)


def data_2d_plotcompress(d, data, mode=None):
    """K:224-237: every row of a 2-D set reduced like _data_plotcompress (the reference uses it once, to build the scan's
    initial waterfall buffer at K:614; the engine's rings are born decimated, so this is the host-side name for callers
    that hold full-width rows)."""
    mode = d["pltCompressHM"] if mode is None else mode
    if mode == "RAW":
        return data
    return np.array([_plotcompress(d, np.asarray(data)[r, :], mode) for r in range(np.asarray(data).shape[0])])


def plot_highs(d, freqs, levels, eng=None, curve=None, scan=False):
    """K:243-272: the strongest points of the last plotted curve, at least pltHighsDelta4Marking of the span apart.
    With an engine the selection runs on the device over the same decimated curve (ksa_read_highs) and only the
    marked cells come back; otherwise (RAW / CONV curves that are on the host anyway) the same walk runs here:
    descending argsort, the lowest point never visited (K:258).  Returns the list of (freq, level)."""
    delta = d["pltHighsDelta4Marking"] * (freqs[-1] - freqs[0])
    count = d["pltHighsNumMarkers"]
    marked = []
    if eng is not None and curve is not None and 1 <= count <= 64 and len(freqs) > 1:
        cell = (freqs[-1] - freqs[0]) / (len(freqs) - 1)
        idx, lvl = eng.highs(len(freqs), d["pltCompress"], curve, min_sep=delta / cell, count=count, scan=scan)
        marked = [(float(freqs[i]), float(v)) for i, v in zip(idx, lvl)]
    else:
        order = np.argsort(levels)
        for j in range(1, len(freqs)):
            i = order[-j]
            if all(abs(freqs[i] - f) >= delta for f, _ in marked):
                marked.append((float(freqs[i]), float(levels[i])))
                if len(marked) >= count:
                    break
    _show_highs(d, marked)
    return marked


def _view_on_device(d, n):
    """The Levels curves can be decimated on the device (pltCompress AVG|MAX|MIN over whole groups, K:186-201): then
    only xRes-sized arrays have to cross PCIe per frame / pass."""
    return d["pltCompress"] in ("AVG", "MAX", "MIN") and n // d["xRes"] > 0 and n % d["xRes"] == 0


def _materialize(d, eng, scan=False):
    """The FULL-width hand-off arrays d['Fft.Cur'|'Fft.Max'|'Fft.Min'|'Fft.Avg'] and the whole [128, W] waterfall
    buffer, read from the device: on demand (RAW / CONV Levels plots, K:205-221), and once when a run ends
    (SaveSigLvls K:736-748, callers of main())."""
    st = eng.scan_state() if scan else eng.state()
    for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg"):
        d[k] = st[k]
    if not scan:
        for flag, k in (("bDataMax", "Fft.Max"), ("bDataMin", "Fft.Min"), ("bDataAvg", "Fft.Avg")):
            if not d[flag]:
                d[k] = None                                                 # still None in the reference, K:471-476
    d["fftHM"], d["fftHMIndex"] = st["fftHM"], st["hm_index"]
    d["handoff.bytes"] = sum(st[k].size for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg", "fftHM")) * 4
    return st


def _handoff(d, eng, freqs, scan=False):
    """What one frame (zeroSpan, K:477-504) or one pass (scan, K:669-697 + K:729) hands to the plots.  With a
    decimating pltCompress everything is reduced on the device and ONE call (ksa_read_view) brings back the four
    xRes-point curves, the peak markers of the last plotted curve and the one new waterfall row: 4*xRes + xRes
    floats instead of 4*N + 128*W.  RAW / CONV plots need the full curves and materialise them."""
    if not _view_on_device(d, len(freqs)):
        _materialize(d, eng, scan)
        _plot_heatmap(d, d["fftHM"])
        _plot_levels(d, freqs, d["Fft.Cur"])
        return
    last = None
    for flag in ("bDataMax", "bDataMin", "bDataAvg", "bDataCur"):        # plotting order of K:489-503: the last one is marked
        if d[flag]:
            last = flag[5:].lower()
    # decimated x axis, cached by VALUE of the axis (an id() of a freed array can come back for a different axis)
    key = (len(freqs), float(freqs[0]), float(freqs[-1]), d["xRes"])
    if d.get("Levels.key") != key:
        d["Levels.x"], d["Levels.key"] = _plotcompress(d, freqs, "AVG"), key
    xs = d["Levels.x"]
    count = d["pltHighsNumMarkers"]
    cell = (xs[-1] - xs[0]) / (len(xs) - 1) if len(xs) > 1 else 1.0
    delta = d["pltHighsDelta4Marking"] * (xs[-1] - xs[0])
    want_marks = last is not None and 1 <= count <= 64 and len(xs) > 1
    lv, idx, lvl, rows, hm_index = eng.view(d["xRes"], d["pltCompress"], curve=last if want_marks else None,
                                            min_sep=delta / cell, count=count, hm_rows=1, scan=scan)
    d["handoff.bytes"] = (lv.size + rows.size + 2 * len(idx)) * 4
    d["Levels"] = {"x": xs, "max": lv[1], "min": lv[2], "avg": lv[3], "cur": lv[0]}
    d["fftHM"][(hm_index - 1) % _engine.HM_ROWS] = rows[0]               # the host copy of the ring takes the new row (K:480 / K:697)
    d["fftHMIndex"] = hm_index
    _plot_heatmap(d, d["fftHM"])
    drawing = d.get("plt") is not None and d["bPltLevels"]
    if drawing:
        d["AxLevels"].cla()
        if d["bGrid"]:
            d["AxLevels"].grid(True)
        for flag, row, colour in (("bDataMax", 1, "r"), ("bDataMin", 2, "y"), ("bDataAvg", 3, "g"), ("bDataCur", 0, "b")):
            if d[flag]:
                d["AxLevels"].plot(xs, lv[row], colour)
    if last is not None:
        if want_marks:
            _show_highs(d, [(float(xs[i]), float(v)) for i, v in zip(idx, lvl)])
        else:     # more markers than the device kernel returns (64), or a one-point curve: the host walk over the decimated curve
            plot_highs(d, xs, lv[{"cur": 0, "max": 1, "min": 2, "avg": 3}[last]])


def _show_highs(d, marked):
    """The drawing half of plot_highs (K:263-267)."""
    d["Highs"] = marked
    if d.get("plt") is not None and d["bPltLevels"]:
        d["AxFreqs"].clear()
        d["AxFreqs"].set_xticks([])
        d["AxFreqs"].set_yticks([])
        for n, (f, lvl) in enumerate(marked):
            d["AxLevels"].plot(f, lvl, "o", label=f)
            d["AxFreqs"].text(0.1, 1.0 - 0.1 * (n + 1), "{}:{}".format(round(f / 1e6, 8), round(lvl, 2)))


def _plot_levels(d, freqs, cur):
    """Levels hand-off from FULL host arrays (K:485-504 / K:670-688): the RAW / CONV modes, which plot every bin."""
    fmax, fmin, favg, fcur = _adj_siglvls(d, cur)
    x = y = None
    curves = (("bDataMax", fmax, "r"), ("bDataMin", fmin, "y"), ("bDataAvg", favg, "g"), ("bDataCur", fcur, "b"))
    if d.get("plt") is not None and d["bPltLevels"]:
        d["AxLevels"].cla()
        if d["bGrid"]:
            d["AxLevels"].grid(True)
    for flag, data, colour in curves:
        if d[flag] and data is not None:
            x, y = data_plotcompress(d, freqs, data)
            if d.get("plt") is not None and d["bPltLevels"]:
                d["AxLevels"].plot(x, y, colour)
    if x is not None:
        plot_highs(d, x, y)


def _plot_heatmap(d, hm):
    if d.get("plt") is None or not d["bPltHeatMap"]:
        return
    if d.get("hm.artist") is None:
        d["hm.artist"] = d["AxHeatMap"].imshow(hm, extent=(0, 1, 0, 1), aspect="auto", interpolation="bicubic")
        d["AxHeatMap"].set_xticks([0, 0.25, 0.5, 0.75, 1])
        d["AxHeatMap"].set_xticklabels([d["startFreq"], (d["startFreq"] + d["centerFreq"]) / 2, d["centerFreq"],
                                        (d["centerFreq"] + d["endFreq"]) / 2, d["endFreq"]])
    d["hm.artist"].set_data(hm)                                    # K:481 / K:729
    d["hm.artist"].autoscale()
    d["plt"].pause(0.0001)


# ------------------------------------------------------------------------------------------ zeroSpan
def zero_span(d):
    """K:426-505.  Per frame: capture, curscan + LogNoGain + Max/Min/Avg/Cur + waterfall row on the GPU,
    then the hand-off arrays are refreshed for the plots."""
    for k in ("Fft.Max", "Fft.Min", "Fft.Avg", "Fft.Cur"):
        d[k] = None
    d["timeWasStr"] = None
    if d.get("sdr") is not None:
        d["sdr"], _ = sdr_setup(d["sdr"], d["centerFreq"], d["samplingRate"], d["gain"])
    freqs = np.fft.fftshift(np.fft.fftfreq(d["fftSize"], 1 / d["samplingRate"]) + d["centerFreq"])   # K:444-445
    d["freqs"] = freqs
    print("ZeroSpan: min[{}] max[{}]".format(min(freqs), max(freqs)))
    eng = get_engine(d)
    eng.reset()
    d["fftHM"], d["fftHMIndex"] = np.zeros((_engine.HM_ROWS, eng.hm_width)), 0     # K:456; the device ring starts the same
    prev = time.time()
    for i in range(d["prgLoopCnt"]):
        now = time.time()
        print("ZeroSpan:{}:{}".format(i, now - prev))
        prev = now
        eng.set_flags(d["bDataMax"], d["bDataMin"], d["bDataAvg"])          # GUI toggles K:471-476
        if sdr_curscan is _gpu_curscan and not d["bUsePSD"]:
            try:
                eng.frame(sdr_read(d["sdr"], d["fullSize"], raw=d.get("iqFormat") == "u8"))   # fused K:464-484
            except EOFError:
                prg_quit(d, "WARN:zero_span: source exhausted, stoping...", False)
        else:
            try:
                cur = sdr_curscan(d)                                        # rebound seam (playback, K:543) / bUsePSD
            except EOFError:
                cur = None
                prg_quit(d, "WARN:zero_span: source exhausted, stoping...", False)
            if cur is not None:
                eng.frame_spectrum(cur)
        if d["cmd.stop"]:
            break
        _handoff(d, eng, freqs)                  # xRes-sized curves + markers + the new waterfall row (row f2)
    if _materialize(d, eng)["frames"] == 0:      # full-width arrays once, for SaveSigLvls and whoever called main()
        for k in ("Fft.Max", "Fft.Min", "Fft.Avg", "Fft.Cur"):
            d[k] = None                          # no frame ran: the curves are still None (K:427-430)


def zero_span_save(d):
    """K:510-526: header then (time, linear spectrum) records; same stream the reference writes."""
    with open(d["zeroSpanSaveFile"], "wb+") as f:
        for k in ("centerFreq", "samplingRate", "gain"):
            pickle.dump(d[k], f)
        d["sdr"], _ = sdr_setup(d["sdr"], d["centerFreq"], d["samplingRate"], d["gain"])
        prev = time.time()
        for i in range(d["prgLoopCnt"]):
            if d["cmd.stop"]:
                break
            now = time.time()
            print("ZeroSpanSave:{}:{}".format(i, now - prev))
            prev = now
            try:
                cur = sdr_curscan(d)
            except EOFError:
                break
            pickle.dump(now, f)
            pickle.dump(cur, f)


def zero_span_play_setup(d):
    """K:530-543: the file's header overrides centre / rate / gain and `sdr_curscan` is rebound."""
    global sdr_curscan
    d["zeroSpanFile"] = f = open(d["zeroSpanPlayFile"], "rb")
    d["centerFreq"], d["samplingRate"], d["gain"] = _load(f), _load(f), _load(f)
    d["startFreq"], d["endFreq"] = _calc_startendfreq(d["centerFreq"], d["samplingRate"])
    sdr_curscan = zero_span_play


def zero_span_play(d):
    """K:547-564: next saved spectrum, or None (and cmd.stop) at the end of the file."""
    try:
        d["timeWas"] = _load(d["zeroSpanFile"])
        ms = int((d["timeWas"] - int(d["timeWas"])) * 1000)
        d["timeWasStr"] = "{}.{:03}".format(time.strftime("%Y%m%d%Z%H%M%S", time.gmtime(d["timeWas"])), ms)
        print("INFO:zeroSpanPlay:timeWas:{}".format(d["timeWasStr"]))
        return _load(d["zeroSpanFile"])
    except Exception:
        prg_quit(d, "WARN:zero_span_play:loading failed, stoping...", False)
        return None


# ------------------------------------------------------------------------------------------------ scan
def scan_geometry(d):
    """K:587-600, K:621, K:689-690: (numGroups, totalEntries, tuned centre of every step)."""
    span = d["samplingRate"]
    q = d["scanRangeNonOverlap"]
    if ((span * q) % 1) != 0:
        prg_quit(d, "ERROR: freqSpan [{}] x scanRangeNonOverlap [{}] is not int".format(span, q))
    if ((d["fftSize"] * q) % 1) != 0:
        prg_quit(d, "ERROR: fftSize[{}] x scanRangeNonOverlap [{}] is not int".format(d["fftSize"], q))
    groups = int((d["endFreq"] - d["startFreq"]) / span)
    centers, cur = [], d["startFreq"] + span / 2
    while cur - span / 2 < d["endFreq"]:
        centers.append(cur)
        cur += span * q
    return groups, groups * d["fftSize"], centers


def scan_range(d):
    """K:712-732 with _scan_range (K:568-698): per pass, capture every tuned band, then one device call
    stitches the bands and updates Cur/Max/Min/Avg and the waterfall row."""
    _fixupfreqs_scanrange(d)
    groups, total, centers = scan_geometry(d)
    print("_scanRange: start:{} end:{} samplingRate:{}".format(d["startFreq"], d["endFreq"], d["samplingRate"]))
    print("_scanRange: totalFreqs:{} numGroups:{} totalEntries:{}".format(d["endFreq"] - d["startFreq"], groups, total))
    steps = len(centers)
    if d["bUsePSD"]:
        print("WARN:_scanRange: bUsePSD is a zeroSpan diagnostic here (the whole pass is one device call); ignored")
    eng = get_engine(d, scan_total=total, max_frames=steps)
    eng.scan_reset()
    eng.scan_set_base_is_raw(d["bScanRangeBaseDataIsRaw"])
    span = groups * d["samplingRate"]
    d["freqsAll"] = np.fft.fftshift(np.fft.fftfreq(total, 1 / span) + d["startFreq"] + span / 2)   # K:609
    u8 = d["iqFormat"] == "u8" and hasattr(d["sdr"], "read_bytes")
    # one pass of capture blocks in page-locked host memory; the library stages it to the GPU (ksa_scan_pass_c64 / _u8)
    stage = _engine.PinnedBuffer((steps, d["fullSize"] * 2) if u8 else (steps, d["fullSize"]), np.uint8 if u8 else np.complex64)
    blocks = stage.array
    d["fftHM"], d["fftHMIndex"] = eng.hm_rows(0, _engine.HM_ROWS, scan=True), 0    # K:613-614, read once
    prev = time.time()
    for i in range(d["prgLoopCnt"]):
        if d["cmd.stop"]:
            break
        now = time.time()
        print("scanRange:{}:{}".format(i, now - prev))
        prev = now
        ok = np.ones(steps, dtype=np.uint8)
        for s, fc in enumerate(centers):
            d["sdr"], bOk = sdr_setup(d["sdr"], fc, d["samplingRate"], d["gain"])
            if not bOk:
                print("WARN:_scanRange: Dummy data for {} to {}".format(fc - d["samplingRate"] / 2, fc + d["samplingRate"] / 2))
                ok[s] = 0                                                   # K:637-639
                continue
            blocks[s] = sdr_read(d["sdr"], d["fullSize"], raw=u8)
        eng.set_flags(d["bDataMax"], d["bDataMin"], True)
        eng.scan_pass(blocks, step_ok=ok)
        _handoff(d, eng, d["freqsAll"], scan=True)
    _materialize(d, eng, scan=True)
    stage.close()


# ------------------------------------------------------------------------------------------------ main
def do_run(d):
    """K:1126-1136."""
    if d["prgMode"] == "SCAN":
        scan_range(d)
    elif d["prgMode"] == "ZEROSPANSAVE":
        zero_span_save(d)
    elif d["prgMode"] == "ZEROSPANPLAY":
        zero_span_play_setup(d)
        zero_span(d)
        d["zeroSpanFile"].close()
    else:
        zero_span(d)


gD = None     # the reference's module-level state dict (K:1139); handle_sigint needs it


def handle_sigint(signum, stack):
    """K:1118-1119."""
    prg_quit(gD, "INFO:sigint: quiting on user request...")


def handle_signals(d):
    """K:1122-1123."""
    signal.signal(signal.SIGINT, handle_sigint)


def main(argv=None):
    global sdr_curscan, gD, _reopen_source
    sdr_curscan = _gpu_curscan
    d = gD = {"cmd.stop": False}
    handle_args(d, argv)
    _load_siglvls(d)
    print_info(d)
    handle_signals(d)
    _reopen_source = lambda: open_source(d)
    plt_figures(d)
    d["sdr"] = None if d["prgMode"] == "ZEROSPANPLAY" else open_source(d)   # playback needs no SDR (appendix B)
    if d["sdr"] is not None:
        sdr_info(d["sdr"])                                                   # K:1147
    try:
        do_run(d)
    finally:
        if d.get("sdr") is not None:
            d["sdr"].close()
        if d.get("ksa.engine") is not None:
            d["ksa.engine"].close()
            d["ksa.engine"] = None
    _save_siglvls(d)
    return d


if __name__ == "__main__":
    main()
