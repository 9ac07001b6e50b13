"""Host-side mirror of the reference's CLI / persistence / plot-side helpers (no GPU needed).
cli_args.json holds what the reference's own handle_args left in gD for each command line
(tests/golden/make_golden.py)."""
import io
import json
import os
import pickle

import numpy as np
import pytest

import ksa_oracle as orc
from conftest import GOLDEN, golden, load_pkg


@pytest.fixture(scope="module")
def K():
    load_pkg()
    return __import__("importlib").import_module("prgs-sdr-kspecanal_amd.kspecanal")


CLI = json.load(open(os.path.join(GOLDEN, "cli_args.json")))


@pytest.mark.parametrize("name", sorted(CLI))
def test_handle_args_matches_reference(K, name, capsys):
    case = CLI[name]
    d = K.handle_args({}, case["argv"] + ["prgLoopCnt", "0"])
    for k, want in case["d"].items():
        assert d[k] == want, (name, k, d[k], want)
    # window tables are the reference's numpy tables (K:932-936)
    assert np.array_equal(d["theWin"], orc.window_table(d["window"], d["fftSize"]))


def test_handle_args_errors_quit(K):
    with pytest.raises(SystemExit):
        K.handle_args({}, ["zeroSpan", "noSuchKey", "1"])          # K:908-910
    with pytest.raises(SystemExit):
        K.handle_args({}, ["zeroSpan", "window", "blackman"])
    d = {}
    with pytest.raises(SystemExit):
        K.handle_args(d, ["zeroSpan", "curScanCumuMode", "median"])
    assert d["cmd.stop"] is True


def test_scan_geometry_matches_reference_probes(K):
    d = K.handle_args({}, ["fmScan"])
    groups, total, centers = K.scan_geometry(d)
    assert (groups, total, len(centers)) == (9, 147456, 18)
    assert centers[0] == 89.2e6 and abs(centers[-1] - 109.6e6) < 1
    d = K.handle_args({}, ["quickFullScan"])
    groups, total, centers = K.scan_geometry(d)
    assert (groups, total, len(centers)) == (613, 39232, 1226)
    assert centers == orc.scan_steps(d["startFreq"], d["endFreq"], d["samplingRate"], 0.5)
    d = K.handle_args({}, ["scan", "startFreq", "100e6", "endFreq", "104.8e6", "scanRangeNonOverlap", "0.3"])
    with pytest.raises(SystemExit):                                 # K:588-593
        K.scan_geometry(d)


def test_plot_side_helpers(K):
    d = K.handle_args({}, ["zeroSpan", "fftSize", "1024", "xRes", "128"])
    y = np.random.default_rng(1).standard_normal(1024)
    for mode in ("MAX", "AVG"):
        assert np.array_equal(K._plotcompress(d, y, mode), orc.plotcompress(y, 128, mode))
    x = np.arange(1024.0)
    xs, ys = K.data_plotcompress(d, x, y, "MAX")
    assert len(xs) == len(ys) == 128 and xs[0] == np.average(x[:8])
    assert K.data_plotcompress(d, x, y, "RAW")[1] is y
    # plot_highs (K:243-272): strongest first, closer than delta*span to a marked one is skipped
    lv = np.full(100, -50.0)
    lv[[10, 11, 40, 90, 60, 20]] = [-5, -6, -7, -8, -9, -10]
    fr = np.linspace(0, 99, 100)
    d.update(pltHighsNumMarkers=4, pltHighsDelta4Marking=0.025, plt=None)
    marks = K.plot_highs(d, fr, lv)
    assert [m[0] for m in marks] == [10.0, 40.0, 90.0, 60.0]       # bin 11 is within 2.475 of bin 10


def test_save_stream_written_by_the_reference_is_readable(K, tmp_path):
    """zeroSpanSave files are pickle streams (K:511-525); ours must read the reference's own output."""
    g = golden("zerospan_save_n512")
    path = tmp_path / "ref.save"
    path.write_bytes(g["stream"].tobytes())
    d = K.handle_args({}, ["zeroSpanPlay", "fftSize", "512", "zeroSpanPlayFile", str(path)])
    K.zero_span_play_setup(d)
    try:
        assert [d["centerFreq"], d["samplingRate"], d["gain"]] == list(g["header"])
        for f in range(int(g["frames"])):
            spec = K.zero_span_play(d)
            assert np.array_equal(spec, g["spectra"][f])
        assert K.zero_span_play(d) is None and d["cmd.stop"] is True   # EOF -> stop flag, K:559-563
    finally:
        d["zeroSpanFile"].close()
        K.sdr_curscan = K._gpu_curscan


def test_restricted_unpickler_refuses_code(K):
    evil = pickle.dumps(os.system)
    with pytest.raises(pickle.UnpicklingError):
        K._load(io.BytesIO(evil))
    ok = pickle.dumps(np.arange(4.0))
    assert np.array_equal(K._load(io.BytesIO(ok)), np.arange(4.0))
    assert K._load(io.BytesIO(pickle.dumps(3.5))) == 3.5


def test_siglvls_roundtrip(K, tmp_path):
    p = str(tmp_path / "lv.pkl")
    d = K.handle_args({}, ["zeroSpan", "fftSize", "64", "SaveSigLvls", p])
    d["Fft.Avg"] = np.linspace(-60, -20, 64)
    K._save_siglvls(d)
    d2 = K.handle_args({}, ["zeroSpan", "fftSize", "64", "AdjSigLvls", p])
    K._load_siglvls(d2)
    assert np.array_equal(d2["Fft.Adj"], d["Fft.Avg"])
    d3 = K.handle_args({}, ["zeroSpan", "fftSize", "64", "centerFreq", "100e6", "AdjSigLvls", p])
    K._load_siglvls(d3)                                             # range mismatch -> dropped (K:759-763)
    assert d3["Fft.Adj"] is None and d3["AdjSigLvls"] == ""


def test_sources_shape_and_determinism():
    pkg = load_pkg()
    src = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.sources")
    a, b = src.SyntheticSdr(seed=5), src.SyntheticSdr(seed=5)
    a.center_freq = b.center_freq = 100.2e6
    x = a.read_samples(4096)
    assert x.dtype == np.complex128 and np.array_equal(x, b.read_samples(4096))
    spec = np.abs(np.fft.fftshift(np.fft.fft(x * np.hanning(4096))))
    f = np.fft.fftshift(np.fft.fftfreq(4096, 1 / 2.4e6)) + 100.2e6
    assert abs(f[np.argmax(spec)] / 1e6 - round(f[np.argmax(spec)] / 1e6)) < 0.002   # tones sit on whole MHz
    raw = src.SyntheticSdr(seed=5).read_bytes(64)
    assert raw.dtype == np.uint8 and len(raw) == 64


def test_c_abi_exports_every_declared_symbol():
    """include/ksa.h vs the built library: every declared entry point is exported (no compute calls)."""
    import re
    pkg = load_pkg()
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "ksa.h")).read()
    declared = set(re.findall(r"\b(ksa_[a-z0-9_]+)\s*\(", hdr))
    lib = __import__("importlib").import_module("prgs-sdr-kspecanal_amd._lib")
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    for name in declared:
        assert hasattr(pkg.lib, name)
    assert pkg.lib.ksa_abi_version() == 3
    # without a GPU the library must fail loudly, not fall back
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(pkg.KsaError):
            pkg.SpectrumEngine(512)


def test_header_is_plain_c_and_library_exports_match(tmp_path):
    """include/ksa.h is the drop-in boundary: it must compile as C99 on its own (plain pointers and sizes, no C++ /
    torch types), and the dynamic symbol table of libksa.so must carry exactly the declared entry points."""
    import re
    import subprocess
    root = os.path.join(os.path.dirname(GOLDEN), "..")
    hdr = os.path.join(root, "include", "ksa.h")
    src = tmp_path / "use_ksa.c"
    names = sorted(set(re.findall(r"\b(ksa_[a-z0-9_]+)\s*\(", open(hdr).read())))
    src.write_text('#include "ksa.h"\n#include <stddef.h>\n'
                   'typedef void (*fn_t)(void);\nstatic const fn_t table[] = {' + ", ".join("(fn_t)%s" % n for n in names) + '};\n'
                   'int use_ksa(void) { ksa_config c; c.abi_version = KSA_ABI_VERSION; return (int)sizeof(table) + c.abi_version + KSA_HM_ROWS; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-c", str(src),
                        "-o", str(tmp_path / "use_ksa.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = os.path.join(root, "prgs-sdr-kspecanal_amd", "libksa.so")
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True)
    assert nm.returncode == 0, nm.stderr
    exported = {ln.split()[-1] for ln in nm.stdout.splitlines() if " T " in ln and ln.split()[-1].startswith("ksa_")}
    assert exported == set(names), exported ^ set(names)


def test_data_2d_plotcompress_rows():
    """K:224-237: row-wise _data_plotcompress; the scan's initial waterfall buffer (K:613-614) is its only use in the reference."""
    load_pkg()
    k = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.kspecanal")
    import ksa_oracle as orc
    d = {"xRes": 8, "pltCompressHM": "MAX"}
    rng = np.random.default_rng(3)
    data = rng.standard_normal((5, 64))
    got = k.data_2d_plotcompress(d, data)
    want = np.array([orc.plotcompress(data[r], 8, "MAX") for r in range(5)])
    assert got.shape == (5, 8) and np.array_equal(got, want)
    assert np.array_equal(k.data_2d_plotcompress(d, data, "AVG"), np.array([orc.plotcompress(data[r], 8, "AVG") for r in range(5)]))
    assert k.data_2d_plotcompress(d, data, "RAW") is data
    hm = np.ones((128, 64)) * 3.9e-8                      # K:613: ones * minAmp4Clip
    assert np.array_equal(k.data_2d_plotcompress(d, hm), np.full((128, 8), 3.9e-8))
