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
    abi = int(re.search(r"#define\s+KSA_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert pkg.lib.ksa_abi_version() == abi == lib.ABI_VERSION     # header, library and binding agree
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


# ------------------------------------------------------------------------------- INTEGRATION.md stays executable
def integration_blocks():
    """{section number: [source of every ```python block of that section]} of INTEGRATION.md."""
    import re
    text = open(os.path.join(os.path.dirname(GOLDEN), "..", "INTEGRATION.md")).read()
    out, sec = {}, None
    pos = 0
    for m in re.finditer(r"^## (\w+)\.[^\n]*$|^```python\n(.*?)^```", text, flags=re.M | re.S):
        if m.group(1) is not None:
            sec = m.group(1)
        else:
            out.setdefault(sec, []).append(m.group(2))
    return out


def _ksa_calls(tree):
    """(name, number of positional arguments or None when a *starred argument makes it unknowable) of every ksa.ksa_*(...) call,
    plus every bare ksa.ksa_* attribute."""
    import ast
    calls, names = [], set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "ksa" and node.attr.startswith("ksa_"):
            names.add(node.attr)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) \
                and node.func.value.id == "ksa" and node.func.attr.startswith("ksa_"):
            starred = any(isinstance(a, ast.Starred) for a in node.args)
            calls.append((node.func.attr, None if starred else len(node.args)))
    return calls, names


def test_integration_doc_blocks_compile_and_match_the_binding():
    """Every ```python block of INTEGRATION.md sections 2-5 is valid Python, names only entry points the library has
    (_lib.SIGNATURES) with the right number of arguments, and its KsaConfig mirrors _lib.Config field for field; the ABI
    handed to ksa_create comes from ksa_abi_version(), never from a literal; the ABI number and the entry-point count the text
    quotes are the header's.  Bumping KSA_ABI_VERSION, adding an entry point or a config field without touching the document
    turns this red."""
    import ast
    import ctypes as C
    import re
    load_pkg()
    lib = __import__("importlib").import_module("prgs-sdr-kspecanal_amd._lib")
    blocks = integration_blocks()
    assert {"2", "3", "4", "5"} <= set(blocks), sorted(blocks)
    seen_cfg = False
    for sec in ("2", "3", "4", "5"):
        for src in blocks[sec]:
            tree = ast.parse(src, "INTEGRATION.md#%s" % sec)
            calls, names = _ksa_calls(tree)
            assert names, "section %s names no entry point" % sec
            for name in names:
                assert name in lib.SIGNATURES, "INTEGRATION.md section %s: %s is not an entry point of include/ksa.h" % (sec, name)
            for name, nargs in calls:
                if nargs is not None:
                    assert nargs == len(lib.SIGNATURES[name][1]), "section %s: %s called with %d arguments, the ABI takes %d" % (
                        sec, name, nargs, len(lib.SIGNATURES[name][1]))
            for node in ast.walk(tree):
                if isinstance(node, ast.ClassDef) and node.name == "KsaConfig":
                    fields = [s for s in node.body if isinstance(s, ast.Assign) and s.targets[0].id == "_fields_"][0]
                    got = eval(compile(ast.Expression(fields.value), "KsaConfig._fields_", "eval"), {"C": C})
                    assert got == list(lib.Config._fields_), "KsaConfig in the document differs from ksa_config / _lib.Config"
                    seen_cfg = True
                if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "KsaConfig":
                    first = node.args[0]
                    assert isinstance(first, ast.Call) and ast.unparse(first.func) == "ksa.ksa_abi_version", \
                        "abi_version must come from ksa.ksa_abi_version(), not from %s" % ast.unparse(first)
                    assert len(node.args) == len(lib.Config._fields_)
    assert seen_cfg
    text = open(os.path.join(os.path.dirname(GOLDEN), "..", "INTEGRATION.md")).read()
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "ksa.h")).read()
    abi = int(re.search(r"#define\s+KSA_ABI_VERSION\s+(\d+)", hdr).group(1))
    quoted = re.search(r"`KSA_ABI_VERSION (\d+)` \((\d+) entry points", text)
    assert quoted and int(quoted.group(1)) == abi == lib.ABI_VERSION, "INTEGRATION.md quotes another ABI version than include/ksa.h"
    assert int(quoted.group(2)) == len(lib.SIGNATURES), "INTEGRATION.md quotes %s entry points, the header declares %d" % (
        quoted.group(2), len(lib.SIGNATURES))


def test_integration_doc_check_catches_a_stale_document():
    """The checker itself: a block that passes a literal ABI, misspells an entry point or drops an argument is caught."""
    import ast
    lib = __import__("importlib").import_module("prgs-sdr-kspecanal_amd._lib")
    calls, names = _ksa_calls(ast.parse("ksa.ksa_frame_c64(h)\nksa.ksa_nope(h)\nksa.ksa_read_state(h, *bufs)"))
    assert ("ksa_frame_c64", 1) in calls and ("ksa_read_state", None) in calls and "ksa_nope" in names
    assert "ksa_nope" not in lib.SIGNATURES and len(lib.SIGNATURES["ksa_frame_c64"][1]) == 2


# ------------------------------------------------------------------------------- every entry point runs on its engine's device
def _c_functions(src):
    """{name: body} of the functions defined at brace depth 0 of a C++ source (good enough for ksa_api.hip's extern "C" part)."""
    import re
    out = {}
    for m in re.finditer(r"^(?:static\s+)?(?:int|void|const char\*)\s+(\w+)\s*\(([^)]*)\)\s*\{", src, flags=re.M):
        depth, i = 1, m.end()
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        out[m.group(1)] = (m.group(2), src[m.end():i])
    return out


def test_every_entry_point_that_touches_hip_selects_its_engines_device():
    """include/ksa.h promises that every entry point selects its engine's device for its own duration and hands the caller's
    current device back.  The default engine stream is the NULL stream -- "the null stream of the CURRENT device" -- so an
    entry point that forgets (ksa_synchronize and ksa_prof_read did in round 4) silently works on another GPU.  Static check of
    ksa_api.hip: every exported ksa_* with an engine (or engine array) parameter whose body issues a HIP call, a launch or one
    of the helpers that do must hold a DeviceGuard.  (A one-GPU box cannot see the difference at run time.)"""
    import re
    root = os.path.join(os.path.dirname(GOLDEN), "..")
    src = open(os.path.join(root, "prgs-sdr-kspecanal_amd", "csrc", "ksa_api.hip")).read()
    ext = src[src.index('extern "C" {'):]
    fns = _c_functions(ext)
    hdr = open(os.path.join(root, "include", "ksa.h")).read()
    exported = set(re.findall(r"\b(ksa_[a-z0-9_]+)\s*\(", hdr))
    assert exported <= set(fns) | {"ksa_abi_version", "ksa_last_error"}, exported - set(fns)
    helpers = r"run_spectrum|run_accumulate|do_commit|fill|scan_reset|join_side|levels_to_scratch|copy_ring_rows|scan_stitch|gather_all|upload|ensure|scan_spectra|copy_d2d|ensure_events"
    touches = re.compile(r"\bhip[A-Z]\w*\s*\(|\bHIP_OK\b|hipLaunchKernelGGL|\b(?:%s)\s*\(" % helpers)
    checked = 0
    for name in sorted(exported & set(fns)):
        params, body = fns[name]
        if "ksa_engine" not in params or not touches.search(body):
            continue
        checked += 1
        assert "DeviceGuard" in body, "%s touches HIP without a DeviceGuard" % name
        # ... and selects the device itself or through a helper that does (scan_stitch, levels_to_scratch, check_handles' callers)
        assert re.search(r"hipSetDevice\s*\(|scan_stitch\s*\(|levels_to_scratch\s*\(|gather_all\s*\(", body), "%s never selects its engine's device" % name
    assert checked >= 25, checked
    for name in ("curscan_host", "frame_host", "scan_pass_host"):          # the static bodies behind the _c64 / _u8 pairs
        assert "DeviceGuard" in fns[name][1] and "hipSetDevice" in fns[name][1], name


# ------------------------------------------------------------------------------- the bench record's limiter sentence
def test_limiter_quotes_its_own_fields():
    """bench.limiter_sentence formats every decimal number it prints from the roofline block it is handed -- there is no
    hard-coded counter figure left to go stale -- and says so when no counter record matches the kernel sources."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("ksa_bench", os.path.join(os.path.dirname(GOLDEN), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for cfg in (2, 3, 4, 5):
        rf = {"frac": 0.4193, "flop_frac": 0.3361, "valu_issue_frac": 0.4916, "lds_frac": 0.3120, "traffic_over_algorithmic": 1.0003}
        text = bench.limiter_sentence(cfg, rf)
        got = sorted(float(t) for t in re.findall(r"(?<![\w.])\d+\.\d+", text))
        assert got == sorted(round(v, 2) for v in rf.values()), (cfg, text)
        bare = bench.limiter_sentence(cfg, {"frac": 0.07, "flop_frac": 0.29})
        assert "counters not available" in bare and sorted(float(t) for t in re.findall(r"(?<![\w.])\d+\.\d+", bare)) == [0.07, 0.29]
    assert set(bench.BOUND) == {2, 3, 4, 5} and all(isinstance(v, str) for v in bench.BOUND.values())
    assert bench.backend_name("nccl").startswith("RCCL") and bench.backend_name("gloo").startswith("gloo")


def test_tool_scripts_parse_and_are_listed():
    """Every script under tools/ parses (bash -n / py_compile) and has a row in tools/README.md -- the GPU box is the wrong
    place to find a syntax error, and an unlisted script is one nobody will find."""
    import glob
    import py_compile
    import subprocess
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    readme = open(os.path.join(tools, "README.md")).read()
    for f in sorted(glob.glob(os.path.join(tools, "*.sh"))):
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)
    for f in sorted(glob.glob(os.path.join(tools, "*.py"))):
        py_compile.compile(f, doraise=True)
    missing = [os.path.basename(f) for f in sorted(glob.glob(os.path.join(tools, "*.sh")) + glob.glob(os.path.join(tools, "*.py")) + glob.glob(os.path.join(tools, "*.hip")))
               if os.path.basename(f) not in readme]
    assert not missing, "not in tools/README.md: %s" % missing
