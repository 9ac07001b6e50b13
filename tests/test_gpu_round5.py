"""GPU tests, round 5: the documented reference-side binding (INTEGRATION.md sections 2 and 3) executed verbatim against the
reference-run goldens, the live shader-clock figure, and the bench record's own
consistency (the limiter sentence quotes the fields beside it; the uint8 side run; the backend the collective names).
All through the C ABI; tolerances as test_gpu_parity.py."""
import ctypes as C
import re

import numpy as np
import pytest

import ksa_oracle as orc
from conftest import golden, load_pkg, ROOT
from test_gpu_parity import assert_db, assert_lin, GAIN
from test_gpu_round2 import _bench
from test_host_cli import integration_blocks

pytestmark = pytest.mark.gpu
CURVES = ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


# ------------------------------------------------------------------------------- INTEGRATION.md sections 2 + 3, verbatim
class _Quit(Exception):
    pass


def _doc_namespace(blocks_iq, d):
    """What the reference's module namespace provides around the patch (K:312-347 sdr_read, K:967-972 prg_quit, the dict gD):
    stand-ins fed from the golden's own IQ blocks."""
    feed = iter(blocks_iq)

    def sdr_read(sdr, length):
        x = next(feed)
        assert len(x) == length
        return x.astype(np.complex128)          # sdr_read hands complex128 (K:335)

    def prg_quit(dd, msg):
        raise _Quit(msg)

    return {"sdr_read": sdr_read, "prg_quit": prg_quit, "gD": d, "d": d, "__name__": "kspecanal_patch"}


def test_integration_section2_runs_verbatim_against_the_curscan_golden(ksa, torch_cuda):
    """VERDICT r04 item 1(c).  The ```python block of INTEGRATION.md section 2 -- the smallest patch a maintainer of the
    reference pastes after handle_args(gD) -- is executed as it stands in the document (only `libksa.so` is resolved to the
    in-tree build) and `sdr_curscan_gpu(d)` must reproduce the reference's own sdr_curscan output (K:351-397) for every
    window x mode pair of the golden."""
    g = golden("curscan_n4096_q05")
    n, full, q = int(g["fft_size"]), int(g["full"]), float(g["non_overlap"])
    src = integration_blocks()["2"][0].replace('C.CDLL("libksa.so")', "C.CDLL(%r)" % ksa.LIB_PATH)
    ran = 0
    for key in [k for k in g.files if k.split("_")[-1] in ("AVG", "MAX", "MIN", "RAW")]:
        win, mode = key.split("_")
        d = {"fftSize": n, "fullSize": full, "curScanNonOverlap": q, "curScanCumuMode": mode,
             "theWin": orc.window_table(win, n), "gain": GAIN, "minAmp4Clip": (1 / 256) * 1e-5, "xRes": 512, "sdr": object()}
        ns = _doc_namespace([g["iq"]], d)
        exec(compile(src, "INTEGRATION.md#2", "exec"), ns)             # runs ksa_open(gD) and rebinds sdr_curscan
        assert ns["sdr_curscan"] is ns["sdr_curscan_gpu"]
        out = ns["sdr_curscan"](d)
        assert out.dtype == np.float64 and out.shape == (n,)
        assert_lin(out, g[key], what="INTEGRATION section 2, %s" % key)
        ns["ksa"].ksa_destroy(d["ksa"])
        ran += 1
    assert ran >= 8


@pytest.mark.parametrize("tag", ["n4096", "hm_n512"])
def test_integration_section3_runs_verbatim_against_the_zerospan_golden(ksa, torch_cuda, tag):
    """Section 3 -- the fused replacement of the frame loop body K:464-484 -- executed verbatim once per frame on the engine
    section 2 opened, against the reference's own zeroSpan runs (Fft.Cur/Max/Min/Avg; the waterfall ring where the golden
    holds it)."""
    g = golden("zerospan_" + tag)
    n, full, q, frames = int(g["fft_size"]), int(g["full"]), float(g["non_overlap"]), int(g["frames"])
    x = g["iq"].reshape(frames, full)
    blocks = integration_blocks()
    open_src = blocks["2"][0].replace('C.CDLL("libksa.so")', "C.CDLL(%r)" % ksa.LIB_PATH)
    body = compile(blocks["3"][0], "INTEGRATION.md#3", "exec")
    xres = int(g["xres"])
    d = {"fftSize": n, "fullSize": full, "curScanNonOverlap": q, "curScanCumuMode": "AVG",
         "theWin": orc.window_table(str(g["window"]), n), "gain": float(g["gain"]), "minAmp4Clip": (1 / 256) * 1e-5, "xRes": xres,
         "sdr": object(), "bDataMax": True, "bDataMin": True, "bDataAvg": True, "PltHeatMapWidth": min(n, xres)}
    ns = _doc_namespace(list(x), d)
    exec(compile(open_src, "INTEGRATION.md#2", "exec"), ns)
    for _ in range(frames):
        exec(body, ns)
    for k in CURVES:
        assert_db(d[k], g[k.replace("Fft.", "").lower()], what="INTEGRATION section 3 " + k)
    assert ns["indexHM"] == frames % 128
    if "hm" in g.files:
        assert_db(ns["fftHM"][:frames], g["hm"][:frames], what="INTEGRATION section 3 waterfall")
    ns["ksa"].ksa_destroy(d["ksa"])


def _engine(ksa, n, full, q, frames, **kw):
    return ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning", gain=GAIN, xres=512, max_frames=frames, **kw)


@pytest.mark.parametrize("tag", ["3band_n512", "dummy_n512"])
def test_integration_section4_runs_verbatim_against_the_scan_goldens(ksa, torch_cuda, tag):
    """Section 4 -- the body of the reference's step loop K:621-668 + the row of K:696-697 as "capture the pass, hand it over"
    (host pointers, ksa_scan_pass_c64) -- executed verbatim once per pass against the reference's own scans: a three-band scan,
    and one with failed tunes (sdr_setup returning False -> the dummy ones band, K:637-639)."""
    import ctypes as C
    from test_gpu_round2 import _regen_iq
    g = golden("scan_" + tag)
    n, full, passes, steps = int(g["fft_size"]), int(g["full"]), int(g["passes"]), int(g["steps"])
    step_ok = g["step_ok"].astype(bool) if "step_ok" in g.files else np.ones((passes, steps), dtype=bool)
    # the stream the reference consumed: one block per SUCCESSFUL tune, in order (a failed tune reads nothing, K:635-639)
    stream = (g["iq"] if "iq" in g.files else _regen_iq(g, full * int(step_ok.sum()))).reshape(-1, full)
    reads = iter(stream)
    groups = int((float(g["end_freq"]) - float(g["start_freq"])) / float(g["sampling_rate"]))
    total = groups * n
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=float(g["non_overlap"]), window=str(g["window"]), gain=float(g["gain"]),
                             min_amp=float(g["min_amp"]), xres=int(g["xres"]), max_frames=steps, scan_total_entries=total,
                             scan_non_overlap=float(g["scan_non_overlap"]))
    body = compile(integration_blocks()["4"][0], "INTEGRATION.md#4", "exec")
    d = {"fullSize": full, "samplingRate": float(g["sampling_rate"]), "gain": float(g["gain"]), "xRes": int(g["xres"]), "sdr": object(),
         "ksa": eng._h}
    for p in range(passes):
        tunes = iter(step_ok[p])
        ns = {"np": np, "C": C, "ksa": ksa.lib, "d": d, "total": total, "centers": list(range(steps)),
              "sdr_setup": lambda sdr, fc, fs, gain: bool(next(tunes)),           # K:630: False = the tune failed
              "sdr_read": lambda sdr, length: next(reads).astype(np.complex128),   # K:636 reads only after a good tune
              "prg_quit": lambda dd, msg: (_ for _ in ()).throw(_Quit(msg))}
        exec(body, ns)
    for k in CURVES:
        assert_db(d[k], g[k.replace("Fft.", "").lower()], what="INTEGRATION section 4 %s %s" % (tag, k))
    assert_db(d["fftHM"][:passes], g["hm"][:passes], what="INTEGRATION section 4 waterfall")
    assert ns["idx"].value == int(g["hm_index"]) and ns["passes"].value == passes
    eng.close()


# ------------------------------------------------------------------------------- live shader clock (ksa_prof_clock)
def test_prof_clock_reports_the_clock_held_under_the_stage(ksa, torch_cuda):
    """VERDICT r04 item 4.  Stamp kernels around each profiled spectrum stage read s_memtime / s_memrealtime per XCD; the median
    quotient is the shader clock the chip held: between 1 and 2.6 GHz on an MI355X (2.4 GHz nominal), from at least one
    sample per XCD that ran a stamp; an engine that never profiled refuses, and the stamps change no result."""
    torch = torch_cuda
    n, full, q, frames = 4096, 32768, 0.5, 8192
    x = orc.synth_iq(full * 64, 17).astype(np.complex64).reshape(64, full)
    iq = torch.view_as_real(torch.from_numpy(x)).to("cuda").repeat(frames // 64, 1, 1).contiguous()
    plain, prof = _engine(ksa, n, full, q, frames), _engine(ksa, n, full, q, frames)
    with pytest.raises(ksa.KsaError):
        plain.prof_clock()
    prof.prof_enable(True)
    for _ in range(4):
        plain.frames_dev(iq, ksa.FMT_C64, frames)
        prof.frames_dev(iq, ksa.FMT_C64, frames)
    ms, launches = prof.prof_read()
    ghz, samples = prof.prof_clock()
    assert launches == 4 and ms > 0
    assert samples >= 4 and 1.0 < ghz < 3.0, (ghz, samples, prof.prof_clock_range)
    a, b = plain.state(), prof.state()
    for k in CURVES + ("fftHM",):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    plain.close()
    prof.close()


# ------------------------------------------------------------------------------- the bench record says what it measured
def _decimals(text):
    return [float(t) for t in re.findall(r"(?<![\w.])\d+\.\d+", text)]


@pytest.mark.parametrize("cfg,extra", [(2, ["--frames", "8192"]), (3, ["--passes", "8"]), (4, ["--passes", "16"]), (5, ["--frames", "16"])])
def test_bench_limiter_quotes_the_fields_beside_it(cfg, extra):
    """every decimal number in roofline.limiter is one of the block's own fields (two decimals), the live
    clock and flop_frac_at_clock are present and consistent."""
    d = _bench(["--config", str(cfg), "--steps", "3", "--warmup", "1", "--no-cpu", "--no-secondary"] + extra)
    rf = d["roofline"]
    fields = [rf[k] for k in ("frac", "flop_frac", "valu_issue_frac", "lds_frac", "traffic_over_algorithmic") if rf.get(k) is not None]
    quoted = _decimals(rf["limiter"])
    assert quoted, rf["limiter"]
    for v in quoted:
        assert any(abs(v - round(f, 2)) < 5e-3 for f in fields), (v, rf["limiter"], fields)
    if rf.get("valu_issue_frac") is None:
        assert "counters not available" in rf["limiter"]
    assert rf["shader_clock_ghz_live"] and 1.0 < rf["shader_clock_ghz_live"] < 2.6 and rf["shader_clock_samples"] >= 3
    want = rf["tflops"] / (157.3 * rf["shader_clock_ghz_live"] / 2.4)
    assert abs(rf["flop_frac_at_clock"] - want) < 1e-9 and rf["flop_frac_at_clock_source"] == "shader_clock_ghz_live"


def test_bench_rehearsal_lines_say_so_and_name_their_backend():
    """VERDICT r04 item 2 / what's weak 7: ranks sharing a device over gloo -> `value_is_rehearsal` and a collective string that
    names gloo, not RCCL."""
    d = _bench(["--config", "2", "--gpus", "2", "--frames", "512", "--steps", "2", "--warmup", "1", "--no-cpu"],
               env={"KSA_BENCH_BACKEND": "gloo"})
    assert d["value_is_rehearsal"] is True and "REHEARSAL" in d["multi_gpu_note"]
    assert d["config"]["collective"].startswith("gloo") and "RCCL" not in d["config"]["collective"]
    e = _bench(["--config", "4", "--inprocess", "--gpus", "2", "--passes", "4", "--steps", "2", "--warmup", "1"])
    assert e["value_is_rehearsal"] is True


# ------------------------------------------------------------------------------- error paths leave nothing behind
def test_read_view_refuses_before_it_enqueues_and_zeroes_found(ksa, torch_cuda):
    """ksa_read_view on an engine without a waterfall (hm_width 0 is impossible through SpectrumEngine, so: the scan ring of an
    engine created without scan geometry, and a row count outside the ring) fails BEFORE its first asynchronous copy -- the
    caller's buffers are untouched -- and a call without markers reports found = 0."""
    import ctypes as C
    n, full = 1024, 8192
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, xres=256, max_frames=4)
    x = orc.synth_iq(full * 4, 3).astype(np.complex64).reshape(4, full)
    for f in range(4):
        eng.frame(x[f])
    lib = ksa.lib
    lv = np.full((4, 256), 7.0, dtype=np.float32)
    rows = np.full((2, 256), 7.0, dtype=np.float32)
    found, hm_index = C.c_int32(99), C.c_int32(-5)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    # scan view on an engine without scan geometry
    rc = lib.ksa_read_view(eng._h, 1, 0, 256, vp(lv), 0, 0.0, 0, None, None, C.byref(found), 2, vp(rows), C.byref(hm_index))
    assert rc != 0 and b"scan" in lib.ksa_last_error()
    assert np.all(lv == 7.0) and np.all(rows == 7.0)
    # more rows than the ring has
    rc = lib.ksa_read_view(eng._h, 0, 0, 256, vp(lv), 0, 0.0, 0, None, None, C.byref(found), 129, vp(rows), C.byref(hm_index))
    assert rc != 0 and np.all(lv == 7.0) and np.all(rows == 7.0)
    # a good call without markers: found is written (0), the rows arrive
    found.value = 99
    rc = lib.ksa_read_view(eng._h, 0, 0, 256, vp(lv), 0, 0.0, 0, None, None, C.byref(found), 2, vp(rows), C.byref(hm_index))
    assert rc == 0 and found.value == 0 and hm_index.value == 4
    assert np.array_equal(rows.astype(np.float64), eng.state()["fftHM"][[2, 3]]) and not np.all(lv == 7.0)
    eng.close()


def test_abandoned_band_sharded_batch_does_not_block_the_next_allstitch(ksa, torch_cuda):
    """ksa_scan_allstitch refuses while an engine still holds unmerged partial rows (scan_rows > 0).  A batch abandoned after
    ksa_scan_stitch_range_dev used to leave that flag set for ever; ksa_scan_reset clears it now, and so does a failing
    allstitch on its way out."""
    torch = torch_cuda
    n, full = 256, 2048
    mk = lambda: ksa.SpectrumEngine(n, full_size=full, non_overlap=0.5, window="hanning", gain=GAIN, min_amp=1e-7, xres=64,
                                    max_frames=16, scan_total_entries=2 * n)
    a, b = mk(), mk()
    steps, passes = 4, 2
    own = torch.full((passes, 2, n), -40.0, dtype=torch.float32, device="cuda")
    # engine a stitches its range alone and never merges the rows: the batch is abandoned
    lo, hi, nhalo, e_lo, e_hi = a.scan_shard(steps, 0, 2)
    a.scan_stitch_range_dev(own, None, 0, lo, hi, steps, passes, e_lo, e_hi)
    with pytest.raises(ksa.KsaError, match="unmerged partial rows"):
        ksa.scan_allstitch([a, b], [own, own], steps, passes)
    a.scan_reset()
    b.scan_reset()
    ksa.scan_allstitch([a, b], [own, own], steps, passes)
    assert a.scan_state()["passes"] == passes and b.scan_state()["passes"] == passes
    for e in (a, b):
        e.close()


# ------------------------------------------------------------------------------------------ the 8 x 8 plan of N = 64
def test_n64_complex_input_runs_the_8x8_kernel_within_its_register_budget(ksa):
    """ksa_kernel_info names what complex64 input runs at N = 64: path 5 = spectrum64_kernel (ksa_kernels64.hpp), one wave per
    workgroup, at most 128 VGPRs (four waves per SIMD) and the LDS of Plan64."""
    eng = ksa.SpectrumEngine(64, full_size=512, non_overlap=0.1, window="ones")
    info = eng.kernel_info()
    assert info["path"] == 5 and info["threads"] == 64 and info["vgprs"] <= 128 and info["lds_bytes"] == 16 * 68 * 8 + 64 * 4
    eng.close()


def test_n64_rectangular_window_kernel_equals_the_generic_one(ksa):
    """An all-ones window table selects spectrum64_kernel<.., W1 = true> (no tap multiplies; the reference's default window,
    K:52).  A table that is all ones except ONE tap of 1 + 2^-23 runs the generic kernel: the two spectra must agree to the
    last-bit effect of that one tap (and each with the oracle) -- x * 1.0f is x."""
    n, full, q = 64, 512, 0.1
    x = orc.synth_iq(full, 909).astype(np.complex64)
    ones = np.ones(n, dtype=np.float32)
    almost = ones.copy()
    almost[17] = np.nextafter(np.float32(1.0), np.float32(2.0))
    for m in ("AVG", "MAX", "MIN"):
        a = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="ones", cumu_mode=m)
        b = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=almost, cumu_mode=m)
        ya, yb = a.curscan(x), b.curscan(x)
        assert_lin(ya, orc.curscan(x, n, q, ones.astype(np.float64), m), what="W1 " + m)
        assert np.max(np.abs(ya - yb)) <= 4e-7 * np.max(ya), "the rectangular-window kernel differs from the generic one by more than one tap's last bit"
        a.close(); b.close()


@pytest.mark.parametrize("full,q", [(64, 0.5), (100, 0.1), (160, 0.5), (512, 0.1), (512, 0.25), (1100, 0.5), (1111, 0.07), (4096, 0.1)])
def test_n64_8x8_kernel_every_fold_mode_and_round_shape(ksa, torch_cuda, full, q):
    """spectrum64_kernel against the oracle's curscan (K:351-397) for window counts of 1, one partial round, exact multiples of
    16, 71 (quickFullScan) and hundreds of windows (many rounds), every fold mode and window; on-bin tone known answer;
    complex64 and uint8 input of the same samples agree (uint8 runs the 4 x 16 kernel: two plans, one spectrum)."""
    n = 64
    starts = orc.window_starts(full, n, q)
    x = orc.synth_iq(full, 4242 + full).astype(np.complex64)
    for w, m in (("hanning", "AVG"), ("kaiser", "MAX"), ("ones", "MIN"), ("hamming", "RAW")):
        eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window=w, cumu_mode=m)
        assert eng.num_windows == len(starts) and eng.kernel_info()["path"] == 5
        assert_lin(eng.curscan(x), orc.curscan(x, n, q, orc.window_table(w, n), m), what="N=64 full=%d q=%s %s %s" % (full, q, w, m))
        eng.close()
    # the same samples as uint8 I,Q (4 x 16 kernel) and as the complex64 values they unpack to (8 x 8 kernel)
    raw = orc.quantize_u8(x * 0.8)
    xq = orc.unpack_u8(raw).astype(np.complex64)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="hanning")
    a, b = eng.curscan(raw), eng.curscan(xq)
    assert_lin(a, b, tol=2e-6, what="u8 (4 x 16) against c64 (8 x 8)")
    eng.close()
    # a tone on bin 19 of 64 under a rectangular window: one bin holds everything (the oracle's value: K:391's 2 / N scale)
    t = np.exp(2j * np.pi * 19 * np.arange(full) / n).astype(np.complex64)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="ones")
    y, want = eng.curscan(t), orc.curscan(t, n, q, orc.window_table("ones", n), "AVG")
    k = int(np.argmax(want))
    assert int(np.argmax(y)) == k and abs(y[k] - want[k]) < 1e-5 * want[k] and np.max(np.delete(y, k)) < 1e-5 * want[k]
    eng.close()


@pytest.mark.parametrize("frames", [1, 15, 16, 17, 4099])
def test_n64_8x8_kernel_batches_smaller_and_larger_than_its_grid(ksa, torch_cuda, frames):
    """frames_dev at N = 64 (complex64): fewer frames than workgroups, one more than a multiple, and more frames than the
    persistent grid holds -- per-frame dB rows, waterfall rows and the running curves against the oracle's sequential loop."""
    torch = torch_cuda
    n, full, q = 64, 512, 0.1
    x = orc.synth_iq(full * frames, 77 + frames).astype(np.complex64).reshape(frames, full)
    st_ref, db_ref, _ = orc.zerospan_batch(x, n, q, orc.window_table("kaiser", n), "AVG", GAIN, 64)
    eng = ksa.SpectrumEngine(n, full_size=full, non_overlap=q, window="kaiser", gain=GAIN, xres=64, max_frames=frames)
    cur_db = torch.empty((frames, n), dtype=torch.float32, device="cuda")
    rows = torch.empty((frames, eng.hm_width), dtype=torch.float32, device="cuda")
    eng.frames_dev(torch.view_as_real(torch.from_numpy(x)).cuda(), ksa.FMT_C64, frames, cur_db=cur_db, hm_rows=rows)
    torch.cuda.synchronize()
    st = eng.state()
    assert_db(cur_db.cpu().numpy(), db_ref, what="per-frame dB, %d frames" % frames)
    want_rows = np.array([orc.plotcompress(r, eng.hm_width, "MAX") for r in db_ref])
    assert_db(rows.cpu().numpy(), want_rows, what="per-frame rows")
    for k in ("cur", "max", "min", "avg"):
        assert_db(st["Fft." + k.capitalize()], getattr(st_ref, k), what="N=64 %s" % k)
    eng.close()
