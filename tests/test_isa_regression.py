"""ISA regression guard (CPU suite; hipcc cross-compiles gfx950 without a GPU).

Round 4's +7 % at the headline configuration rests on two things only the compiler can take away again:
the exchange reads of the window loop are SINGLE ds_read_b64 (hipcc's load / store optimizer otherwise merges pairs into
ds_read2_b64 / ds_read2st64_b64 -- half the rate, and banked over 16-lane groups for which the layout is not conflict-free;
`lds_ld64` keeps them apart), and no spilled register is touched INSIDE the window loop (round 5's ping-pong form of the
config-2 kernel has none at all).  Nothing in the
numeric suites would notice a toolchain or source change that breaks either: the results stay right, only slower.

This test compiles the product translation unit (csrc/ksa_api.hip, the flags of build.py) to device assembly and checks, for
the three kernels the bench configurations run -- spectrum_kernel<4096,c64,RM=8,AVG> (config 2), spectrum64_kernel<c64,AVG,ones>
(config 4), spectrum32_kernel<16384,c64,AVG> (config 3) --: VGPRs / occupancy / scratch bytes as DESIGN.md states them, the
instruction mix of the window loop (the innermost loop that holds the fold's 16 / 32 v_sqrt_f32), no merged exchange reads, no
scratch traffic where the design says there is none.  A second, small translation unit with -DKSA_LDS_ATOMIC_LD=0 proves that
the check bites (the merged reads come back)."""
import collections
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "prgs-sdr-kspecanal_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-unused-value", "--cuda-device-only", "-S"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(src, out, extra=()):
    r = subprocess.run([_hipcc()] + FLAGS + list(extra) + ["-o", out, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out).read()


def _kernels(asm):
    """{demangled name: (body, resource comment block)} of every ksa:: kernel in a device assembly listing."""
    out = {}
    for m in re.finditer(r"^(_ZN3ksa\w+):[^\n]*\n", asm, flags=re.M):
        end = asm.index(".Lfunc_end", m.end())
        out[m.group(1)] = (asm[m.end():end], asm[end:end + 6000])
    names = subprocess.run(["c++filt"] + list(out), capture_output=True, text=True).stdout.split("\n")
    return {d: out[k] for k, d in zip(out, names)}


def _mix(text):
    return collections.Counter(re.findall(r"^\s+([a-z][a-z_0-9]+)", text, flags=re.M))


def _resource(tail, key):
    return int(re.search(r"; %s: *(\d+)" % key, tail).group(1))


def _window_loop(body, sqrts):
    """The innermost loop (smallest span between a label and a backward branch to it) that holds the fold's `sqrts`
    v_sqrt_f32 -- the AVG fold takes one square root per bin per window (K:391-395)."""
    labels = {m.group(1): m.start() for m in re.finditer(r"^(\.LBB\d+_\d+):", body, flags=re.M)}
    best = None
    for m in re.finditer(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", body, flags=re.M):
        t = m.group(1)
        if t in labels and labels[t] < m.start():
            seg = body[labels[t]:m.start()]
            c = _mix(seg)
            if c["v_sqrt_f32_e32"] + c["v_sqrt_f32_e64"] >= sqrts and (best is None or len(seg) < len(best)):
                best = seg
    assert best is not None, "no loop with %d v_sqrt_f32 found" % sqrts
    return best


def _find(kernels, needle):
    hits = [k for k in kernels if needle in k]
    assert len(hits) == 1, (needle, hits)
    return kernels[hits[0]]


@pytest.fixture(scope="module")
def product_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa")
    return _kernels(_asm(os.path.join(CSRC, "ksa_api.hip"), str(d / "ksa_api.s")))


def test_config2_kernel_keeps_its_single_exchange_reads_and_no_scratch_in_the_window_loop(product_asm):
    """spectrum_kernel<4096, c64, RM = 8, AVG> in its ping-pong form (round 5: the window loop holds TWO windows, the halves of the
    sample registers swap roles with the window's parity instead of being moved): 168 VGPRs = three waves per SIMD, NO spilled
    register (8 in the rolled form with the branch-free output stage, 2 before it), and per window exactly the 32 single
    ds_read_b64 of the two exchanges, four barriers, 8 sample loads (+ the 16 of a frame's first window, a rd == 0 block inside
    the loop's span) and 8 register moves per window where the rolled loop had 24 (rounds 4 and 5)."""
    body, tail = _find(product_asm, "spectrum_kernel<4096, 0, 8, 1>")
    assert _resource(tail, "NumVgprs") <= 168 and _resource(tail, "Occupancy") == 3
    assert _resource(tail, "ScratchSize") == 0, "the ping-pong kernel spills again"
    _, tail_max = _find(product_asm, "spectrum_kernel<4096, 0, 8, 2>")
    assert _resource(tail_max, "ScratchSize") <= 12, "the MAX-fold variant spills more than its documented two registers"
    loop = _mix(_window_loop(body, 32))            # two windows per trip
    assert loop["ds_read_b64"] == 64, dict(loop)
    assert loop["ds_read2_b64"] + loop["ds_read2st64_b64"] == 0, "hipcc merged exchange reads into ds_read2_b64 again"
    assert not [k for k in loop if k.startswith("scratch_")], "scratch traffic inside the window loop"
    assert loop["s_barrier"] == 8
    assert loop["buffer_load_dwordx2"] == 32, "two windows load 8 new samples each; the frame's first window its 16 (rd == 0 block)"
    assert loop["v_mov_b64_e32"] <= 16, "the halves are being moved again instead of swapping roles (8 moves per window; the rolled loop had 24)"
    # the uint8 variant of the same kernel (row A0) spills nothing either
    body8, tail8 = _find(product_asm, "spectrum_kernel<4096, 1, 8, 1>")
    assert _resource(tail8, "ScratchSize") == 0 and _resource(tail8, "Occupancy") == 3
    assert _mix(_window_loop(body8, 32))["ds_read2_b64"] == 0
    # the 75 %-overlap kernel keeps the rolled loop (RM = 4): its spills stay outside the window loop
    body4, tail4 = _find(product_asm, "spectrum_kernel<4096, 0, 4, 1>")
    assert _resource(tail4, "ScratchSize") <= 36
    assert not [k for k in _mix(_window_loop(body4, 16)) if k.startswith("scratch_")]


def test_config4_kernel_runs_four_waves_per_simd_without_scratch(product_asm):
    """spectrum64_kernel<c64, AVG> (the 8 x 8 plan of round 5): single-wave workgroups, <= 128 VGPRs (four waves per SIMD), no
    scratch, EIGHT 16-byte sample loads per round (two adjacent samples each -- the point of the plan), the one exchange read
    back by 16 single ds_read_b64; and spectrum_kernel<64, ., 0, AVG>, which uint8 input still runs, as before."""
    mixes = {}
    for w1 in ("false", "true"):       # true: the all-ones window table (the reference's default, what config 4 runs): no tap reads, no multiplies
        body, tail = _find(product_asm, "spectrum64_kernel<0, 1, %s>" % w1)
        assert _resource(tail, "NumVgprs") <= 128 and _resource(tail, "Occupancy") >= 4 and _resource(tail, "ScratchSize") == 0
        loop = mixes[w1] = _mix(_window_loop(body, 16))
        assert loop["buffer_load_dwordx4"] == 8 and loop["buffer_load_dwordx2"] == 0, dict(loop)
        assert loop["ds_read_b64"] >= 16 and loop["ds_read2_b64"] + loop["ds_read2st64_b64"] == 0, dict(loop)
    assert mixes["true"]["ds_read_b128"] == 0 and mixes["false"]["ds_read_b128"] + mixes["false"]["ds_read_b64"] > mixes["true"]["ds_read_b64"]
    valu = lambda m: sum(v for k, v in m.items() if k.startswith("v_"))
    assert valu(mixes["true"]) <= valu(mixes["false"]) - 16, "the rectangular-window kernel still multiplies by its taps"
    for name in ("spectrum_kernel<64, 0, 0, 1>", "spectrum_kernel<64, 1, 0, 1>"):
        body, tail = _find(product_asm, name)
        assert _resource(tail, "NumVgprs") <= 128 and _resource(tail, "Occupancy") >= 4 and _resource(tail, "ScratchSize") == 0
        loop = _mix(_window_loop(body, 16))
        assert loop["ds_read_b64"] == 16 and loop["ds_read2_b64"] + loop["ds_read2st64_b64"] == 0, dict(loop)


def test_config3_kernel_keeps_its_documented_spill_budget(product_asm):
    """spectrum32_kernel<16384, c64, AVG>: 256 VGPRs at two waves per SIMD, <= 3 spilled registers (16 bytes per lane): stored
    before the window loop, reloaded by at most 4 scratch_load per window, never stored inside it; the 64 exchange reads of a
    window are single ds_read_b64 (the only ds_read2_b64 left are the 15 pairs of middle-pass twiddles, a broadcast table read)
    (DESIGN.md 4.4)."""
    body, tail = _find(product_asm, "spectrum32_kernel<16384, 0, 1>")
    assert _resource(tail, "NumVgprs") == 256 and _resource(tail, "Occupancy") == 2
    assert _resource(tail, "ScratchSize") <= 16
    loop = _mix(_window_loop(body, 32))
    assert loop["ds_read_b64"] >= 64, dict(loop)
    assert loop["ds_read2_b64"] + loop["ds_read2st64_b64"] <= 15
    assert not [k for k in loop if k.startswith("scratch_store")], "a spill STORE inside the window loop"
    assert sum(v for k, v in loop.items() if k.startswith("scratch_load")) <= 4
    assert loop["buffer_load_dwordx4"] == 8 and loop["buffer_load_dwordx2"] == 32      # 8 tap loads of 16 bytes, 32 IQ loads


def test_the_guard_bites_without_the_atomic_exchange_loads(tmp_path):
    """The same config-2 kernel from a small translation unit built with -DKSA_LDS_ATOMIC_LD=0 (plain LDS loads): the
    load / store optimizer merges the exchange reads into ds_read2_b64 / ds_read2st64_b64 -- exactly what the first test
    refuses.  (If a future hipcc stops merging, this test fails and lds_ld64 can go.)"""
    src = tmp_path / "one_kernel.hip"
    src.write_text('#include "%s"\n'
                   'template __global__ void ksa::spectrum_kernel<4096, ksa::FMT_C64, 8, ksa::CUMU_AVG>(const ksa::SpecParams);\n'
                   % os.path.join(CSRC, "ksa_kernels.hpp"))
    k = _kernels(_asm(str(src), str(tmp_path / "one_kernel.s"), ["-DKSA_LDS_ATOMIC_LD=0"]))
    body, _ = _find(k, "spectrum_kernel<4096, 0, 8, 1>")
    loop = _mix(_window_loop(body, 32))
    assert loop["ds_read2_b64"] + loop["ds_read2st64_b64"] > 0 and loop["ds_read_b64"] < 64, dict(loop)
