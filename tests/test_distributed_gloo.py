"""world_size-2 CPU test (gloo) of the time-chunk sharding algebra in prgs-sdr-kspecanal_amd/distributed.py:
two ranks each own half of a run, their partials (built here from the oracle in the layout libksa's
accumulate kernels produce) are merged with the product's merge_partials / merge_ring, and the result
must equal the oracle's sequential loop over the whole run (python/kspecanal.py:464-484)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ksa_oracle as orc
from conftest import load_pkg, ROOT

N, FULL, Q, GAIN, XRES = 256, 2048, 0.5, 19.1, 64



def _partial_from_oracle(db, first_index, total, has_prev, owns_last):
    """[max, cur-or--inf, -min, sum_k 2^-(n-k+1) x_k] -- what accumulate_partial/reduce leave on a rank."""
    f, n = db.shape
    w = np.empty(f)
    for i in range(f):
        kg = first_index + i
        e = total - kg
        if kg == 0 and not has_prev:
            e = total - 1
        w[i] = 2.0 ** -e
    part = np.empty((4, n), dtype=np.float32)
    part[0] = db.max(axis=0)
    part[1] = db[-1] if owns_last else -np.inf
    part[2] = -db.min(axis=0)
    part[3] = (w[:, None] * db).sum(axis=0)
    return part


def _worker(rank, world, port, frames_per_rank, idx0, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ksa_dist = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    total = frames_per_rank * world
    x = orc.synth_iq(FULL * total, 4242).astype(np.complex64).reshape(total, FULL)
    win = orc.window_table("hanning", N)
    mine = x[rank * frames_per_rank:(rank + 1) * frames_per_rank]
    db = np.array([orc.log_no_gain(orc.curscan(fr, N, Q, win, "AVG"), GAIN) for fr in mine])
    part = torch.from_numpy(_partial_from_oracle(db, rank * frames_per_rank, total, False, rank == world - 1))
    part_local = part.clone()
    ksa_dist.merge_partials(part)
    # ring: every rank writes its rows at the globally correct slots, as the spectrum kernel does
    ring = torch.full((128, XRES), 7.0)          # stale content from "before this run"
    first = max(0, frames_per_rank - 128)
    for f in range(first, frames_per_rank):
        g = rank * frames_per_rank + f
        ring[(idx0 + g) % 128] = torch.from_numpy(orc.plotcompress(db[f], XRES, "MAX")).float()
    ring_local = ring.clone()
    ksa_dist.merge_ring(ring, idx0, frames_per_rank, world)
    # the one-collective form the GPU path uses: all-gather of [partial | ring], merged locally
    send = torch.cat([part_local.reshape(-1), ring_local.reshape(-1)])
    recv = torch.empty((world, send.numel()))
    ksa_dist.all_gather_flat(recv, send)
    part1, ring1, defined = ksa_dist.merge_gathered_reference(recv, N, XRES, idx0, frames_per_rank)
    ring1 = torch.where(defined.view(-1, 1), ring1, ring_local)
    if rank == 0:
        np.savez(out_path, part=part.numpy(), ring=ring.numpy(), part1=part1.numpy(), ring1=ring1.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,frames_per_rank,idx0", [(2, 5, 0), (2, 70, 100), (2, 200, 17), (3, 50, 9), (8, 12, 120), (8, 40, 3)])
def test_time_chunk_merge_equals_sequential_run(tmp_path, world, frames_per_rank, idx0):
    """2, 3 and 8 ranks (the node size): ring ownership when the last 128 frames span 4 ranks (8 x 40) and when the
    whole run is shorter than the ring (8 x 12)."""
    load_pkg()
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(world, _free_port(), frames_per_rank, idx0, out), nprocs=world, join=True)
    got = np.load(out)
    total = frames_per_rank * world
    x = orc.synth_iq(FULL * total, 4242).astype(np.complex64).reshape(total, FULL)
    st, db, _ = orc.zerospan_batch(x, N, Q, orc.window_table("hanning", N), "AVG", GAIN, XRES)
    assert np.allclose(got["part"][0], st.max, rtol=0, atol=1e-4)
    assert np.allclose(got["part"][1], st.cur, rtol=0, atol=1e-4)
    assert np.allclose(-got["part"][2], st.min, rtol=0, atol=1e-4)
    assert np.allclose(got["part"][3], st.avg, rtol=0, atol=1e-3)     # float32 transport of a float64 EMA
    # waterfall ring: the oracle's ring starts at index 0; rotate to the run's starting slot
    want = np.full((128, XRES), 7.0)
    for g in range(max(0, total - 128), total):
        want[(idx0 + g) % 128] = orc.plotcompress(db[g], XRES, "MAX")
    assert np.allclose(got["ring"], want, rtol=0, atol=1e-4)
    # all-gather + local merge: same maxima bit for bit, sum in rank order
    assert np.array_equal(got["part1"][0:3], got["part"][0:3])
    assert np.allclose(got["part1"][3], got["part"][3], rtol=0, atol=1e-5)
    assert np.array_equal(got["ring1"], got["ring"])


def test_ring_owner_map():
    load_pkg()
    ksa_dist = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    own = ksa_dist.ring_owner(0, 5, 2)           # 10 frames: slots 0..4 rank 0, 5..9 rank 1, rest untouched
    assert own[:5].tolist() == [0] * 5 and own[5:10].tolist() == [1] * 5 and bool((own[10:] == -1).all())
    own = ksa_dist.ring_owner(3, 200, 2)         # 400 frames: every slot holds one of the last 128 -> rank 1
    assert bool((own == 1).all())
    own = ksa_dist.ring_owner(0, 100, 2)         # last 128 of 200 frames: 72..99 rank 0, 100..199 rank 1
    slots = {int(s): int(o) for s, o in enumerate(own)}
    assert slots[72 % 128] == 1 or slots[72] == 0
    newest = {(g % 128): g // 100 for g in range(72, 200)}
    assert all(slots[s] == r for s, r in newest.items())


# ------------------------------------------------------------------------------------------ scan band shard
class FakeScanEngine:
    """CPU stand-in for SpectrumEngine in the band-sharded scan driver (test infrastructure: numpy + the oracle).
    It restates what libksa does per rank -- clipped dB spectra of its bands, the range stitch over the elements it
    owns (K:643-668) and the partial waterfall rows (K:696-697) -- so that distributed.ShardedScan's message plan
    (halo send/recv, row all-gather, gather_state) can run with 2, 3 and 8 gloo ranks without a GPU."""

    def __init__(self, n, full, q_win, window, gain, min_amp, xres, total, scan_q):
        self.fft_size, self.full_size, self.q_win, self.gain, self.min_amp = n, full, q_win, gain, min_amp
        self.win = orc.window_table(window, n)
        self.scan_total, self.scan_hop, self.scan_hm_width = total, int(n * scan_q), xres
        floor = 10 * np.log10(min_amp) - gain
        self.state = np.stack([np.full(total, floor), np.full(total, floor), np.full(total, -gain), np.full(total, floor)])
        self.hm = np.full((128, xres), min_amp)
        self.hm_index, self.passes, self._rows = 0, 0, None

    def scan_shard(self, nsteps, rank, world):
        return load_pkg().SpectrumEngine.scan_shard(self, nsteps, rank, world)

    def set_stream(self, s):
        pass

    def scan_spectra_dev(self, iq, fmt, nframes, out, step_ok=None, frame_stride=None):
        x = iq.numpy().reshape(nframes, self.full_size, 2)
        o = out.view(-1, self.fft_size)
        for f in range(nframes):
            lin = orc.curscan(x[f, :, 0] + 1j * x[f, :, 1], self.fft_size, self.q_win, self.win, "AVG")
            if step_ok is not None and not step_ok[f]:
                lin = np.ones(self.fft_size)                              # dummy band, K:637-639
            o[f] = torch.from_numpy(orc.log_no_gain(orc.clip2minamp(lin, self.min_amp), self.gain, inf_to=0).astype(np.float32))

    def scan_stitch_range_dev(self, own, halo, nhalo, lo, hi, nsteps, npasses, e_lo, e_hi, own_band_major=False):
        n, hop, tot = self.fft_size, self.scan_hop, self.scan_total
        own = None if own is None else own.numpy().astype(np.float64)
        halo = None if halo is None else halo.numpy().astype(np.float64)
        band = lambda ps, i: (own[i - lo, ps] if own_band_major else own[ps, i - lo]) if i >= lo else halo[i - (lo - nhalo), ps]
        rows = min(npasses, 128)
        self._rows = np.full((rows, self.scan_hm_width), -np.inf)
        g = tot // self.scan_hm_width
        for ps in range(npasses):
            for e in range(e_lo, e_hi):
                i0 = 0 if e - n + 1 <= 0 else (e - n + hop) // hop
                i1 = min(e // hop, nsteps - 1)
                if i0 > i1:
                    continue
                c = band(ps, i0)[e - i0 * hop]
                for i in range(i0 + 1, i1 + 1):
                    c = (c + band(ps, i)[e - i * hop]) * 0.5
                cur, mx, mn, av = self.state[:, e]
                first = self.passes == 0 and ps == 0
                self.state[:, e] = (c, max(mx, c), min(mn, c), c if first else (av + c) * 0.5)
            if ps >= npasses - rows and e_hi > e_lo:
                r = ps - (npasses - rows)
                for cell in range(self.scan_hm_width):
                    a, b = max(cell * g, e_lo), min((cell + 1) * g, e_hi)
                    if b > a:
                        self._rows[r, cell] = self.state[3, a:b].max()
        self.passes += npasses
        self._npasses = npasses

    def scan_rows(self):
        return torch.from_numpy(self._rows.astype(np.float32))

    def scan_merge_rows(self, gathered, world, rows, npasses):
        m = gathered.numpy().reshape(world, rows, self.scan_hm_width).max(axis=0)
        for r in range(rows):
            self.hm[(self.hm_index + npasses - rows + r) % 128] = m[r]
        self.hm_index = (self.hm_index + npasses) % 128

    def scan_state(self):
        out = {k: self.state[i].copy() for i, k in enumerate(("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg"))}
        out.update(fftHM=self.hm.copy(), hm_index=self.hm_index, passes=self.passes)
        return out


SCAN_CASES = {   # name: n, scanRangeNonOverlap, bands of 2.4 MHz, passes per batch
    "half": (64, 0.5, 9, 3),        # 18 bands like fmScan: shares of 2-3 at 8 ranks
    "eighth": (64, 0.125, 2, 2),    # halo of 7 bands: spans several ranks at 8 ranks
    "whole": (32, 1.0, 11, 2),      # hop == N: no halo at all
    "tiny": (32, 0.5, 2, 2),        # 4 bands over 8 ranks: half of the ranks own nothing
}


def _scan_case(name):
    n, q, bands, passes = SCAN_CASES[name]
    fs, start = 2.4e6, 100e6
    end = start + bands * fs
    steps = len(orc.scan_steps(start, end, fs, q))
    full = 8 * n
    x = orc.synth_iq(full * steps * passes * 2, 77).astype(np.complex64).reshape(2, passes, steps, full)
    return n, q, fs, start, end, steps, full, passes, x


def _bad_steps(steps, passes):
    """Tunes that fail in the dummy-band variant of a case: [2][passes][steps] flags, 0 = failed."""
    ok = np.ones((2, passes, steps), dtype=np.uint8)
    ok[0, 0, steps // 2] = 0
    ok[1, passes - 1, 0] = 0
    ok[1, 0, steps - 1] = 0
    return ok


def _sharded_scan_worker(rank, world, port, name, out_path, members=None, dummy=False):
    """members: run the scan on a SUB-GROUP of the job (its ranks, in group order); the other ranks only take part in
    creating the group.  Peers inside distributed.py are ranks of the group, not of the job."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ksa_dist = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    group, grank, gworld = None, rank, world
    if members is not None:
        group = dist.new_group(list(members))
        if rank not in members:
            dist.barrier()
            dist.destroy_process_group()
            return
        grank, gworld = dist.get_rank(group), dist.get_world_size(group)     # (new_group orders its members by global rank)
    n, q, fs, start, end, steps, full, passes, x = _scan_case(name)
    total = int((end - start) / fs) * n
    eng = FakeScanEngine(n, full, 0.5, "hanning", GAIN, 1e-7, 32, total, q)
    run = ksa_dist.ShardedScan(eng, grank, gworld, group=group, device="cpu")
    lo, hi = ksa_dist.step_range(steps, grank, gworld)
    bad = _bad_steps(steps, passes) if dummy else None
    for batch in range(2):
        iq = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(x[batch][:, lo:hi]))).contiguous()
        run.run_passes(iq, 0, steps, passes, step_ok=None if bad is None else bad[batch][:, lo:hi])
    st = run.gather_state(steps)
    cb = run.collective_bytes()
    assert cb["halo_recv"] == sum(passes * (n - c0) * 4 for _, _, c0 in ksa_dist.halo_plan(steps, gworld, n, eng.scan_hop)[grank]["recv"])
    np.savez(out_path % rank, hm_index=st["hm_index"], **{k: st[k] for k in ("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg", "fftHM")})
    if members is not None:
        # the time-chunk merges on the same sub-group: the broadcast of merge_ring (>= 128 frames per rank) addresses
        # its source by global rank too
        part = torch.full((4, 8), float(grank))
        ksa_dist.merge_partials(part, group)
        assert part[0, 0] == gworld - 1 and part[3, 0] == sum(range(gworld))
        ring = torch.full((128, 4), float(grank))
        ksa_dist.merge_ring(ring, 0, 130, gworld, group)
        assert bool((ring == gworld - 1).all())
        ring = torch.full((128, 4), float(grank))
        ksa_dist.merge_ring(ring, 0, 3, gworld, group)
        assert ring[:3 * gworld, 0].tolist() == [float(r) for r in range(gworld) for _ in range(3)]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("half", 2), ("half", 3), ("half", 8), ("eighth", 8), ("whole", 3), ("eighth", 2), ("tiny", 8)])
def test_band_sharded_scan_driver_over_gloo(tmp_path, name, world):
    """distributed.ShardedScan with 2 / 3 / 8 ranks on CPU (gloo) over an engine stand-in: after two batches every
    rank's gathered curves and waterfall ring equal the oracle's sequential scan (K:621-668, K:696-697, K:732)."""
    load_pkg()
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_sharded_scan_worker, args=(world, _free_port(), name, out), nprocs=world, join=True)
    n, q, fs, start, end, steps, full, passes, x = _scan_case(name)
    ref = orc.ScanState(n, start, end, fs, GAIN, 1e-7, 32, scan_non_overlap=q)
    win = orc.window_table("hanning", n)
    for batch in range(2):
        for p in range(passes):
            ref.run_pass([orc.curscan(x[batch, p, s], n, 0.5, win, "AVG") for s in range(steps)])
    for r in range(world):
        got = np.load(out % r)
        assert int(got["hm_index"]) == ref.hm_index
        for k, want in (("Fft.Cur", ref.cur), ("Fft.Max", ref.max), ("Fft.Min", ref.min), ("Fft.Avg", ref.avg)):
            assert np.max(np.abs(got[k] - want)) < 2e-4, (k, r)          # float32 transport of the dB spectra
        assert np.max(np.abs(got["fftHM"][:2 * passes] - ref.hm[:2 * passes])) < 2e-4, r
        assert np.allclose(got["fftHM"][2 * passes:], ref.hm[2 * passes:])


@pytest.mark.parametrize("name,world,members,dummy", [("half", 4, (1, 3), False), ("eighth", 5, (4, 0, 2), False), ("half", 3, None, True)])
def test_band_sharded_scan_on_a_sub_group_and_with_failed_tunes(tmp_path, name, world, members, dummy):
    """halo peers are ranks INSIDE the group the driver was given; isend / irecv / broadcast want global
    ranks -- a job of 4 (5) ranks runs the scan on the sub-group [1, 3] ([4, 0, 2]: group order differs from job order).
    And a scan with failed tunes (step_ok): the dummy band (K:637-639) is written on the engine's side of the driver."""
    load_pkg()
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_sharded_scan_worker, args=(world, _free_port(), name, out, members, dummy), nprocs=world, join=True)
    n, q, fs, start, end, steps, full, passes, x = _scan_case(name)
    ref = orc.ScanState(n, start, end, fs, GAIN, 1e-7, 32, scan_non_overlap=q)
    win = orc.window_table("hanning", n)
    bad = _bad_steps(steps, passes) if dummy else np.ones((2, passes, steps), dtype=np.uint8)
    for batch in range(2):
        for p in range(passes):
            ref.run_pass([orc.curscan(x[batch, p, s], n, 0.5, win, "AVG") if bad[batch, p, s] else None for s in range(steps)])
    for r in (members if members is not None else range(world)):
        got = np.load(out % r)
        assert int(got["hm_index"]) == ref.hm_index
        for k, want in (("Fft.Cur", ref.cur), ("Fft.Max", ref.max), ("Fft.Min", ref.min), ("Fft.Avg", ref.avg)):
            assert np.max(np.abs(got[k] - want)) < 2e-4, (k, r)
        assert np.max(np.abs(got["fftHM"][:2 * passes] - ref.hm[:2 * passes])) < 2e-4, r


def test_scan_band_shares_and_halo_plan():
    load_pkg()
    ksa_dist = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    assert [ksa_dist.step_range(7, r, 3) for r in range(3)] == [(0, 2), (2, 4), (4, 7)]
    assert [ksa_dist.step_range(1226, r, 8)[1] - ksa_dist.step_range(1226, r, 8)[0] for r in range(8)] == [153, 153, 153, 154, 153, 153, 153, 154]
    assert [ksa_dist.step_range(18, r, 8)[1] - ksa_dist.step_range(18, r, 8)[0] for r in range(8)] == [2, 2, 2, 3, 2, 2, 2, 3]
    # fmScan at 8 ranks: every rank but the first receives the upper half of ONE band from its left neighbour
    plan = ksa_dist.halo_plan(18, 8, 16384, 8192)
    assert plan[0]["recv"] == [] and plan[7]["send"] == []
    for r in range(1, 8):
        (src, band, col0), = plan[r]["recv"]
        assert src == r - 1 and band == ksa_dist.step_range(18, r, 8)[0] - 1 and col0 == 8192
    # every receive has its matching send, in the same order per pair of ranks
    for nsteps, world, n, hop in ((18, 8, 16384, 8192), (17, 8, 64, 8), (7, 3, 64, 16), (3, 8, 64, 32)):
        plan = ksa_dist.halo_plan(nsteps, world, n, hop)
        for r in range(world):
            for src in range(world):
                want = [(j, c) for s_, j, c in plan[r]["recv"] if s_ == src]
                have = [(j, c) for d_, j, c in plan[src]["send"] if d_ == r]
                assert want == have
