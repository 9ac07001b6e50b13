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


@pytest.mark.parametrize("frames_per_rank,idx0", [(5, 0), (70, 100), (200, 17)])
def test_two_rank_merge_equals_sequential_run(tmp_path, frames_per_rank, idx0):
    load_pkg()
    world = 2
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
def _scan_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ksa_dist = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    n, full, steps = 64, 512, 7                                  # 7 steps over 3 ranks: shares 2/2/3
    x = orc.synth_iq(full * steps, 77).astype(np.complex64).reshape(steps, full)
    lo, hi = ksa_dist.step_range(steps, rank, world)
    win = orc.window_table("ones", n)
    local = np.array([orc.curscan(x[s], n, 0.1, win, "AVG") for s in range(lo, hi)], dtype=np.float32).reshape(hi - lo, n)
    got = ksa_dist.gather_steps(torch.from_numpy(local), steps, rank, world)
    if rank == world - 1:
        np.save(out_path, got.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_scan_band_shard_gather(tmp_path):
    load_pkg()
    ksa_dist = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    assert [ksa_dist.step_range(7, r, 3) for r in range(3)] == [(0, 2), (2, 4), (4, 7)]
    assert [ksa_dist.step_range(1226, r, 8)[1] - ksa_dist.step_range(1226, r, 8)[0] for r in range(8)] == [153, 153, 153, 154, 153, 153, 153, 154]
    out = str(tmp_path / "steps.npy")
    mp.spawn(_scan_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    n, full, steps = 64, 512, 7
    x = orc.synth_iq(full * steps, 77).astype(np.complex64).reshape(steps, full)
    want = np.array([orc.curscan(x[s], n, 0.1, orc.window_table("ones", n), "AVG") for s in range(steps)], dtype=np.float32)
    assert np.array_equal(np.load(out), want)


def _scan_batch_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ksa_dist = __import__("importlib").import_module("prgs-sdr-kspecanal_amd.distributed")
    passes, steps, n = 3, 7, 16
    full = torch.arange(passes * steps * n, dtype=torch.float32).reshape(passes, steps, n)
    lo, hi = ksa_dist.step_range(steps, rank, world)
    got = ksa_dist.gather_steps(full[:, lo:hi].contiguous(), steps, rank, world)
    if rank == 0:
        np.save(out_path, got.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_scan_band_shard_gather_batch_of_passes(tmp_path):
    """The multi-pass form: every rank holds [passes][its bands][N]; one all-gather rebuilds [passes][steps][N]."""
    load_pkg()
    out = str(tmp_path / "batch.npy")
    mp.spawn(_scan_batch_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    want = np.arange(3 * 7 * 16, dtype=np.float32).reshape(3, 7, 16)
    assert np.array_equal(np.load(out), want)
