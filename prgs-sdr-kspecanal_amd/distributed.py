"""Time-chunk sharding of a zeroSpan run over the GPUs of one node (one process per GPU, RCCL).

The reference's frame loop (python/kspecanal.py:460-484) is sequential only through its accumulators.
Frames are independent up to the dB spectrum, so rank r takes frames [r*F, (r+1)*F) of a run of
G*F frames and the accumulators are merged with the algebra of SURVEY.md 8(e):

  Max, Min  -> elementwise all-reduce MAX (Min travels negated in the same buffer)
  Avg       -> the (a+x)/2 recursion (K:137-139) in closed form is sum_k 2^-(n-k+1) x_k; every rank sums
               its frames with their GLOBAL weights (libksa does that given first_index/total_frames), then
               one all-reduce SUM
  Cur       -> the globally last frame: -inf on every other rank, merged by the MAX all-reduce
  waterfall -> the ring keeps the last 128 rows of the run; every rank writes its rows at the globally
               correct ring slots, one all-gather picks each slot from the rank that owns its newest frame.

Message sizes are a few KiB..MiB (latency bound on xGMI), so there is ONE collective per batch: every rank's
partial block and ring are one contiguous device block (4N + 128W floats, 320 KiB at config 2), all-gathered
in rank order; libksa's merge kernel then reduces and commits locally (ksa_merge_gathered_dev).  Summing in
rank order makes the result bit-identical on every rank and from run to run, which an all-reduce does not
promise.  merge_partials/merge_ring are the same algebra as separate all-reduces on plain tensors
(device agnostic: tested on CPU with gloo), and merge_gathered_reference restates the kernel in torch.
"""
import numpy as np
import torch
import torch.distributed as dist

HM_ROWS = 128


def global_rank(group, rank_in_group):
    """dist.broadcast / isend / irecv / P2POp address peers by their rank in the DEFAULT group even when `group=` is
    given; everything in this module computes ranks inside the group it was handed."""
    return rank_in_group if group is None else dist.get_global_rank(group, rank_in_group)


def merge_partials(partial, group=None):
    """partial: float32[4, N] = {max, cur-or--inf, -min, weighted sum} of this rank's chunk, in place.
    The minimum travels negated so that ONE MAX all-reduce covers three rows; one SUM covers the fourth."""
    dist.all_reduce(partial[0:3], op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(partial[3], op=dist.ReduceOp.SUM, group=group)
    return partial


def ring_owner(idx0, frames_per_rank, world):
    """For every ring slot: the rank whose chunk holds the newest frame stored there, or -1 if no frame of
    this run maps to the slot within the last 128 frames.  Global frame g lives in slot (idx0+g) % 128."""
    total = frames_per_rank * world
    slots = torch.arange(HM_ROWS)
    # newest g < total with (idx0 + g) % 128 == slot
    last = total - 1
    back = (idx0 + last - slots) % HM_ROWS
    g = last - back
    owner = torch.where(g >= 0, g // max(frames_per_rank, 1), torch.full_like(g, -1))
    return owner


def merge_ring(ring, idx0, frames_per_rank, world, group=None):
    """ring: float32[128, W] of this rank (rows written at globally correct slots), merged in place."""
    if frames_per_rank >= HM_ROWS:
        dist.broadcast(ring, src=global_rank(group, world - 1), group=group)   # the last 128 frames all live on the last rank
        return ring
    gathered = [torch.empty_like(ring) for _ in range(world)]
    dist.all_gather(gathered, ring.contiguous(), group=group)
    owner = ring_owner(idx0, frames_per_rank, world).to(ring.device)
    stack = torch.stack(gathered)                       # [world, 128, W]
    pick = owner.clamp(min=0).view(1, HM_ROWS, 1).expand(1, HM_ROWS, ring.shape[1])
    merged = torch.gather(stack, 0, pick)[0]
    keep = (owner < 0).view(HM_ROWS, 1)
    ring.copy_(torch.where(keep, ring, merged))
    return ring


def all_gather_flat(recv, send, group=None):
    """recv: [world, n] <- every rank's send [n], rank order; one collective."""
    try:
        dist.all_gather_into_tensor(recv.view(-1), send, group=group)
    except (RuntimeError, NotImplementedError, AttributeError):   # backends without the flat form
        dist.all_gather(list(recv.unbind(0)), send, group=group)
    return recv


def merge_gathered_reference(gathered, n, hm_w, idx0, frames_per_rank):
    """Torch restatement of ksa::merge_gathered_kernel: gathered [world, 4n + 128*hm_w] -> (partial [4, n],
    ring rows [128, hm_w] with a mask of the rows this run defines)."""
    world = gathered.shape[0]
    parts = gathered[:, :4 * n].reshape(world, 4, n)
    partial = torch.empty((4, n), dtype=gathered.dtype, device=gathered.device)
    partial[0:3] = parts[:, 0:3].amax(dim=0)            # amax propagates NaN, as the kernel's nan_max does
    acc = parts[0, 3].clone()
    for r in range(1, world):
        acc = acc + parts[r, 3]                         # rank order
    partial[3] = acc
    rings = gathered[:, 4 * n:].reshape(world, HM_ROWS, hm_w)
    owner = ring_owner(idx0, frames_per_rank, world).to(gathered.device)
    pick = owner.clamp(min=0).view(1, HM_ROWS, 1).expand(1, HM_ROWS, hm_w)
    ring = torch.gather(rings, 0, pick)[0]
    return partial, ring, owner >= 0


class ShardedZeroSpan:
    """Drives one engine per rank; with world == 1 it is a plain frames_dev call on the engine's own stream.  Stream
    contract of the collective path: every step points the engine at torch's current stream (_follow_current_stream),
    the stream RCCL orders against.  `group`: ranks are counted INSIDE it (rank / world = position in / size of the group)."""

    def __init__(self, engine, rank=0, world=1, group=None, always_collective=False):
        self.eng, self.rank, self.world, self.group = engine, rank, world, group
        self.hm_index = 0                              # global ring position (identical on all ranks)
        self.collective = world > 1 or always_collective   # always_collective: run the merge path on one rank too
        if self.collective:
            self._send = torch.as_tensor(engine.exchange(), device="cuda")
            self._recv = torch.empty((world, self._send.numel()), dtype=torch.float32, device="cuda")

    def step(self, iq, fmt, frames, cur_db=None, hm_rows=None):
        """Every rank passes its own `frames` capture blocks (its time chunk of a world*frames run)."""
        eng = self.eng
        if not self.collective:
            eng.frames_dev(iq, fmt, frames, cur_db=cur_db, hm_rows=hm_rows)
            self.hm_index = (self.hm_index + frames) % HM_ROWS
            return
        _follow_current_stream(eng)
        total = frames * self.world
        eng.set_hm_index((self.hm_index + self.rank * frames) % HM_ROWS)
        eng.frames_dev(iq, fmt, frames, first_index=self.rank * frames, total_frames=total,
                       cur_db=cur_db, hm_rows=hm_rows, commit=False)
        if self.world > 1 or dist.is_initialized():     # (a one-rank group still exercises the RCCL call)
            all_gather_flat(self._recv, self._send, self.group)
            gathered = self._recv
        else:
            gathered = self._send                       # always_collective on one rank: the gather is the identity
        eng.merge_gathered(gathered, self.world, frames, self.hm_index)
        self.hm_index = (self.hm_index + total) % HM_ROWS


# ------------------------------------------------------------------------------------------------
# Frequency-band sharding of a scan (BASELINE configs 3 and 4)
def step_range(nsteps, rank, world):
    """Contiguous, balanced share of the tuned bands of one pass: [lo, hi)."""
    lo = (nsteps * rank) // world
    hi = (nsteps * (rank + 1)) // world
    return lo, hi


def halo_plan(nsteps, world, fft_size, hop):
    """Who needs which band.  Rank r owns bands [lo_r, hi_r) and the stitched elements from lo_r*hop on; the
    reference averages every band's lower part with the upper part of the bands before it (K:643-650), so rank r also
    needs the nhalo = ceil(N/hop) - 1 bands in front of lo_r (1 at the usual hop of N/2) -- and of band j only the
    columns from (lo_r - j)*hop on.  Returns per rank {'recv': [(src, band, col0)], 'send': [(dst, band, col0)]}."""
    reach = -(-fft_size // hop) - 1
    owner = []
    for r in range(world):
        lo, hi = step_range(nsteps, r, world)
        owner += [r] * (hi - lo)
    plan = [{"recv": [], "send": []} for _ in range(world)]
    for r in range(world):
        lo, _ = step_range(nsteps, r, world)
        for j in range(max(0, lo - reach), lo):
            col0 = (lo - j) * hop
            plan[r]["recv"].append((owner[j], j, col0))
            plan[owner[j]]["send"].append((r, j, col0))
    return plan


def _follow_current_stream(eng):
    """torch.distributed collectives are ordered against torch's CURRENT stream; libksa launches on the engine's
    stream.  Both must be the same stream, or a collective could read a block before the kernels that fill it have
    run.  The sharded drivers therefore point the engine at torch's current stream before a step that communicates
    (ksa_set_stream orders the new stream behind work still queued on the old one); a step without a collective
    leaves the engine's stream alone."""
    if torch.cuda.is_available():
        eng.set_stream(torch.cuda.current_stream().cuda_stream)


class _P2P:
    """One halo exchange in flight: every receive and send is posted at construction, finish() waits for all of them.
    ops_*: [(tensor, peer)], peer = rank INSIDE `group` (translated to the global rank the P2P calls want).  NCCL (= RCCL) moves device tensors directly and asynchronously (one ncclGroupStart/End:
    the transfers run on RCCL's stream behind the work already queued on the current stream, kernels launched afterwards
    overlap them; finish() makes the current stream wait); gloo (CPU rehearsals, several ranks on one GPU) goes through
    host copies of device tensors."""

    def __init__(self, ops_send, ops_recv, group=None):
        self.staged, self.reqs = [], []
        if not ops_send and not ops_recv:
            return
        ops_send = [(t, global_rank(group, peer)) for t, peer in ops_send]
        ops_recv = [(t, global_rank(group, peer)) for t, peer in ops_recv]
        if dist.get_backend(group) == "nccl":
            ops = [dist.P2POp(dist.irecv, t, peer, group) for t, peer in ops_recv]
            ops += [dist.P2POp(dist.isend, t, peer, group) for t, peer in ops_send]
            self.reqs = dist.batch_isend_irecv(ops)
            return
        for t, peer in ops_recv:
            buf = torch.empty(t.shape, dtype=t.dtype) if t.is_cuda else t
            self.staged.append((t, buf))
            self.reqs.append(dist.irecv(buf, src=peer, group=group))
        for t, peer in ops_send:
            self.reqs.append(dist.isend(t.cpu() if t.is_cuda else t, dst=peer, group=group))

    def finish(self):
        for r in self.reqs:
            r.wait()
        for t, buf in self.staged:
            if buf is not t:
                t.copy_(buf)
        self.reqs, self.staged = [], []


class ShardedScan:
    """Scan passes over G GPUs, sharded by the STITCHED RANGE: rank r transforms the tuned bands [lo, hi) of every
    pass (each band has its own IQ capture, K:636-641) and owns the elements [lo*hop, hi*hop) of Fft.Cur/Max/Min/Avg
    (the last rank up to totalEntries).  Per batch of passes it receives from its left neighbour(s) the overlap part
    of the band(s) in front of its first one (halo send/recv: (N - hop) floats per pass at the usual hop of N/2 --
    32 KiB at fmScan), stitches and accumulates its own elements only (ksa_scan_stitch_range_dev), and ONE all-gather of
    the partial waterfall rows ([min(passes,128)][W] floats per rank) completes the ring on every rank
    (ksa_scan_merge_rows_dev).  The curves stay sharded until someone reads them (gather_state).  Buffers are
    allocated once per (nsteps, npasses).  With few bands per rank (<= BAND_MAJOR_MAX: fmScan at 8 ranks has 2-3) every band
    is its own strided spectrum launch and the bands a neighbour waits for are transformed first, so that the halo travels
    while the remaining bands are transformed (RCCL send/recv run asynchronously beside the engine's stream)."""

    def __init__(self, engine, rank=0, world=1, group=None, device=None):
        self.eng, self.rank, self.world, self.group = engine, rank, world, group
        self.device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self._key = None

    BAND_MAJOR_MAX = 8     # up to this many own bands: one strided spectrum launch per band, boundary bands first

    def _setup(self, nsteps, npasses):
        eng = self.eng
        self._key = (nsteps, npasses)
        self.lo, self.hi, self.nhalo, self.e_lo, self.e_hi = eng.scan_shard(nsteps, self.rank, self.world)
        mine, n = self.hi - self.lo, eng.fft_size
        f32 = dict(dtype=torch.float32, device=self.device)
        # few bands per rank (fmScan at 8 ranks: 2-3): the own block is band major, every band is one strided launch over the
        # passes, the bands a neighbour waits for go first and travel while the others are transformed.  Many bands per rank
        # (quickFullScan: 153): one launch over everything, pass major; the halo is a sliver of it.
        self.band_major = self.world > 1 and 0 < mine <= self.BAND_MAJOR_MAX
        self.own = torch.empty((max(mine, 1), npasses, n) if self.band_major else (npasses, max(mine, 1), n), **f32)
        self.halo = torch.zeros((max(self.nhalo, 1), npasses, n), **f32)          # band major
        plan = halo_plan(nsteps, self.world, n, eng.scan_hop)[self.rank]
        self.recv = [(src, j, c0, torch.empty((npasses, n - c0), **f32)) for src, j, c0 in plan["recv"]]
        self.send = [(dst, j, c0, torch.empty((npasses, n - c0), **f32)) for dst, j, c0 in plan["send"]]
        self.send_bands = sorted({j for _, j, _, _ in self.send})
        self.rows = min(npasses, HM_ROWS)
        self.rows_all = torch.empty((self.world, self.rows * eng.scan_hm_width), **f32)
        self.halo_bytes_in = sum(t.numel() * 4 for *_, t in self.recv)
        self.halo_bytes_out = sum(t.numel() * 4 for *_, t in self.send)

    def collective_bytes(self):
        """Bytes this rank sends / receives per batch: halo P2P and its share of the row all-gather."""
        row = self.rows * self.eng.scan_hm_width * 4
        return {"halo_send": self.halo_bytes_out, "halo_recv": self.halo_bytes_in, "rows_allgather_send": row,
                "rows_allgather_recv": row * (self.world - 1)}

    def run_pass(self, iq_local, fmt, nsteps, step_ok=None):
        """iq_local: this rank's capture blocks, [hi-lo] frames of fullSize samples, on its GPU."""
        self.run_passes(iq_local, fmt, nsteps, 1, step_ok)

    def run_passes(self, iq_local, fmt, nsteps, npasses, step_ok=None):
        """iq_local: [npasses][hi-lo] capture blocks of this rank (pass-major), on its GPU.  step_ok (optional):
        [npasses][hi-lo], 0 marks a band whose tune failed -> dummy ones (K:637-639)."""
        eng = self.eng
        if self._key != (nsteps, npasses):
            self._setup(nsteps, npasses)
        mine, n = self.hi - self.lo, eng.fft_size
        if self.world > 1:
            _follow_current_stream(eng)      # the halo copies and the collectives below are torch work on engine output
        # a band whose tune failed becomes the dummy band (K:637-641): written by the library on the engine's stream
        # (ksa_scan_spectra_dev), so no torch kernel touches engine output outside the collective path
        ok = None
        if step_ok is not None and mine > 0:
            ok = np.ascontiguousarray(np.asarray(step_ok).reshape(npasses, mine) != 0, dtype=np.uint8)
            ok = None if bool(ok.all()) else ok
        if self.world == 1:
            if mine > 0:
                eng.scan_spectra_dev(iq_local, fmt, npasses * mine, self.own, step_ok=ok)
            eng.scan_stitch_dev(self.own, nsteps, npasses)
            return

        def post_halo():
            for _, j, c0, buf in self.send:
                buf.copy_(self.own[j - self.lo, :, c0:] if self.band_major else self.own[:, j - self.lo, c0:])
            return _P2P([(buf, dst) for dst, _, _, buf in self.send], [(buf, src) for src, _, _, buf in self.recv], self.group)

        if self.band_major:
            blocks = iq_local.reshape(npasses, mine, -1)
            first = [b for b in range(mine) if self.lo + b in self.send_bands]
            rest = [b for b in range(mine) if self.lo + b not in self.send_bands]
            xchg = None
            for b in first + rest:
                if xchg is None and b in rest:
                    xchg = post_halo()                 # everything a neighbour waits for is queued: it travels under the rest
                eng.scan_spectra_dev(blocks[:, b], fmt, npasses, self.own[b], step_ok=None if ok is None else ok[:, b],
                                     frame_stride=mine * eng.full_size)
            if xchg is None:
                xchg = post_halo()
        else:
            if mine > 0:
                eng.scan_spectra_dev(iq_local, fmt, npasses * mine, self.own, step_ok=ok)
            xchg = post_halo()
        xchg.finish()
        for _, j, c0, buf in self.recv:
            self.halo[j - (self.lo - self.nhalo), :, c0:] = buf
        eng.scan_stitch_range_dev(self.own if mine > 0 else None, self.halo if self.nhalo > 0 else None, self.nhalo,
                                  self.lo, self.hi, nsteps, npasses, self.e_lo, self.e_hi, own_band_major=self.band_major)
        rows = torch.as_tensor(eng.scan_rows(), device=self.device).reshape(-1)
        all_gather_flat(self.rows_all, rows, self.group)
        eng.scan_merge_rows(self.rows_all, self.world, self.rows, npasses)

    def gather_state(self, nsteps):
        """The plotting hand-off on every rank: the four curves assembled from the ranks' slices (one all-gather of
        the padded slices, on demand -- not part of a step) + the complete waterfall ring."""
        eng = self.eng
        st = eng.scan_state()
        if self.world == 1:
            return st
        if self._key is None or self._key[0] != nsteps:
            self._setup(nsteps, self._key[1] if self._key else 1)
        shards = [eng.scan_shard(nsteps, r, self.world) for r in range(self.world)]
        most = max(s[4] - s[3] for s in shards)
        mine = torch.zeros((4, most), dtype=torch.float64)
        for k, name in enumerate(("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg")):
            mine[k, :self.e_hi - self.e_lo] = torch.from_numpy(st[name][self.e_lo:self.e_hi])
        send = mine.to(self.device) if dist.get_backend(self.group) == "nccl" else mine
        every = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(every, send, group=self.group)
        for r, (_, _, _, e_lo, e_hi) in enumerate(shards):
            for k, name in enumerate(("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg")):
                st[name][e_lo:e_hi] = every[r][k, :e_hi - e_lo].cpu().numpy()
        return st
