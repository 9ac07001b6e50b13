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
import torch
import torch.distributed as dist

HM_ROWS = 128


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
        dist.broadcast(ring, src=world - 1, group=group)   # the last 128 frames all live on the last rank
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
    """Drives one engine per rank; with world == 1 it is a plain frames_dev call.  Stream contract: every step
    re-points the engine at torch's current stream (_follow_current_stream), the stream RCCL orders against."""

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
        _follow_current_stream(eng)
        if not self.collective:
            eng.frames_dev(iq, fmt, frames, cur_db=cur_db, hm_rows=hm_rows)
            self.hm_index = (self.hm_index + frames) % HM_ROWS
            return
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
# Frequency-band sharding of a scan pass (BASELINE config 4)
def step_range(nsteps, rank, world):
    """Contiguous, balanced share of the tuned bands of one pass: [lo, hi)."""
    lo = (nsteps * rank) // world
    hi = (nsteps * (rank + 1)) // world
    return lo, hi


def gather_steps(local, nsteps, rank, world, group=None):
    """local: float32[hi-lo, N] per-step spectra of this rank's bands (or [P, hi-lo, N] for a batch of P passes)
    -> float32[nsteps, N] ([P, nsteps, N]) on every rank.  Shares differ by at most one step, so every rank pads
    to the largest share and ONE all-gather moves the whole batch."""
    if world == 1:
        return local
    squeeze = local.dim() == 2
    if squeeze:
        local = local.unsqueeze(0)
    npasses, mine, n = local.shape
    most = max(step_range(nsteps, r, world)[1] - step_range(nsteps, r, world)[0] for r in range(world))
    pad = torch.zeros((npasses, most, n), dtype=local.dtype, device=local.device)
    pad[:, :mine] = local
    recv = torch.empty((world,) + tuple(pad.shape), dtype=local.dtype, device=local.device)
    all_gather_flat(recv.view(world, -1), pad.view(-1), group)
    out = torch.empty((npasses, nsteps, n), dtype=local.dtype, device=local.device)
    for r in range(world):
        lo, hi = step_range(nsteps, r, world)
        out[:, lo:hi] = recv[r, :, :hi - lo]
    return out[0] if squeeze else out


def _follow_current_stream(eng):
    """torch.distributed collectives are ordered against torch's CURRENT stream; libksa launches on the engine's
    stream.  Both must be the same stream, or the all-gather could read the exchange block before the kernels that
    fill it have run (and the merge could read the receive buffer before the gather lands).  The sharded drivers
    therefore re-point the engine at torch's current stream at every call -- callers may switch streams freely."""
    eng.set_stream(torch.cuda.current_stream().cuda_stream)


class ShardedScan:
    """Scan passes over G GPUs: the steps (tuned bands, each with its own IQ capture) are independent up to
    their dB spectrum (K:636-641), so rank r transforms steps [lo, hi) of every pass; one all-gather of
    float32[passes][steps][N] per batch (153 KiB per pass at quickFullScan, 1.1 MiB at fmScan) hands every rank the
    whole batch and each rank runs the stitch + Max/Min/Avg + waterfall-row kernels on it (K:643-668, K:696-697)
    -- identical state everywhere, no second collective.  Stream contract: see _follow_current_stream."""

    def __init__(self, engine, rank=0, world=1, group=None):
        self.eng, self.rank, self.world, self.group = engine, rank, world, group

    def run_pass(self, iq_local, fmt, nsteps):
        """iq_local: this rank's capture blocks, [hi-lo] frames of fullSize samples, on its GPU."""
        self.run_passes(iq_local, fmt, nsteps, 1)

    def run_passes(self, iq_local, fmt, nsteps, npasses):
        """iq_local: [npasses][hi-lo] capture blocks of this rank (pass-major), on its GPU."""
        from ._lib import OUT_DB_CLIP
        eng = self.eng
        _follow_current_stream(eng)
        lo, hi = step_range(nsteps, self.rank, self.world)
        mine = hi - lo
        local = torch.empty((npasses, max(mine, 1), eng.fft_size), dtype=torch.float32, device="cuda")
        if mine > 0:
            eng.curscan_dev(iq_local, fmt, npasses * mine, local, out_mode=OUT_DB_CLIP)
        full = gather_steps(local[:, :mine], nsteps, self.rank, self.world, self.group)
        eng.scan_stitch_dev(full.contiguous(), nsteps, npasses)
