"""Host side of the spectrum engine: reference-compatible geometry + the ctypes calls.

Everything numeric happens in libksa.so on the GPU.  What stays here is what the reference also
does once per run on the host (python/kspecanal.py, "K:"): the capture-block size (K:926-929), the
window table and its amplitude compensation (K:932-936, K:373), the window start offsets with the
reference's float64 truncation (K:368, K:386-390) and the waterfall width rule (K:449-455).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check, Config, CUMU, FMT_C64, FMT_U8, OUT_LINEAR, OUT_DB, OUT_DB_CLIP, HM_ROWS, KsaError

MIN_AMP_DEFAULT = (1 / 256) * 0.00001   # gMinAmp4Clip K:53
FFT2FULL_LESS, FFT2FULL_MORE = 8, 2     # K:49-50


def full_size_for(fft_size, sampling_rate):
    """K:926-929."""
    return fft_size * FFT2FULL_LESS if fft_size < (sampling_rate // 8) else fft_size * FFT2FULL_MORE


def window_starts(full_size, fft_size, non_overlap):
    """K:368 + K:386-390, evaluated exactly like the reference (float64 product, int() truncation)."""
    num_loops = int(full_size / (fft_size * non_overlap))
    starts = []
    for i in range(num_loops):
        s = int(i * fft_size * non_overlap)
        if s + fft_size > full_size:
            break
        starts.append(s)
    if not starts:
        raise KsaError("no complete window fits: fullSize %d fftSize %d" % (full_size, fft_size))
    return np.asarray(starts, dtype=np.int32)


def window_table(name, n):
    """K:932-935; accepts the CLI spelling or the 'WIN.X' dict key of the reference."""
    key = name.upper().replace("WIN.", "")
    if key == "ONES":
        return np.ones(n)
    if key == "HANNING":
        return np.hanning(n)
    if key == "HAMMING":
        return np.hamming(n)
    if key == "KAISER":
        return np.kaiser(n, 64)
    raise KsaError("unknown window [%s]" % name)


def heatmap_width(fft_size, xres):
    """K:449-455 with pltCompressHM = MAX (K:67)."""
    return xres if fft_size > xres else fft_size


def _ptr(a):
    """Device/host address of a torch tensor, numpy array, ctypes pointer or int."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return a


class DevArray:
    """View of library-owned device memory for torch.as_tensor (via __cuda_array_interface__)."""

    def __init__(self, ptr, shape, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class SpectrumEngine:
    """One GPU's engine for one (fftSize, window, overlap, mode) configuration."""

    def __init__(self, fft_size, full_size=None, sampling_rate=2.4e6, non_overlap=0.1, window="ones",
                 cumu_mode="AVG", gain=19.1, min_amp=MIN_AMP_DEFAULT, xres=512, max_frames=1, device=0,
                 scan_total_entries=0, scan_non_overlap=0.5, scan_xres=None,
                 u8_offset=127.5, u8_scale=127.5, stream=None):
        self.fft_size = int(fft_size)
        self.full_size = int(full_size) if full_size is not None else full_size_for(self.fft_size, sampling_rate)
        self.non_overlap = float(non_overlap)
        self.cumu_mode = cumu_mode.upper()
        if self.cumu_mode not in CUMU:
            raise KsaError("unknown cumuMode [%s]" % cumu_mode)
        self.gain = float(gain)
        self.min_amp = float(min_amp)
        self.max_frames = int(max_frames)
        self.device = int(device)
        win = window_table(window, self.fft_size) if isinstance(window, str) else np.asarray(window, dtype=np.float64)
        if len(win) != self.fft_size:
            raise KsaError("window table has %d taps, fftSize is %d" % (len(win), self.fft_size))
        self.win = win
        self.win_adj = len(win) / np.sum(win)                      # K:373
        self.starts = window_starts(self.full_size, self.fft_size, self.non_overlap)
        self.hm_width = heatmap_width(self.fft_size, int(xres))
        if self.fft_size % self.hm_width:
            raise KsaError("xRes %d does not divide fftSize %d (the reference fixes xRes up at K:937-949)" % (xres, fft_size))
        self.scan_total = int(scan_total_entries)
        self.scan_hop = 0
        self.scan_hm_width = 0
        if self.scan_total:
            hop = self.fft_size * scan_non_overlap
            if hop % 1 != 0:                                       # K:591-593
                raise KsaError("fftSize[%d] x scanRangeNonOverlap [%s] is not int" % (self.fft_size, scan_non_overlap))
            self.scan_hop = int(hop)
            self.scan_hm_width = int(scan_xres if scan_xres is not None else xres)
        self._starts32 = np.ascontiguousarray(self.starts, dtype=np.int32)
        self._win32 = np.ascontiguousarray(win, dtype=np.float32)
        cfg = Config(
            abi_version=_lib.ABI_VERSION, device=self.device, fft_size=self.fft_size, full_size=self.full_size,
            num_windows=len(self._starts32),
            window_starts=self._starts32.ctypes.data_as(C.POINTER(C.c_int32)),
            window=self._win32.ctypes.data_as(C.POINTER(C.c_float)),
            mag_scale=2.0 * self.win_adj / self.fft_size,          # K:391
            cumu_mode=CUMU[self.cumu_mode], gain=self.gain, min_amp=self.min_amp, hm_width=self.hm_width,
            max_frames=self.max_frames, u8_offset=u8_offset, u8_scale=u8_scale,
            scan_total_entries=self.scan_total, scan_hop=self.scan_hop, scan_hm_width=self.scan_hm_width)
        h = C.c_void_p()
        check(lib.ksa_create(C.byref(cfg), C.byref(h)))
        self._h = h
        if stream is not None:
            self.set_stream(stream)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib.ksa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """stream: a hipStream_t as int (torch.cuda.current_stream().cuda_stream) or None."""
        check(lib.ksa_set_stream(self._h, C.c_void_p(stream or 0)))

    def synchronize(self):
        check(lib.ksa_synchronize(self._h))

    @property
    def num_windows(self):
        return len(self.starts)

    def kernel_info(self):
        v = [C.c_int32() for _ in range(5)]
        check(lib.ksa_kernel_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("threads", "lds_bytes", "vgprs", "grid", "path"), [x.value for x in v]))

    # -- sdr_curscan drop-in (K:351-397) ---------------------------------------------------------
    def _host_iq(self, samples):
        a = np.asarray(samples)
        if a.dtype == np.uint8:
            if a.size != 2 * self.full_size:
                raise KsaError("uint8 block has %d bytes, need %d" % (a.size, 2 * self.full_size))
            return np.ascontiguousarray(a), FMT_U8
        if a.size != self.full_size:
            raise KsaError("block has %d samples, fullSize is %d" % (a.size, self.full_size))
        return np.ascontiguousarray(a, dtype=np.complex64), FMT_C64

    def curscan(self, samples):
        """One captured block -> float64[fftSize] linear magnitudes, fftshifted (what sdr_curscan returns)."""
        a, fmt = self._host_iq(samples)
        out = np.empty(self.fft_size, dtype=np.float32)
        fn = lib.ksa_curscan_u8 if fmt == FMT_U8 else lib.ksa_curscan_c64
        check(fn(self._h, _ptr(a), _ptr(out)))
        return out.astype(np.float64)

    def curscan_dev(self, iq, fmt, nframes, out, out_mode=OUT_LINEAR, frame_stride=None):
        stride = self.full_size if frame_stride is None else int(frame_stride)
        check(lib.ksa_curscan_dev(self._h, _ptr(iq), fmt, stride, int(nframes), out_mode, _ptr(out)))

    # -- zeroSpan frame loop body (K:464-484) ------------------------------------------------------
    def frame(self, samples):
        a, fmt = self._host_iq(samples)
        fn = lib.ksa_frame_u8 if fmt == FMT_U8 else lib.ksa_frame_c64
        check(fn(self._h, _ptr(a)))

    def frame_spectrum(self, mag):
        a = np.ascontiguousarray(mag, dtype=np.float32)
        if a.size != self.fft_size:
            raise KsaError("spectrum has %d bins, fftSize is %d" % (a.size, self.fft_size))
        check(lib.ksa_frame_spectrum(self._h, _ptr(a)))

    def frames_dev(self, iq, fmt, nframes, first_index=0, total_frames=None, cur_db=None, hm_rows=None,
                   commit=True, frame_stride=None):
        stride = self.full_size if frame_stride is None else int(frame_stride)
        total = int(nframes) if total_frames is None else int(total_frames)
        check(lib.ksa_frames_dev(self._h, _ptr(iq), fmt, stride, int(nframes), int(first_index), total,
                                 _ptr(cur_db), _ptr(hm_rows), 1 if commit else 0))

    def partial(self):
        """Device view float32[4, N] = {max, cur-or--inf, -min, weighted sum} of the uncommitted batch."""
        p = C.c_void_p()
        check(lib.ksa_partial_dev(self._h, C.byref(p)))
        return DevArray(p.value, (4, self.fft_size), self)

    def exchange(self):
        """Device view float32[4*N + 128*W]: the partial block followed by the waterfall ring -- what one rank
        contributes to the all-gather of a sharded run."""
        p, n = C.c_void_p(), C.c_int64()
        check(lib.ksa_exchange_dev(self._h, C.byref(p), C.byref(n)))
        return DevArray(p.value, (n.value,), self)

    def merge_gathered(self, gathered, world, frames_per_rank, hm_index0):
        """gathered: device float32[world, 4*N + 128*W] (rank order) -> merged partial + ring, committed."""
        check(lib.ksa_merge_gathered_dev(self._h, _ptr(gathered), int(world), int(frames_per_rank), int(hm_index0)))

    def commit(self, total_frames):
        check(lib.ksa_commit(self._h, int(total_frames)))

    def set_flags(self, b_max=True, b_min=True, b_avg=True):
        check(lib.ksa_set_flags(self._h, int(b_max), int(b_min), int(b_avg)))

    def set_adj(self, adj, scan=None):
        """d['Fft.Adj'] (K:400-411).  scan: False = the zeroSpan baseline (fftSize entries), True = the scan baseline
        (totalEntries entries); None picks by length and, when a one-band scan makes both lengths equal, sets both.
        adj None clears (both targets when scan is None)."""
        if adj is None:
            for target in ((0, 1) if scan is None else (int(bool(scan)),)):
                if target == 0 or self.scan_total:
                    check(lib.ksa_set_adj(self._h, target, None, 0))
            return
        a = np.ascontiguousarray(adj, dtype=np.float32)
        if scan is None:
            targets = [t for t, n in ((0, self.fft_size), (1, self.scan_total)) if n and a.size == n]
            if not targets:
                raise KsaError("adj length %d matches neither fftSize %d nor totalEntries %d" % (a.size, self.fft_size, self.scan_total))
        else:
            targets = [int(bool(scan))]
        for t in targets:
            check(lib.ksa_set_adj(self._h, t, _ptr(a), a.size))

    def reset(self):
        check(lib.ksa_reset_state(self._h))

    def state(self):
        """The plotting hand-off of the reference: d['Fft.Cur'|'Fft.Max'|'Fft.Min'|'Fft.Avg'] as
        float64[N] (K:470-476) and the waterfall buffer fed to hm.set_data (K:481)."""
        n = self.fft_size
        bufs = [np.empty(n, dtype=np.float32) for _ in range(4)]
        hm = np.empty((HM_ROWS, self.hm_width), dtype=np.float32)
        idx, seen = C.c_int32(), C.c_int64()
        check(lib.ksa_read_state(self._h, *[_ptr(b) for b in bufs], _ptr(hm), C.byref(idx), C.byref(seen)))
        out = {"Fft.Cur": bufs[0], "Fft.Max": bufs[1], "Fft.Min": bufs[2], "Fft.Avg": bufs[3]}
        out = {k: v.astype(np.float64) for k, v in out.items()}
        out.update(fftHM=hm.astype(np.float64), hm_index=idx.value, frames=seen.value)
        return out

    def state_dev(self):
        s, h = C.c_void_p(), C.c_void_p()
        check(lib.ksa_state_dev(self._h, C.byref(s), C.byref(h)))
        return DevArray(s.value, (4, self.fft_size), self), DevArray(h.value, (HM_ROWS, self.hm_width), self)

    def set_hm_index(self, i):
        check(lib.ksa_set_hm_index(self._h, int(i)))

    # -- scan (K:621-668, K:696-697) ----------------------------------------------------------------
    def scan_pass_dev(self, iq, fmt, nsteps, step_ok=None, frame_stride=None):
        stride = self.full_size if frame_stride is None else int(frame_stride)
        ok = None
        if step_ok is not None:
            ok = np.ascontiguousarray(step_ok, dtype=np.uint8)
        check(lib.ksa_scan_pass_dev(self._h, _ptr(iq), fmt, stride, int(nsteps), _ptr(ok)))

    def scan_pass(self, blocks, step_ok=None):
        """One pass from HOST memory (K:621-668 + K:696-697): blocks = [nsteps][fullSize] complex64, or
        [nsteps][2*fullSize] uint8 I,Q; staged through an engine-owned device buffer (ksa_scan_pass_c64 / _u8)."""
        a = np.asarray(blocks)
        if a.dtype == np.uint8:
            a, fn, per = np.ascontiguousarray(a), lib.ksa_scan_pass_u8, 2 * self.full_size
        else:
            a, fn, per = np.ascontiguousarray(a, dtype=np.complex64), lib.ksa_scan_pass_c64, self.full_size
        if a.ndim != 2 or a.shape[1] != per:
            raise KsaError("scan_pass wants [nsteps][%d] %s, got %s" % (per, a.dtype, a.shape))
        ok = None if step_ok is None else np.ascontiguousarray(step_ok, dtype=np.uint8)
        if ok is not None and ok.size != a.shape[0]:
            raise KsaError("step_ok has %d entries for %d steps" % (ok.size, a.shape[0]))
        check(fn(self._h, _ptr(a), int(a.shape[0]), _ptr(ok)))

    def scan_spectra_dev(self, iq, fmt, nframes, out, step_ok=None, frame_stride=None):
        """Spectrum stage of a scan alone (K:636-641): Clip2MinAmp + LogNoGain spectra of nframes blocks into the device
        buffer `out`; step_ok (host, optional): 0 -> the dummy band, written by the library on the engine's stream."""
        stride = self.full_size if frame_stride is None else int(frame_stride)
        ok = None if step_ok is None else np.ascontiguousarray(step_ok, dtype=np.uint8)
        if ok is not None and ok.size != int(nframes):
            raise KsaError("step_ok has %d entries for %d blocks" % (ok.size, nframes))
        check(lib.ksa_scan_spectra_dev(self._h, _ptr(iq), fmt, stride, int(nframes), _ptr(ok), _ptr(out)))

    def scan_stitch_dev(self, step_db, nsteps, npasses=1):
        check(lib.ksa_scan_stitch_passes_dev(self._h, _ptr(step_db), int(nsteps), int(npasses)))

    # band-sharded scan: this engine owns bands [step_lo, step_hi) and elements [elem_lo, elem_hi)
    def scan_shard(self, nsteps, rank, world):
        """(step_lo, step_hi, nhalo, elem_lo, elem_hi) of rank `rank` of `world`: contiguous balanced bands, the
        elements from its first band's start up to the next rank's (the last rank: up to totalEntries), and the
        number of bands in front of its own that still cover its elements (ceil(N/hop) - 1, at most step_lo)."""
        lo, hi = (nsteps * rank) // world, (nsteps * (rank + 1)) // world
        e_hi = self.scan_total if rank == world - 1 else min(self.scan_total, hi * self.scan_hop)
        e_lo = min(lo * self.scan_hop, e_hi)
        nhalo = min(lo, -(-self.fft_size // self.scan_hop) - 1)
        return lo, hi, nhalo, e_lo, e_hi

    def scan_stitch_range_dev(self, own_db, halo_db, nhalo, step_lo, step_hi, nsteps, npasses, elem_lo, elem_hi,
                              own_band_major=False):
        check(lib.ksa_scan_stitch_range_dev(self._h, _ptr(own_db), int(bool(own_band_major)), _ptr(halo_db), int(nhalo),
                                            int(step_lo), int(step_hi), int(nsteps), int(npasses), int(elem_lo), int(elem_hi)))

    def scan_rows(self):
        """Device view float32[rows, W] of the partial waterfall rows of the last scan_stitch_range_dev."""
        p, r = C.c_void_p(), C.c_int32()
        check(lib.ksa_scan_rows_dev(self._h, C.byref(p), C.byref(r)))
        return DevArray(p.value, (r.value, self.scan_hm_width), self)

    def scan_merge_rows(self, gathered, world, rows, npasses):
        check(lib.ksa_scan_merge_rows_dev(self._h, _ptr(gathered), int(world), int(rows), int(npasses)))

    def scan_passes_dev(self, iq, fmt, nsteps, npasses, step_ok=None, frame_stride=None):
        """A batch of captured passes ([npasses][nsteps] blocks) in one call: same state as pass-by-pass calls."""
        stride = self.full_size if frame_stride is None else int(frame_stride)
        ok = None if step_ok is None else np.ascontiguousarray(step_ok, dtype=np.uint8)
        check(lib.ksa_scan_passes_dev(self._h, _ptr(iq), fmt, stride, int(nsteps), int(npasses), _ptr(ok)))

    def scan_state(self):
        t = self.scan_total
        bufs = [np.empty(t, dtype=np.float32) for _ in range(4)]
        hm = np.empty((HM_ROWS, self.scan_hm_width), dtype=np.float32)
        idx, passes = C.c_int32(), C.c_int64()
        check(lib.ksa_scan_read_state(self._h, *[_ptr(b) for b in bufs], _ptr(hm), C.byref(idx), C.byref(passes)))
        out = {"Fft.Cur": bufs[0], "Fft.Max": bufs[1], "Fft.Min": bufs[2], "Fft.Avg": bufs[3]}
        out = {k: v.astype(np.float64) for k, v in out.items()}
        out.update(fftHM=hm.astype(np.float64), hm_index=idx.value, passes=passes.value)
        return out

    def scan_reset(self):
        check(lib.ksa_scan_reset(self._h))

    def scan_set_base_is_raw(self, on):
        check(lib.ksa_scan_set_base_is_raw(self._h, int(bool(on))))

    def levels(self, cells, mode="AVG", scan=False):
        """Cur/Max/Min/Avg decimated on the device to `cells` points (pltCompress AVG|MAX|MIN, K:205-221),
        baseline-adjusted like _adj_siglvls (K:400-411): float64[4, cells] in the order cur, max, min, avg."""
        code = {"AVG": 0, "MAX": 1, "MIN": 2}[mode.upper()]
        out = np.empty((4, int(cells)), dtype=np.float32)
        check(lib.ksa_read_levels(self._h, int(bool(scan)), code, int(cells), _ptr(out)))
        return out.astype(np.float64)

    def highs(self, cells, mode="AVG", curve="cur", min_sep=0.0, count=5, scan=False):
        """plot_highs (K:243-272) on the device: the cells of the decimated curve that the reference would mark,
        in marking order.  min_sep = delta4Marking in units of cells.  Returns (idx int32[found], levels float64[found])."""
        code = {"AVG": 0, "MAX": 1, "MIN": 2}[mode.upper()]
        cv = {"cur": 0, "max": 1, "min": 2, "avg": 3}[curve.lower()]
        idx = np.empty(int(count), dtype=np.int32)
        lvl = np.empty(int(count), dtype=np.float32)
        found = C.c_int32()
        check(lib.ksa_read_highs(self._h, int(bool(scan)), code, int(cells), cv, float(min_sep), int(count),
                                 _ptr(idx), _ptr(lvl), C.byref(found)))
        return idx[:found.value].copy(), lvl[:found.value].astype(np.float64)

    def hm_rows(self, row0, nrows, scan=False):
        """Rows [row0, row0+nrows) (mod 128) of the waterfall ring: float64[nrows, width]."""
        w = self.scan_hm_width if scan else self.hm_width
        out = np.empty((int(nrows), w), dtype=np.float32)
        check(lib.ksa_read_hm_rows(self._h, int(bool(scan)), int(row0) % HM_ROWS, int(nrows), _ptr(out)))
        return out.astype(np.float64)

    def view(self, cells, mode="AVG", curve=None, min_sep=0.0, count=0, hm_rows=1, scan=False):
        """The per-frame plot hand-off in one call (ksa_read_view): (levels float64[4, cells], marker cells, marker
        levels, newest `hm_rows` ring rows oldest first float64[hm_rows, width], hm_index).  Only cells-sized data
        crosses PCIe."""
        code = {"AVG": 0, "MAX": 1, "MIN": 2}[mode.upper()]
        cv = {"cur": 0, "max": 1, "min": 2, "avg": 3}[(curve or "cur").lower()]
        count = int(count) if curve is not None else 0
        w = self.scan_hm_width if scan else self.hm_width
        lv = np.empty((4, int(cells)), dtype=np.float32)
        idx = np.empty(max(count, 1), dtype=np.int32)
        lvl = np.empty(max(count, 1), dtype=np.float32)
        rows = np.empty((int(hm_rows), w), dtype=np.float32)
        found, hm_index = C.c_int32(0), C.c_int32()
        check(lib.ksa_read_view(self._h, int(bool(scan)), code, int(cells), _ptr(lv), cv, float(min_sep), count, _ptr(idx),
                                _ptr(lvl), C.byref(found), int(hm_rows), _ptr(rows) if hm_rows else None, C.byref(hm_index)))
        return (lv.astype(np.float64), idx[:found.value].copy(), lvl[:found.value].astype(np.float64),
                rows.astype(np.float64), hm_index.value)

    # -- measurement ------------------------------------------------------------------------------------
    def prof_enable(self, on=True):
        check(lib.ksa_prof_enable(self._h, int(on)))

    def prof_read(self):
        ms, n = C.c_double(), C.c_int64()
        check(lib.ksa_prof_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def prof_clock(self):
        """(shader clock in GHz held under the profiled spectrum launches, workgroup samples behind it); (None, 0) when the
        kernel in use carries no stamps."""
        ghz, lo, hi, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        check(lib.ksa_prof_clock(self._h, C.byref(ghz), C.byref(lo), C.byref(hi), C.byref(n)))
        self.prof_clock_range = (lo.value, hi.value) if n.value else None
        return (ghz.value if n.value else None), n.value


# ---- several engines of ONE process (one per GPU of the node, or several on one GPU): no torch, no RCCL ----------
def _handles(engines):
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    return arr, len(engines)


def allreduce_state(engines, frames_per_rank, hm_index0=0):
    """SURVEY 8(b) allreduce_state(handles[], n): merge the uncommitted time chunks of `engines` (rank order) so that
    every engine ends with the state of the whole run (ksa_allreduce_state)."""
    arr, n = _handles(engines)
    check(lib.ksa_allreduce_state(arr, n, int(frames_per_rank), int(hm_index0)))


def scan_allstitch(engines, own_db, nsteps, npasses):
    """Band-sharded scan inside one process: own_db[r] = device float32[npasses][bands of rank r][N] of engine r
    (bands split as nsteps*r/n .. nsteps*(r+1)/n): halo copies, range stitch, waterfall-row merge."""
    arr, n = _handles(engines)
    ptrs = (C.c_void_p * n)(*[(_ptr(b).value if b is not None else None) for b in own_db])
    check(lib.ksa_scan_allstitch(arr, n, ptrs, int(nsteps), int(npasses)))


def scan_gather_state(engines, nsteps):
    """The four stitched curves assembled from the engines' owned slices: dict of float64[totalEntries]."""
    arr, n = _handles(engines)
    t = engines[0].scan_total
    bufs = [np.empty(t, dtype=np.float32) for _ in range(4)]
    check(lib.ksa_scan_gather_state(arr, n, int(nsteps), *[_ptr(b) for b in bufs]))
    return {k: b.astype(np.float64) for k, b in zip(("Fft.Cur", "Fft.Max", "Fft.Min", "Fft.Avg"), bufs)}


class PinnedBuffer:
    """Page-locked host memory from ksa_host_alloc as a numpy array (capture blocks copy faster from it)."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(int(x) for x in shape), np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        check(lib.ksa_host_alloc(C.byref(self._p), nbytes))
        self.array = np.frombuffer((C.c_char * nbytes).from_address(self._p.value), dtype=self.dtype).reshape(self.shape)

    def close(self):
        if getattr(self, "_p", None) and self._p.value:
            self.array = None
            lib.ksa_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
