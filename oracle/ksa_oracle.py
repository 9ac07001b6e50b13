"""CPU oracle for the kspecanal hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A float64 numpy restatement of the overlapped windowed-FFT -> magnitude ->
cumulate -> dB -> Max/Min/Avg/Cur + waterfall arithmetic of the reference
(`/root/reference/python/kspecanal.py`, cited below as K:<line>).

Who may use this file: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- as the checker / the timed CPU baseline.
The shipped package (`prgs-sdr-kspecanal_amd/`) never imports it and has no CPU
fallback: without the HIP library it raises.

Parity status: PINNED for rows A2-A13 of SURVEY.md section 8 by
`tests/golden/*.npz`, which were produced by executing the reference itself in
the build container (`tests/golden/make_golden.py`) and are compared against
this file bit-for-bit (float64) in `tests/test_oracle_golden.py`.
Row A0 (uint8 IQ unpack) is "parity unpinned": the arithmetic lives in the
third-party `pyrtlsdr` package (version not pinned by the reference,
README.rst:62; not installed here, no source in the tree).  The convention
`(b - 127.5) / 127.5` used below is pyrtlsdr's published `packed_bytes_to_iq`
numpy path restated from memory; the in-tree precedents differ
(kspecanal.old.py:126-135 uses (b-127)/128, octave/load_rtlsdr.m:9-12 uses
b-127), so offset and scale are parameters.
"""

import numpy as np

CUMU_RAW, CUMU_AVG, CUMU_MAX, CUMU_MIN = "RAW", "AVG", "MAX", "MIN"
HM_ROWS = 128  # K:448, K:611


# --------------------------------------------------------------------------- A0
def unpack_u8(raw, offset=127.5, scale=127.5):
    """Interleaved I,Q uint8 -> complex128.  pyrtlsdr packed_bytes_to_iq (call
    sites K:301, K:339, K:346): iq = bytes.view(pairs); iq/127.5 - (1+1j)."""
    b = np.asarray(raw, dtype=np.uint8).astype(np.float64)
    return (b[0::2] - offset) / scale + 1j * ((b[1::2] - offset) / scale)


def quantize_u8(x):
    """Inverse used by the synthetic source: clip(round((x+1)*127.5))."""
    x = np.asarray(x)
    out = np.empty(2 * x.size, dtype=np.uint8)
    out[0::2] = np.clip(np.round((x.real + 1.0) * 127.5), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.round((x.imag + 1.0) * 127.5), 0, 255).astype(np.uint8)
    return out


# --------------------------------------------------------------------------- A2
def full_size(fft_size, sampling_rate):
    """Samples captured per curscan (K:926-929, multipliers K:49-50)."""
    return fft_size * 8 if fft_size < (sampling_rate // 8) else fft_size * 2


def window_starts(full, fft_size, non_overlap):
    """Start index of every window the reference actually transforms.

    K:368 numLoops = int(fullSize/(fftSize*nonOverlap)); K:386 iStart =
    int(i*fftSize*nonOverlap) evaluated in float64 (left-to-right product, then
    truncation); K:389-390 stops at the first window that runs past the block."""
    num_loops = int(full / (fft_size * non_overlap))
    starts = []
    for i in range(num_loops):
        s = int(i * fft_size * non_overlap)
        if s + fft_size > full:
            break
        starts.append(s)
    return np.asarray(starts, dtype=np.int64)


# --------------------------------------------------------------------------- A3
def window_table(name, n, kaiser_beta=64):
    """K:932-935.  `name` is the CLI spelling (ones|hanning|hamming|kaiser)."""
    name = name.lower().replace("win.", "")
    if name == "ones":
        return np.ones(n)
    if name == "hanning":
        return np.hanning(n)
    if name == "hamming":
        return np.hamming(n)
    if name == "kaiser":
        return np.kaiser(n, kaiser_beta)
    raise ValueError("unknown window [%s]" % name)


def win_adj(win):
    """K:373: amplitude compensation N / sum(w)."""
    return len(win) / np.sum(win)


# --------------------------------------------------------------------- A7 / A10
def data_cumu(mode, cur, c0, c1, new, n0, n1):
    """K:124-147.  None seeds by copy; AVG is (cur+new)/2 (an alpha=1/2 EMA)."""
    if cur is None:
        return np.copy(new)
    if mode == CUMU_RAW:
        cur[c0:c1] = new[n0:n1]
    elif mode == CUMU_AVG:
        cur[c0:c1] += new[n0:n1]
        cur[c0:c1] /= 2
    elif mode == CUMU_MAX:
        cur[c0:c1] = np.max([new[n0:n1], cur[c0:c1]], axis=0)
    elif mode == CUMU_MIN:
        cur[c0:c1] = np.min([new[n0:n1], cur[c0:c1]], axis=0)
    else:
        raise ValueError("unknown cumuMode [%s]" % mode)
    return cur


# ---------------------------------------------------------------------- A4 - A8
def window_spectrum(seg, win):
    """One window: K:391  winAdj*2*abs(fft(seg*win))/len(seg)."""
    return win_adj(win) * 2 * np.abs(np.fft.fft(seg * win)) / len(seg)


def curscan(samples, fft_size, non_overlap, win, cumu_mode=CUMU_AVG):
    """sdr_curscan minus the device read (K:368-397): slide, window, FFT,
    magnitude, fold with data_cumu, fftshift.  Returns float64[fft_size]."""
    samples = np.asarray(samples, dtype=np.complex128)  # K:335 allocates complex128
    acc = None
    for s in window_starts(len(samples), fft_size, non_overlap):
        cur = window_spectrum(samples[s:s + fft_size], win)
        if acc is None:
            acc = cur
        else:
            acc = data_cumu(cumu_mode, acc, 0, len(acc), cur, 0, len(cur))
    return np.fft.fftshift(acc)


# --------------------------------------------------------------------------- A9
def clip2minamp(vals, min_amp):
    """K:100-101."""
    return np.clip(vals, min_amp, None)


def log_no_gain(vals, gain, inf_to=None):
    """K:106-112: 10*log10(v) - gain (10, not 20), optional +-inf replacement."""
    with np.errstate(divide="ignore", invalid="ignore"):
        out = 10 * np.log10(vals) - gain
    if inf_to is not None:
        out[np.isinf(out)] = inf_to
    return out


# -------------------------------------------------------------------------- A12
def plotcompress(data, xres, mode="MAX"):
    """_data_plotcompress K:168-202 for MAX / AVG (MIN is unreachable there)."""
    rows = xres
    cols = len(data) // rows
    if cols == 0:
        return data
    t = data.reshape(rows, cols)
    if mode == "MAX":
        return np.max(t, axis=1)
    if mode == "AVG":
        return np.average(t, axis=1)
    raise ValueError("plotcompress mode [%s]" % mode)


def heatmap_width(fft_size, xres):
    """K:449-455 with pltCompressHM fixed to MAX (K:67)."""
    return xres if fft_size > xres else fft_size


# ----------------------------------------------------- (f2) plot-side reductions
def data_plotcompress(x, y, xres, mode):
    """K:205-221: the x axis is group-averaged, y reduced with `mode`; RAW passes both through."""
    if mode == "RAW":
        return x, y
    return plotcompress(x, xres, "AVG"), plotcompress(y, xres, mode)


def plot_highs(freqs, levels, delta_frac, num_markers):
    """Peak markers of plot_highs (K:243-272) as a list of (freq, level): walk the points from the highest
    level down, mark one unless an already marked frequency lies closer than delta_frac * (freqs[-1]-freqs[0])
    (K:249-250, K:261-262), stop after num_markers (K:268-269).  The walk covers i = -1 .. -(len-1) of the
    ascending argsort (K:258), i.e. the lowest point is never visited."""
    delta = delta_frac * (freqs[-1] - freqs[0])
    order = np.argsort(levels)
    marked = []
    for j in range(1, len(freqs)):
        i = order[-j]
        if all(abs(f - freqs[i]) >= delta for f, _ in marked):
            marked.append((float(freqs[i]), float(levels[i])))
            if len(marked) >= num_markers:
                break
    return marked


def adj_siglvls(state, adj):
    """_adj_siglvls K:400-411 on a ZeroSpanState / ScanState: (max, min, avg, cur) minus the saved baseline."""
    if adj is None:
        return state.max, state.min, state.avg, state.cur
    return state.max - adj, state.min - adj, state.avg - adj, state.cur - adj


# ---------------------------------------------------------------- A10 - A12 loop
class ZeroSpanState:
    """The per-frame body of zero_span (K:464-484) without SDR / plotting."""

    def __init__(self, fft_size, xres, gain, adj=None,
                 b_max=True, b_min=True, b_avg=True):
        self.gain = gain
        self.xres = xres
        self.adj = adj
        self.b_max, self.b_min, self.b_avg = b_max, b_min, b_avg
        self.cur = self.max = self.min = self.avg = None
        self.hm = np.zeros((HM_ROWS, heatmap_width(fft_size, xres)))  # K:456
        self.hm_index = 0

    def push(self, fft_cur):
        """fft_cur = curscan output (linear magnitude, fftshifted)."""
        pr = log_no_gain(fft_cur, self.gain)                      # K:469
        self.cur = pr                                             # K:470
        n = len(pr)
        if self.b_max:
            self.max = data_cumu(CUMU_MAX, self.max, 0, n, pr, 0, n)   # K:472
        if self.b_min:
            self.min = data_cumu(CUMU_MIN, self.min, 0, n, pr, 0, n)   # K:474
        if self.b_avg:
            self.avg = data_cumu(CUMU_AVG, self.avg, 0, n, pr, 0, n)   # K:476
        row_src = pr - self.adj if self.adj is not None else pr   # K:405 / K:410
        self.hm[self.hm_index, :] = plotcompress(row_src, self.xres, "MAX")  # K:480
        self.hm_index = (self.hm_index + 1) % HM_ROWS             # K:484
        return pr


# -------------------------------------------------------------------------- A13
def fixup_scan_range(start_freq, end_freq, sampling_rate):
    """_fixupfreqs_scanrange K:701-709 -> (end_freq, center_freq)."""
    bands = (end_freq - start_freq) / sampling_rate
    if (bands % 1) != 0:
        end_freq = start_freq + np.ceil(bands) * sampling_rate
    return end_freq, start_freq + (end_freq - start_freq) / 2


def scan_steps(start_freq, end_freq, sampling_rate, scan_non_overlap):
    """Tuned centre frequency of every step of one pass (K:594, K:621, K:689-690)."""
    span = sampling_rate
    cur = start_freq + span / 2
    lo = cur - span / 2
    out = []
    while lo < end_freq:
        out.append(cur)
        cur += span * scan_non_overlap
        lo = cur - span / 2
    return out


class ScanState:
    """_scan_range (K:568-698) without SDR / plotting: stitch + accumulate."""

    def __init__(self, fft_size, start_freq, end_freq, sampling_rate, gain,
                 min_amp, xres, scan_non_overlap=0.5, adj=None,
                 b_max=True, b_min=True, base_is_raw=False):
        self.n = fft_size
        self.q = scan_non_overlap
        self.gain, self.min_amp, self.xres, self.adj = gain, min_amp, xres, adj
        self.b_max, self.b_min, self.base_is_raw = b_max, b_min, base_is_raw
        total_freqs = end_freq - start_freq
        self.num_groups = int(total_freqs / sampling_rate)         # K:599
        self.total = self.num_groups * fft_size                     # K:600
        self.centers = scan_steps(start_freq, end_freq, sampling_rate, scan_non_overlap)
        floor = log_no_gain(np.ones(self.total) * min_amp, gain, inf_to=0)  # K:603-604
        self.cur = floor
        self.max = np.copy(floor)                                   # K:605
        self.avg = np.copy(floor)                                   # K:606
        self.min = log_no_gain(np.ones(self.total), gain, inf_to=0)  # K:607-608
        hm = np.ones([HM_ROWS, self.total]) * min_amp               # K:613
        self.hm = np.array([plotcompress(hm[r, :], xres, "MAX") for r in range(HM_ROWS)])  # K:614
        self.hm_index = 0
        self.passes = 0

    def run_pass(self, step_spectra):
        """step_spectra[i] = curscan output for step i (or None -> dummy ones, K:637-639)."""
        n, q, total = self.n, self.q, self.total
        mode_avg = CUMU_RAW if self.passes == 0 else CUMU_AVG       # K:615-618
        i_old_end = 0
        for i, fft_cur in enumerate(step_spectra):
            i_start = int(i * n * q)                                # K:622
            i_end = i_start + n
            i_done = int((i + 1) * n * q)                           # K:624
            s_start = 0
            s_end = (i_end - i_start - (i_end - total)) if i_end > total else (i_end - i_start)
            if fft_cur is None:
                fft_cur = np.ones(n)
            fft_cur = clip2minamp(fft_cur, self.min_amp)            # K:640
            pr = log_no_gain(np.copy(fft_cur), self.gain, inf_to=0)  # K:641
            s_raw_start = s_start + (n - (i_end - i_old_end))       # K:643
            self.cur = data_cumu(CUMU_RAW, self.cur, i_old_end, i_end, pr, s_raw_start, s_end)
            if i_old_end != 0:
                if i_old_end > total:
                    i_old_end = total
                s_avg_end = s_start + (i_old_end - i_start)
                self.cur = data_cumu(CUMU_AVG, self.cur, i_start, i_old_end, pr, s_start, s_avg_end)
            i_old_end = i_end
            if self.base_is_raw:                                    # K:651-656
                d0, d1, src, s0, s1 = i_start, i_end, pr, s_start, s_end
            else:                                                   # K:657-662
                d0, d1, src, s0, s1 = i_start, i_done, self.cur, i_start, i_done
            if self.b_max:
                self.max = data_cumu(CUMU_MAX, self.max, d0, d1, src, s0, s1)
            if self.b_min:
                self.min = data_cumu(CUMU_MIN, self.min, d0, d1, src, s0, s1)
            self.avg = data_cumu(mode_avg, self.avg, d0, d1, src, s0, s1)   # K:667-668
        avg_src = self.avg - self.adj if self.adj is not None else self.avg  # K:669
        self.hm[self.hm_index, :] = plotcompress(avg_src, self.xres, "MAX")  # K:697
        self.hm_index = (self.hm_index + 1) % HM_ROWS                        # K:732
        self.passes += 1


# ------------------------------------------------------------ synthetic IQ source
def synth_iq(n, seed, tones=((0.125, 0.5), (-0.25, 0.25), (0.3173, 0.05)), sigma=0.05):
    """SURVEY 8(d) synthetic input: three complex tones + complex Gaussian noise,
    generated in float64.  Returns complex128[n]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    x = np.zeros(n, dtype=np.complex128)
    for f, a in tones:
        x += a * np.exp(2j * np.pi * f * t)
    x += sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x


# ---------------------------------------------------------- batch-level helpers
def zerospan_batch(frames_c, fft_size, non_overlap, win, cumu_mode, gain, xres,
                   state=None, adj=None):
    """Run ZeroSpanState over frames_c[F, fullSize]; returns (state, cur_db[F,N], lin[F,N])."""
    if state is None:
        state = ZeroSpanState(fft_size, xres, gain, adj=adj)
    lin = np.empty((len(frames_c), fft_size))
    db = np.empty((len(frames_c), fft_size))
    for f, fr in enumerate(frames_c):
        lin[f] = curscan(fr, fft_size, non_overlap, win, cumu_mode)
        db[f] = state.push(np.copy(lin[f]))
    return state, db, lin
