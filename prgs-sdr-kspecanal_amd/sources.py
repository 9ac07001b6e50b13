"""IQ sources behind d['sdr'] -- the device seam of the reference (python/kspecanal.py:13-14, :281-308).

The reference duck-types `rtlsdr.RtlSdr`: attributes sample_rate / center_freq / gain /
valid_gains_db / bandwidth / freq_correction, methods read_samples(n) -> complex ndarray and close().
Anything with that surface can be plugged in.  Two sources ship here because the GPU box has no
dongle: a synthetic tone generator (the role of the reference's python/testfft.py, written for
numpy >= 2) and a reader for raw `rtl_sdr` uint8 captures (octave/load_rtlsdr.m:8-12,
octave/hkvc-dump_samples.sh:6).  Both can hand over uint8 I,Q pairs (`read_bytes`) so that the unpack
runs on the GPU (row A0).
"""
import numpy as np


class _SdrBase:
    valid_gains_db = [0.0, 0.9, 1.4, 2.7, 3.7, 7.7, 8.7, 12.5, 14.4, 15.7, 16.6, 19.7, 20.7, 22.9, 25.4,
                      28.0, 29.7, 32.8, 33.8, 36.4, 37.2, 38.6, 40.2, 42.1, 43.4, 43.9, 44.5, 48.0, 49.6]
    bandwidth = 0
    freq_correction = 0

    def __init__(self):
        self.sample_rate = 2.4e6
        self.center_freq = 92e6
        self.gain = 19.1

    def close(self):
        pass


class SyntheticSdr(_SdrBase):
    """Complex tones at every whole MHz that falls inside the tuned band (amplitude 0.1 .. 0.9 by MHz
    index) plus complex Gaussian noise; deterministic for a seed.  Levels follow the reference fake's
    convention of scaling by 10**(gain/10) (python/testfft.py:63) relative to its default gain."""

    def __init__(self, seed=20201226, noise=0.01, ref_gain=19.1):
        super().__init__()
        self._rng = np.random.default_rng(seed)
        self.noise = noise
        self.ref_gain = ref_gain
        self._t0 = 0

    def read_samples(self, n):
        n = int(n)
        fs, fc = float(self.sample_rate), float(self.center_freq)
        t = (self._t0 + np.arange(n, dtype=np.float64)) / fs
        self._t0 += n
        x = np.zeros(n, dtype=np.complex128)
        lo, hi = fc - fs / 2, fc + fs / 2
        for mhz in range(int(np.ceil(lo / 1e6)), int(np.floor(hi / 1e6)) + 1):
            f = mhz * 1e6
            if lo <= f < hi:
                x += (0.1 + 0.1 * (mhz % 9)) * np.exp(2j * np.pi * (f - fc) * t)
        x *= 0.25 * 10 ** ((float(self.gain) - self.ref_gain) / 10)
        x += self.noise * (self._rng.standard_normal(n) + 1j * self._rng.standard_normal(n))
        return x

    def read_bytes(self, nbytes):
        """uint8 I,Q interleaved, as the dongle delivers them (clip(round((x+1)*127.5)))."""
        x = self.read_samples(int(nbytes) // 2)
        out = np.empty(2 * len(x), dtype=np.uint8)
        out[0::2] = np.clip(np.round((x.real + 1.0) * 127.5), 0, 255)
        out[1::2] = np.clip(np.round((x.imag + 1.0) * 127.5), 0, 255)
        return out


class FileSdr(_SdrBase):
    """Replays a raw `rtl_sdr -n ... file.bin` capture: interleaved uint8 I,Q.  read_samples applies the
    documented unpack (b - 127.5)/127.5; read_bytes hands the bytes to the GPU unpack untouched.
    At end of file it raises EOFError (the reference's playback path treats any load failure as
    end of stream, python/kspecanal.py:559-563)."""

    def __init__(self, path, sample_rate=2.4e6, center_freq=92e6, loop=False):
        super().__init__()
        self.sample_rate, self.center_freq = sample_rate, center_freq
        self._raw = np.memmap(path, dtype=np.uint8, mode="r")
        self._pos = 0
        self.loop = loop

    def read_bytes(self, nbytes):
        nbytes = int(nbytes)
        if self._pos + nbytes > len(self._raw):
            if not self.loop or nbytes > len(self._raw):
                raise EOFError("capture exhausted")
            self._pos = 0
        out = np.array(self._raw[self._pos:self._pos + nbytes])
        self._pos += nbytes
        return out

    def read_samples(self, n):
        b = self.read_bytes(2 * int(n)).astype(np.float64)
        return (b[0::2] - 127.5) / 127.5 + 1j * ((b[1::2] - 127.5) / 127.5)
