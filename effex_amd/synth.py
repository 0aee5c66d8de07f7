"""Deterministic synthetic IQ source (host side).

The reference gets its samples from two RTL-SDR dongles (``/root/reference/effex/effex.py:81-82``,
``:652``); there is no hardware here, so benchmarks and parity tests use this counter-based
generator instead.  The same arithmetic is implemented on the device by ``fxc_synth_fill``
(``effex_amd/csrc/k_synth.h``) and the two are bit-identical (tests/test_gpu_parity.py), so a
multi-GiB pool can be produced in HBM without a host copy.

Model (all float32, RTL-SDR-like 8-bit quantised IQ, ``(byte - 127.5) / 127.5``):

    g        = G0 + chunk*num_samp + n                    global sample index of the stream
    x_a[g]   = sky[g - d_a] + 0.5 * rx_a[g] + tone[(g - d_a) mod TONE_PERIOD]

``sky`` is a noise stream common to all antennas (so the cross-spectrum is non-trivial), ``rx_a``
is receiver noise private to antenna ``a``, ``d_a`` an integer geometric delay and ``tone`` a
0.1-amplitude complex exponential with an integer period of 24 samples (+100 kHz at 2.4 Msps).
"""
import numpy as np

MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)
G0 = 1 << 20              # index offset so g - d_a never goes negative
TONE_PERIOD = 24          # samples; 2.4e6 / 24 = +100 kHz
TONE_AMP = 0.1
RX_SCALE = 0.5            # exact power of two: fma and mul+add round identically
DEFAULT_DELAYS = (0, 3, 7, 12, 18, 25, 33, 42)

_C_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C_M1 = np.uint64(0xBF58476D1CE4E5B9)
_C_M2 = np.uint64(0x94D049BB133111EB)
_C_STREAM = np.uint64(0xD1B54A32D192ED03)
_C_SEED = np.uint64(0x8CB92BA72F3D8DD7)


def mix64(z):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + _C_GOLD)
        z = (z ^ (z >> np.uint64(30))) * _C_M1
        z = (z ^ (z >> np.uint64(27))) * _C_M2
        return z ^ (z >> np.uint64(31))


def _stream_key(seed, stream):
    with np.errstate(over="ignore"):
        return np.uint64(seed) * _C_SEED + np.uint64(stream) * _C_STREAM


def _iq_from_hash(h):
    """Two uniform bytes of the hash -> (re, im) float32 in [-1, 1]."""
    b_re = (h & np.uint64(0xFF)).astype(np.float32)
    b_im = ((h >> np.uint64(8)) & np.uint64(0xFF)).astype(np.float32)
    scale = np.float32(127.5)
    return (b_re - scale) / scale, (b_im - scale) / scale


def tone_table():
    """complex64[TONE_PERIOD]; formed in float64 on the host and handed to the device as data."""
    k = np.arange(TONE_PERIOD, dtype=np.float64)
    return (TONE_AMP * np.exp(2j * np.pi * k / TONE_PERIOD)).astype(np.complex64)


def synth_iq(seed, n_chunks, n_ant, num_samp, first_chunk=0, delays=DEFAULT_DELAYS):
    """Return complex64 array [n_chunks, n_ant, num_samp] of the synthetic stream.

    ``first_chunk`` offsets the global chunk index so a rank can generate only its own shard.
    """
    if n_ant > len(delays):
        raise ValueError("not enough delays for n_ant")
    tone = tone_table()
    out = np.empty((n_chunks, n_ant, num_samp), dtype=np.complex64)
    n = np.arange(num_samp, dtype=np.uint64)
    key_sky = _stream_key(seed, 0)
    with np.errstate(over="ignore"):
        for c in range(n_chunks):
            g = np.uint64(G0) + np.uint64((first_chunk + c) * num_samp) + n
            for a in range(n_ant):
                gd = g - np.uint64(delays[a])
                s_re, s_im = _iq_from_hash(mix64(key_sky + gd))
                r_re, r_im = _iq_from_hash(mix64(_stream_key(seed, a + 1) + g))
                t = tone[(gd % np.uint64(TONE_PERIOD)).astype(np.int64)]
                half = np.float32(RX_SCALE)
                re = (s_re + half * r_re) + t.real
                im = (s_im + half * r_im) + t.imag
                out[c, a].real = re
                out[c, a].imag = im
    return out
