"""ORACLE support — the seeded inputs behind tests/golden/ (shared by make_golden.py and tests/).

TEST INFRASTRUCTURE ONLY.  Inputs are regenerated from these definitions instead of being
stored; ``tests/golden/reference_outputs.*`` hold the reference's outputs for them.
"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effex_amd import synth  # noqa: E402

WINDOW_CASES = ((4096, 4), (2048, 4), (4096, 32), (2048, 32))

SEED_PARITY = 1234
SEED_TESTS = 77777          # the reference's own test seed, tests/test_effex.py:10

BANDWIDTH = 2.4e6
FREQUENCY = 1.4204e9
XCORR_CASES = (("SPECTRUM", 0.0), ("SPECTRUM", 1e-6), ("CONTINUUM", 0.0), ("CONTINUUM", 1e-6),
               ("TEST", 1e-6))

SMALL_S = 8192
SMALL_N = 512
SMALL_CHUNKS = 5

# the other --nfft values (effex.py:778) at the reference's fixed ntaps = 4: (nbins, num_samp, chunk pairs, delay)
NFFT_CASES = ((1024, 1024 * 12, 2, 1e-6), (2048, 2048 * 9 + 5, 2, -3e-7), (8192, 8192 * 5, 1, 1e-6),
              # --resolution is a free integer (effex.py:733-739): counts that are not a power of two, one of them prime
              (1000, 1000 * 12 + 3, 2, 1e-6), (96, 96 * 300, 1, -3e-7), (997, 997 * 9, 1, 0.0), (1536, 1536 * 7, 1, 2e-7))
SEED_NFFT = 4321

# nbins changed AFTER construction (tests/test_effex.py:142-144): the window built once by the constructor for 4096 bins
# (effex.py:126-127) stays, and channelize_poly derives its tap count from it: (new nbins, num_samp, chunk pairs, delay)
STALE_NBINS_CASES = ((2048, 2048 * 12 + 3, 2, 1e-6), (8192, 8192 * 4, 1, 0.0), (1000, 1000 * 6 + 7, 1, -2e-7))
SEED_STALE = 2468

CSV_NBINS = 256
CSV_S = 4096

DELAY_RATE = 2.4e6


def window_sample_indices(length):
    return np.array([0, 1, 2, length // 4, length // 2 - 1, length // 2, length - 2, length - 1])


def kat_input():
    """N=4, T=2 known-answer input of SURVEY.md §2.3: 13 samples (the last one is dropped)."""
    n = np.arange(13)
    x = (n + 1) - 1j * n
    h = np.arange(1, 9) / 10.0
    return x.astype(np.complex128), h


def tone_cases():
    """The 32 parametrisations of tests/test_effex.py:62-66, in pytest's nesting order."""
    return [(num_samp, rate, freq, taps, branches)
            for branches, taps, freq, rate, num_samp in itertools.product(
                [2048, 4096], [4, 32], [2e4, 1e5], [1e6, 2.4e6], [3 + 2 ** 12, 2 ** 18])]


def tone_iq(num_samp, rate, freq):
    """Noise-free complex tone of tests/test_effex.py:31-41 (cos + i sin on a linspace time axis)."""
    t = np.linspace(0, num_samp / rate, num=num_samp)
    omega = 2. * np.pi * freq
    return np.cos(omega * t) + 1j * np.sin(omega * t)


def spec_sample_indices(shape):
    """16 (row, col) sample positions of a (P, N) spectrum array."""
    p, n = shape
    k = np.arange(16)
    rows = (k * 5) % p
    cols = (k * 977 + 13) % n
    return rows, cols


def xcorr_input():
    """[2, 262144] complex64 — the §8d synthetic chunk pair, seed 1234."""
    return synth.synth_iq(SEED_PARITY, 1, 2, 2 ** 18)[0]


def small_input():
    return synth.synth_iq(SEED_PARITY, SMALL_CHUNKS, 2, SMALL_S)


def nfft_input(nbins, num_samp, chunks):
    return synth.synth_iq(SEED_NFFT + nbins, chunks, 2, num_samp)


def stale_input(nbins, num_samp, chunks):
    return synth.synth_iq(SEED_STALE + nbins, chunks, 2, num_samp)


def csv_row(mode):
    iq = synth.synth_iq(SEED_TESTS, 1, 1, CSV_NBINS)[0, 0].astype(np.complex128)
    if mode == "SPECTRUM":
        return iq * 1e-5
    return iq[:1] * 1e-12


def delay_cases():
    return [(num_samp, off) for num_samp in (3 + 2 ** 12, 2 ** 18)
            for off in (-2000, -1001, -1, 0, 1, 999, 2000)]


def noise_iq(num_samp):
    """Stand-in for the reference's Gaussian noise (tests/test_effex.py:44-49): uniform IQ noise
    from the counter-based generator, scaled to ~0.1 rms."""
    return (synth.synth_iq(SEED_TESTS, 1, 1, num_samp, delays=(0,))[0, 0] * np.float32(0.15)).astype(np.complex128)
