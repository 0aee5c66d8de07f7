"""PFB prototype-filter design (host side, float64, numpy only).

Mirrors the window expression of the reference constructor,
``/root/reference/effex/effex.py:126-127``::

    window = get_window("hamming", ntaps*nbins) * firwin(ntaps*nbins, cutoff=1/nbins, window='rectangular')

``cusignal.get_window`` / ``cusignal.firwin`` are ports of the scipy.signal functions of the
same name, so the closed forms below are scipy's: a *periodic* Hamming window (``fftbins=True``
default) times a rectangular-windowed sinc low-pass with Nyquist-normalised cutoff ``1/nbins``
scaled to unit DC gain.  No scipy is needed at run time.
"""
import numpy as np


def hamming_periodic(length):
    """scipy.signal.get_window("hamming", length) (fftbins=True)."""
    if length == 1:
        return np.ones(1)
    n = np.arange(length, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / length)


def firwin_rect_lowpass(length, cutoff):
    """scipy.signal.firwin(length, cutoff, window='rectangular'), single-band low-pass, fs=2."""
    if not 0.0 < cutoff <= 1.0:
        # scipy raises for cutoff outside (0, 1); cutoff == 1 (nbins == 1) also raises there.
        raise ValueError("cutoff must be in (0, 1]")
    alpha = 0.5 * (length - 1)
    m = np.arange(length, dtype=np.float64) - alpha
    h = cutoff * np.sinc(cutoff * m)
    # scale so the gain at DC (the centre of the first pass-band, scale frequency 0) is 1
    return h / h.sum()


def design_window(ntaps, nbins):
    """The reference's PFB window, float64, length ntaps*nbins (effex.py:126-127)."""
    length = int(ntaps) * int(nbins)
    return hamming_periodic(length) * firwin_rect_lowpass(length, 1.0 / nbins)
