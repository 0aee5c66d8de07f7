"""ORACLE — CPU restatement of effex's F- and X-stage.  TEST INFRASTRUCTURE ONLY.

This module is the *checker* for the HIP path.  Nothing under ``effex_amd/`` imports it; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.  It is never
the thing measured as the product and never a fallback.

Parity status: **pinned against the reference executed in the build container** — the reference's
own ``effex.py`` is imported through stand-in modules (``oracle/ref_standins.py``) and the outputs of
its unmodified ``_spectrometer_poly`` / ``_pfb_xcorr`` / ``_write_metadata`` /
``_estimate_delay_gaussian`` are committed as fixtures under ``tests/golden/`` by
``oracle/make_golden.py``; ``tests/test_oracle.py`` checks this restatement against them.  The one
piece of arithmetic that is *not* in the reference tree is ``cusignal.filtering.channelize_poly``
(cusignal is an un-vendored, un-pinned dependency: ``requirements.txt`` is empty, README lists the
name only).  Its published definition (the Python reference loop in cusignal's own documentation /
tests: a per-branch shift register fed with the conjugated, branch-reversed input frame, dotted
with the conjugated polyphase taps, followed by ``conj(fft(.))``) is restated here twice — literal
loop and vectorised closed form — and the two are checked against each other.  What pins which fact of that
third-party function (``tests/test_oracle.py``):

* tap order (``h[t N + m]`` meets ``x[(i - t) N + N - 1 - m]``), branch reversal, zero history before sample 0,
  dropped tail, unit scale: the **filter-bank identity** — channel k equals the input filtered with the prototype
  modulated to that channel and decimated by N, computed with ``scipy.signal.lfilter`` from first principles
  (``test_channelizer_is_the_textbook_analysis_filter_bank``, 1e-11), which shares no code or memory with the
  restatement;
* sign of the frequency axis: the same identity, and independently the **reference's own test** criterion
  (``tests/test_effex.py:62-89``, tone arg-max within 1 %, all 32 cases);
* that cusignal's kernel IS this filter bank (in particular its output conjugation ``conj(fft(.))`` over conjugated
  inputs, which equals the +i kernel only for the real prototype filters the reference designs): **cusignal's
  published definition only** — cusignal is not in this image and cannot be executed; a release whose CUDA kernel
  deviated from its own documented loop could not be detected offline.  Recorded in DESIGN.md.

Every function cites the reference lines it follows (paths are relative to /root/reference).
"""
import numpy as np


# --------------------------------------------------------------------------------------------
# cusignal.filtering.channelize_poly — call site effex/effex.py:553 (third-party definition)
# --------------------------------------------------------------------------------------------
def channelize_poly_loop(x, h, n_chans):
    """Literal transcription of cusignal's documented CPU definition (slow; small inputs only).

    Returns (n_chans, n_pts) like cusignal.  Follows the call at effex/effex.py:553.
    """
    x = np.asarray(x)
    h = np.asarray(h)
    n_taps = int(len(h) / n_chans)
    if n_taps > 32:
        # cusignal only ships 8x8 / 16x16 / 32x32 kernels
        raise NotImplementedError("Number of taps ({}) must be less than (32).".format(n_taps))
    n_pts = int(len(x) / n_chans)
    dtype = np.promote_types(x.dtype, h.dtype)
    dtype = np.promote_types(dtype, np.complex64)
    hh = np.conj(np.reshape(h[: n_taps * n_chans].astype(dtype), (n_taps, n_chans)).T)
    reg = np.zeros((n_chans, n_taps), dtype=dtype)
    vv = np.empty(n_chans, dtype=dtype)
    yy = np.empty((n_chans, n_pts), dtype=dtype)
    for i in range(n_pts):
        nn = i * n_chans
        # shift register: newest frame in column 0, branch order reversed, conjugated
        reg[:, 1:n_taps] = reg[:, 0:(n_taps - 1)].copy()
        reg[:, 0] = np.conj(x[nn:nn + n_chans][::-1])
        for mm in range(n_chans):
            vv[mm] = np.dot(reg[mm, :], hh[mm, :])
        yy[:, i] = np.conj(np.fft.fft(vv))
    return yy


def pfb_fir(x, h, n_chans):
    """FIR half of channelize_poly in the un-conjugated closed form of SURVEY.md §2.3:

        v[i, m] = sum_{t < T, i-t >= 0} x[(i-t)*N + (N-1-m)] * h[t*N + m]        (h real)

    Zero history before sample 0; trailing ``len(x) mod N`` samples are ignored.
    """
    x = np.asarray(x)
    h = np.asarray(h)
    n_taps = int(len(h) / n_chans)
    n_pts = int(len(x) / n_chans)
    frames = x[: n_pts * n_chans].reshape(n_pts, n_chans)[:, ::-1]
    hh = h[: n_taps * n_chans].reshape(n_taps, n_chans)
    v = np.zeros((n_pts, n_chans), dtype=np.promote_types(x.dtype, np.result_type(h.dtype, np.complex64)))
    for t in range(min(n_taps, n_pts)):
        v[t:] += frames[: n_pts - t] * hh[t]
    return v


def channelize_poly(x, h, n_chans):
    """Vectorised channelize_poly: (n_chans, n_pts), natural (un-shifted) bin order.

    spec[i, k] = sum_m v[i, m] * exp(+2*pi*i*k*m/N)   (= conj(fft(conj(v)))) for real h.
    """
    n_taps = int(len(h) / n_chans)
    if n_taps > 32:
        raise NotImplementedError("Number of taps ({}) must be less than (32).".format(n_taps))
    v = pfb_fir(x, h, n_chans)
    spec = np.fft.ifft(v, axis=1) * n_chans
    return spec.T


# --------------------------------------------------------------------------------------------
# Correlator._spectrometer_poly — effex/effex.py:530-555
# --------------------------------------------------------------------------------------------
def spectrometer_poly(x, ntaps, n_branches, window, dtype=np.complex128):
    """(len(x)//n_branches, n_branches) complex.  ``ntaps`` is unused by the reference too.

    effex.py:551 is a complex128 *copy*, not a pad (``zeros(len+r)[:len] + x``); effex.py:553 is
    ``channelize_poly(x, window, n_branches).T``.
    """
    x = np.asarray(x)
    x = np.zeros(len(x) + len(x) % n_branches, dtype=dtype)[: len(x)] + x
    return channelize_poly(x, np.asarray(window), n_branches).T


# --------------------------------------------------------------------------------------------
# Correlator._pfb_xcorr — effex/effex.py:497-527
# --------------------------------------------------------------------------------------------
def rot_table(nbins, bandwidth, frequency, calibrated_delay):
    """effex.py:516,519 — rot[k] = exp(-2j*pi*freqs*(-delay)), natural bin order, complex128."""
    freqs = np.fft.fftfreq(nbins, d=1 / bandwidth) + frequency
    return np.exp(-2j * np.pi * freqs * (-calibrated_delay))


def xpower(f0, f1, rot):
    """effex.py:520-521 — fftshift(mean_i(f0 * conj(f1 * rot)))."""
    xpower_spec = f0 * np.conj(f1 * rot)
    return np.fft.fftshift(xpower_spec.mean(axis=0))


def pfb_xcorr(iq_0, iq_1, ntaps, nbins, window, bandwidth, frequency, calibrated_delay, mode,
              dtype=np.complex128):
    """One reference ``_run_task()``: (nbins,) complex (SPECTRUM) or complex scalar."""
    f0 = spectrometer_poly(iq_0, ntaps, nbins, window, dtype)
    f1 = spectrometer_poly(iq_1, ntaps, nbins, window, dtype)
    rot = rot_table(f0.shape[-1], bandwidth, frequency, calibrated_delay)
    xpower_spec = xpower(f0, f1, rot)
    if mode in ("CONTINUUM", "TEST"):
        return xpower_spec.mean(axis=0) / bandwidth      # effex.py:523-524
    return xpower_spec


def fx_integrate(x, nbins, window, rot=None, dtype=np.complex128):
    """Build extension: multi-antenna, multi-chunk integration.

    x: [n_chunks, n_ant, num_samp].  Returns [n_baselines, nbins] complex =
    fftshift( mean over all chunks and spectra of spec_a * conj(spec_b * rot) ), baselines ordered
    (0,1),(0,2)...(A-2,A-1).  With n_chunks == 1 and n_ant == 2 this is exactly ``pfb_xcorr``.
    """
    x = np.asarray(x)
    n_chunks, n_ant, _ = x.shape
    pairs = [(a, b) for a in range(n_ant) for b in range(a + 1, n_ant)]
    acc = np.zeros((len(pairs), nbins), dtype=np.complex128)
    n_spec = 0
    ntaps = len(window) // nbins
    for c in range(n_chunks):
        specs = [spectrometer_poly(x[c, a], ntaps, nbins, window, dtype) for a in range(n_ant)]
        n_spec += specs[0].shape[0]
        for p, (a, b) in enumerate(pairs):
            acc[p] += (specs[a] * np.conj(specs[b])).sum(axis=0)
    acc /= max(n_spec, 1)
    if rot is not None:
        acc = acc * np.conj(rot)
    return np.fft.fftshift(acc, axes=-1)


# --------------------------------------------------------------------------------------------
# Input conditioning — effex/effex.py:394-395 and pyrtlsdr's byte -> sample conversion (effex.py:652)
# --------------------------------------------------------------------------------------------
def remove_dc(x):
    """effex.py:394-395 — subtract the mean of the real and of the imaginary part (per chunk)."""
    x = np.asarray(x)
    return (x.real - x.real.mean()) + 1j * (x.imag - x.imag.mean())


def u8_to_complex(iq_u8):
    """pyrtlsdr ``packed_bytes_to_iq`` [third party, recollection]: interleaved uint8 I,Q -> complex128."""
    b = np.asarray(iq_u8, dtype=np.float64)
    return (b[..., 0] - 127.5) / 127.5 + 1j * (b[..., 1] - 127.5) / 127.5


# --------------------------------------------------------------------------------------------
# Output format — effex/effex.py:667-696 (consumers: effex.py:798, post_process.py:201-219)
# --------------------------------------------------------------------------------------------
def metadata_header(run_time, bandwidth, frequency, num_samp, nbins, gain, mode):
    """effex.py:672-678."""
    fields = (('run_time', run_time), ('bandwidth', bandwidth), ('frequency', frequency),
              ('num_samp', num_samp), ('resolution', nbins), ('gain', gain), ('mode', mode))
    return ','.join('{}:{}'.format(k, v) for k, v in fields) + '\n'


def write_metadata(fh, run_time, bandwidth, frequency, num_samp, nbins, gain, mode):
    """effex.py:667-684."""
    fh.write(metadata_header(run_time, bandwidth, frequency, num_samp, nbins, gain, mode))
    if 'SPECTRUM' == mode:
        freqs = np.fft.fftshift(np.fft.fftfreq(nbins, d=1 / bandwidth)) + frequency
        np.savetxt(fh, [freqs], delimiter=',')
    else:
        np.savetxt(fh, [])


def write_row(fh, vis):
    """effex.py:693 — one csv line per chunk-pair."""
    np.savetxt(fh, [np.asarray(vis)], delimiter=',')


# --------------------------------------------------------------------------------------------
# Delay calibration — effex/effex.py:583-627 (SURVEY.md §8f "next" #2)
# --------------------------------------------------------------------------------------------
def estimate_delay_gaussian(iq_0, iq_1, rate):
    """effex.py:598-627."""
    assert len(iq_0) == len(iq_1)
    n = len(iq_0)
    a = np.zeros(2 * n, dtype=np.complex128)
    b = np.zeros(2 * n, dtype=np.complex128)
    a[0:n] += np.asarray(iq_0)
    b[0:n] += np.asarray(iq_1)
    f0 = np.fft.fft(a)
    f1 = np.fft.fft(b)
    xcorr = np.fft.fftshift(np.fft.ifft(f0 * np.conj(f1)))
    imax = int(np.argmax(np.abs(xcorr)))
    xprev = np.abs(xcorr[imax - 1])
    xbest = np.abs(xcorr[imax])
    xnext = np.abs(xcorr[imax + 1])
    delta = 0.5 * (np.log(xprev) - np.log(xnext)) / (np.log(xprev) - 2. * np.log(xbest) + np.log(xnext))
    return (n - (imax + delta)) / rate
