"""Pin the oracle (oracle/fx_oracle.py) against the reference-derived golden vectors.

The expected values in tests/golden/ were produced by the *unmodified* reference
(/root/reference/effex/effex.py) imported through oracle/ref_standins.py — see oracle/make_golden.py.
These tests run without a GPU and without /root/reference.
"""
import hashlib
import io

import numpy as np
import pytest

import fx_oracle
import golden_inputs as gi
from effex_amd.window import design_window


def test_window_matches_reference_expression(golden):
    meta, _ = golden
    for nbins, ntaps in gi.WINDOW_CASES:
        g = meta["window"]["%d_%d" % (nbins, ntaps)]
        w = design_window(ntaps, nbins)
        assert len(w) == nbins * ntaps
        np.testing.assert_allclose(w.sum(), g["sum"], rtol=1e-13)
        np.testing.assert_allclose(w.max(), g["max"], rtol=1e-13)
        np.testing.assert_allclose(w.min(), g["min"], rtol=1e-12)
        assert int(np.argmax(w)) == g["argmax"]
        np.testing.assert_allclose(w[g["sample_idx"]], g["samples"], rtol=1e-9, atol=1e-22)


def test_window_constants_of_survey():
    # SURVEY.md §2.3 constants for N=4096, T=4
    w = design_window(4, 4096)
    np.testing.assert_allclose(w.sum(), 1.1119704004049782, rtol=1e-13)
    np.testing.assert_allclose(w[8192], 2.704190405881311e-4, rtol=1e-13)
    assert np.abs(w - w[::-1]).max() > 1e-8     # periodic Hamming x symmetric sinc: not symmetric


def test_kat_n4_t2(golden):
    _, arrays = golden
    x, h = gi.kat_input()
    spec = fx_oracle.spectrometer_poly(x, 2, 4, h)
    np.testing.assert_allclose(spec, arrays["kat_spec"], rtol=0, atol=1e-13)
    # the table printed in SURVEY.md §2.3 (channelize_poly layout = spec.T)
    expect = np.array([[2.0 - 1.0j, 12.0 - 8.4j, 26.4 - 22.8j],
                       [0.2 + 0.2j, 0.4 + 0.4j, -2.8 + 0.4j],
                       [0.0 - 0.2j, 0.0 - 0.4j, -1.6 + 1.2j],
                       [-0.6 - 0.2j, -1.2 - 0.4j, -1.2 + 2.8j]])
    np.testing.assert_allclose(spec.T, expect, atol=1e-12)


def test_loop_and_vectorised_channelizer_agree():
    rng = np.random.default_rng(5)
    for n_chans, n_taps, n in ((8, 3, 8 * 7 + 3), (16, 4, 16 * 9), (5, 2, 53), (1, 4, 37)):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        h = rng.standard_normal(n_chans * n_taps)
        a = fx_oracle.channelize_poly_loop(x, h, n_chans)
        b = fx_oracle.channelize_poly(x, h, n_chans)
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


def test_channelizer_rejects_more_than_32_taps():
    with pytest.raises(NotImplementedError):
        fx_oracle.channelize_poly(np.zeros(64 * 40, complex), np.zeros(64 * 33), 64)


def test_reference_tone_cases(golden):
    """The reference's own criterion (tests/test_effex.py:62-89) plus sampled outputs."""
    meta, arrays = golden
    cases = gi.tone_cases()
    assert len(cases) == 32 == len(meta["tones"])
    for idx, case in enumerate(cases):
        num_samp, rate, freq, taps, branches = case
        g = meta["tones"][idx]
        assert g["case"] == list(case)
        spec = fx_oracle.spectrometer_poly(gi.tone_iq(num_samp, rate, freq), taps, branches,
                                           design_window(taps, branches))
        assert list(spec.shape) == g["shape"]
        psd = np.fft.fftshift(np.real(spec * np.conj(spec)).mean(axis=0))
        freqs = np.fft.fftshift(np.fft.fftfreq(len(psd), d=1 / rate))
        assert int(np.argmax(psd)) == g["peak_shifted_bin"]
        assert 100. * abs(freqs[np.argmax(psd)] - freq) / freq < 1.
        rows, cols = gi.spec_sample_indices(spec.shape)
        ref = arrays["tone_samples"][idx]
        np.testing.assert_allclose(spec[rows, cols], ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())


def test_pfb_xcorr_against_reference(golden):
    meta, arrays = golden
    iq = gi.xcorr_input()
    np.testing.assert_array_equal(iq[:, :8], arrays["xcorr_input_head"])
    window = design_window(4, 4096)
    for item in meta["xcorr"]:
        vis = fx_oracle.pfb_xcorr(iq[0].astype(np.complex128), iq[1].astype(np.complex128), 4, 4096, window,
                                  gi.BANDWIDTH, gi.FREQUENCY, item["delay"], item["mode"])
        ref = arrays[item["key"]]
        np.testing.assert_allclose(vis, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())


def test_small_multichunk_rows_and_integration(golden):
    _, arrays = golden
    x = gi.small_input()
    window = design_window(4, gi.SMALL_N)
    rows = np.stack([fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], 4, gi.SMALL_N, window, gi.BANDWIDTH, gi.FREQUENCY, 0,
                                         "SPECTRUM") for c in range(x.shape[0])])
    np.testing.assert_allclose(rows, arrays["small_rows"], rtol=1e-10, atol=1e-18)
    # integrating a batch == averaging the reference's rows (equal n_pts per chunk)
    integ = fx_oracle.fx_integrate(x, gi.SMALL_N, window)
    np.testing.assert_allclose(integ[0], arrays["small_rows"].mean(axis=0), rtol=1e-10, atol=1e-18)


def test_csv_bytes(golden):
    meta, _ = golden
    for mode in ("SPECTRUM", "CONTINUUM"):
        fh = io.StringIO()
        fx_oracle.write_metadata(fh, 1, gi.BANDWIDTH, gi.FREQUENCY, gi.CSV_S, gi.CSV_NBINS, 49.6, mode)
        fx_oracle.write_row(fh, gi.csv_row(mode))
        assert fh.getvalue() == meta["csv"][mode]
    # the consumer that pins the format: np.loadtxt(dtype=complex128) (post_process.py:219, effex.py:798)
    text = meta["csv"]["SPECTRUM"]
    back = np.loadtxt(io.StringIO(text), dtype=np.complex128, delimiter=",", skiprows=2)
    np.testing.assert_allclose(back, gi.csv_row("SPECTRUM"), rtol=1e-15)


def test_delay_estimator(golden):
    meta, _ = golden
    for g in meta["delay"][:7]:      # the 4099-sample cases; the 2^18 ones are covered by make_golden's assert
        iq_0 = gi.noise_iq(g["num_samp"])
        est = fx_oracle.estimate_delay_gaussian(iq_0, np.roll(iq_0, g["offset"]), gi.DELAY_RATE)
        np.testing.assert_allclose(est, g["est"], rtol=1e-9, atol=1e-13)
        assert abs(g["offset"] - est * gi.DELAY_RATE) < 0.5


def test_other_nfft_rows_match_reference(golden):
    """The reference's constructor + _run_task at --nfft 1024 / 2048 / 8192 and at 1000 / 96 / 997 / 1536 channels (effex.py:733-739,
    778: a free integer), ragged num_samp."""
    meta, arrays = golden
    for case in meta["nfft"]:
        nbins, num_samp, chunks, delay = case["nbins"], case["num_samp"], case["chunks"], case["delay"]
        x = gi.nfft_input(nbins, num_samp, chunks)
        w = design_window(4, nbins)
        for c in range(chunks):
            row = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], 4, nbins, w, gi.BANDWIDTH, gi.FREQUENCY, delay, "SPECTRUM")
            np.testing.assert_allclose(row, arrays[case["key"]][c], rtol=1e-10, atol=1e-18)


def test_stale_window_after_nbins_change_matches_reference(golden):
    """nbins set after construction (tests/test_effex.py:142-144): the reference keeps the constructor's 4 * 4096-tap
    window (effex.py:126-127) and channelize_poly derives len(h) / nbins taps from it — 8 taps at 2048 bins, 2 at 8192,
    16 (of the first 16 000 coefficients) at 1000."""
    meta, arrays = golden
    w = design_window(4, 4096)
    for case in meta["stale_nbins"]:
        nbins, num_samp, chunks, delay = case["nbins"], case["num_samp"], case["chunks"], case["delay"]
        assert case["window_len"] == len(w) and case["ntaps_effective"] == len(w) // nbins
        x = gi.stale_input(nbins, num_samp, chunks)
        for c in range(chunks):
            row = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], 4, nbins, w, gi.BANDWIDTH, gi.FREQUENCY, delay, "SPECTRUM")
            np.testing.assert_allclose(row, arrays[case["key"]][c], rtol=1e-10, atol=1e-18)


@pytest.mark.parametrize("nchan,ntaps,n_samples", [(8, 3, 8 * 9), (16, 4, 16 * 7 + 5), (64, 1, 64 * 4), (32, 8, 32 * 20 + 31),
                                                   (256, 4, 256 * 6 + 100)])
def test_channelizer_is_the_textbook_analysis_filter_bank(nchan, ntaps, n_samples):
    """Independent pin of the restated cusignal.filtering.channelize_poly (call site effex.py:553), from first
    principles rather than from the restatement's own loop form: channel k of a critically sampled analysis filter bank
    is the input filtered with the prototype modulated to that channel, h[n] exp(+2 pi i k n / N), and decimated by N
    (output i taken at sample i N + N - 1).  Random complex input, random real prototype, ragged length.
    What this pins: tap order (h[t N + m] meets x[(i - t) N + N - 1 - m]), branch reversal, the zero history before
    sample 0, the dropped tail, the unit scale and the sign of the frequency axis.  (The reference's own tone test pins the
    frequency sign a second time; the output conjugation convention conj(fft(conj(.))) is equivalent to this +i kernel for
    the real prototype filters the reference designs, effex.py:126-127.)"""
    from scipy.signal import lfilter
    rng = np.random.default_rng(nchan * 100 + ntaps)
    x = rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)
    h = rng.standard_normal(ntaps * nchan)
    n_pts = n_samples // nchan
    n = np.arange(len(h))
    ref = np.empty((nchan, n_pts), dtype=np.complex128)
    for k in range(nchan):
        y = lfilter(h * np.exp(2j * np.pi * k * n / nchan), 1.0, x)
        ref[k] = y[nchan - 1::nchan][:n_pts]
    scale = np.abs(ref).max()
    assert np.abs(fx_oracle.channelize_poly(x, h, nchan) - ref).max() < 1e-11 * scale
    if n_samples * nchan <= 1 << 16:       # the literal shift-register loop is slow
        assert np.abs(fx_oracle.channelize_poly_loop(x, h, nchan) - ref).max() < 1e-11 * scale
