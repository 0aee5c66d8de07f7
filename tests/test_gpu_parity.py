"""Parity of the HIP path (through the C ABI) with the oracle and the reference-derived goldens.

Tolerances (float32 pipeline vs the float64 oracle on identical complex64 input):
  * F-stage spectra:      max|d| <= 2e-6 * max|spec|   (observed ~3e-7)
  * per-chunk visibility: max|d| <= 1e-5 * max|vis|    (numpy c64-vs-c128 alone is 2.7e-7)
  * integrations:         max|d| <= 1e-5 * max|vis|    (per-chunk float32 sums, float64 across chunks)
"""
import ctypes
import os

import numpy as np
import pytest

import fx_oracle
import golden_inputs as gi
from effex_amd import synth
from effex_amd.window import design_window

pytestmark = pytest.mark.gpu

from tolerances import TOL_CONT, TOL_SPEC, TOL_SPEC_ANY, TOL_TONE, TOL_VIS      # bounds from measurement, never above 2e-6 / 1e-5


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def plan_mod(torch):
    from effex_amd import plan
    return plan


def rel_err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


# --------------------------------------------------------------------------------------------
# F-stage: _spectrometer_poly (effex.py:530-555)
# --------------------------------------------------------------------------------------------
def test_kat_n4_t2(plan_mod, golden):
    _, arrays = golden
    x, h = gi.kat_input()
    with plan_mod.FxPlan(1, 4, 2, len(x), window=h) as p:
        spec = p.channelize(x.astype(np.complex64))[0]
    assert spec.shape == (3, 4)
    assert rel_err(spec, arrays["kat_spec"]) < TOL_SPEC


def test_reference_tone_cases(plan_mod, torch, golden):
    """The reference's 32 spectrometer cases (tests/test_effex.py:62-89) through the drop-in method."""
    from effex_amd.correlator import Correlator, SyntheticSource
    meta, arrays = golden
    cor = Correlator(source=SyntheticSource())
    try:
        for idx, (num_samp, rate, freq, taps, branches) in enumerate(gi.tone_cases()):
            iq = gi.tone_iq(num_samp, rate, freq)
            window = design_window(taps, branches)
            spec = cor._spectrometer_poly(iq, taps, branches, window)
            g = meta["tones"][idx]
            assert list(spec.shape) == g["shape"]
            psd = np.fft.fftshift(np.real(spec * np.conj(spec)).mean(axis=0))
            freqs = np.fft.fftshift(np.fft.fftfreq(len(psd), d=1 / rate))
            assert 100. * abs(freqs[np.argmax(psd)] - freq) / freq < 1.          # the reference's criterion
            assert int(np.argmax(psd)) == g["peak_shifted_bin"]
            rows, cols = gi.spec_sample_indices(spec.shape)
            ref = arrays["tone_samples"][idx]
            scale = np.abs(fx_oracle.spectrometer_poly(iq.astype(np.complex64), taps, branches, window)).max()
            assert np.abs(spec[rows, cols] - ref).max() < TOL_TONE * scale, (idx, g["case"])
    finally:
        cor.close()


@pytest.mark.parametrize("nchan,ntaps,num_samp", [(4096, 4, 4096 * 6 + 5), (2048, 32, 2048 * 40), (512, 4, 8192),
                                                  (1, 4, 3000), (2, 3, 101), (96, 5, 96 * 20 + 7), (8192, 2, 8192 * 3),
                                                  (16384, 1, 16384 * 2)])
def test_channelize_matches_oracle(plan_mod, torch, nchan, ntaps, num_samp):
    rng = np.random.default_rng(nchan + ntaps)
    x = synth.synth_iq(11, 2, 1, num_samp)[:, 0]
    h = design_window(ntaps, nchan) if nchan > 1 else rng.standard_normal(ntaps)
    with plan_mod.FxPlan(1, nchan, ntaps, num_samp, window=h) as p:
        host = p.channelize(x)
        dev = p.channelize(torch.from_numpy(x).cuda()).cpu().numpy()
    np.testing.assert_array_equal(host, dev)
    for s in range(2):
        ref = fx_oracle.spectrometer_poly(x[s], ntaps, nchan, h)
        assert host[s].shape == ref.shape
        assert rel_err(host[s], ref) < (TOL_SPEC_ANY if (nchan & (nchan - 1)) else TOL_SPEC)


@pytest.mark.parametrize("n_streams", [1, 2, 5])
def test_channelize_default_shape_uses_f_only_kernel(plan_mod, torch, n_streams):
    """nchan 4096 / ntaps 4 (the constructor default): pairs of streams go through the F-only tiled kernel (an odd
    last stream as a half-empty pair); it must agree with the oracle and with the generic kernels."""
    num_samp = 4096 * 9 + 100
    x = synth.synth_iq(17, n_streams, 1, num_samp)[:, 0]
    window = design_window(4, 4096)
    with plan_mod.FxPlan(1, 4096, 4, num_samp) as p, plan_mod.FxPlan(1, 4096, 4, num_samp, path="generic") as g:
        spec = p.channelize(torch.from_numpy(x).cuda()).cpu().numpy()
        spec_h = p.channelize(x)
        spec_g = g.channelize(x)
    np.testing.assert_array_equal(spec, spec_h)
    for s_ in range(n_streams):
        ref = fx_oracle.spectrometer_poly(x[s_], 4, 4096, window)
        assert rel_err(spec[s_], ref) < TOL_SPEC
        assert rel_err(spec[s_], spec_g[s_]) < TOL_SPEC


def test_more_than_32_taps_raises(plan_mod):
    with pytest.raises(NotImplementedError):
        plan_mod.FxPlan(1, 64, 33, 64 * 64, window=np.zeros(64 * 33))


# --------------------------------------------------------------------------------------------
# F+X: _pfb_xcorr / _run_task (effex.py:490-527)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["fused", "generic"])
def test_pfb_xcorr_against_reference_goldens(plan_mod, torch, golden, path):
    meta, arrays = golden
    iq = gi.xcorr_input()
    xd = torch.from_numpy(iq[None]).cuda()
    with plan_mod.FxPlan(2, 4096, 4, 2 ** 18, path=path) as p:
        assert p.path == path
        for item in meta["xcorr"]:
            p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, item["delay"])
            ref = arrays[item["key"]]
            if item["mode"] == "SPECTRUM":
                vis = p.fx_rows(xd, "SPECTRUM").cpu().numpy()[0, 0]
                assert rel_err(vis, ref) < TOL_VIS, item
                # single-chunk integration == the row
                p.acc_reset()
                p.fx_accumulate(xd)
                assert rel_err(p.finalize("SPECTRUM")[0], ref) < TOL_VIS
            else:
                vis = p.fx_rows(xd, item["mode"], gi.BANDWIDTH).cpu().numpy()[0, 0]
                assert abs(vis - ref) < TOL_CONT * abs(ref) + 1e-8 * np.abs(arrays["xcorr_SPECTRUM_0"]).max() / gi.BANDWIDTH
                p.acc_reset()
                p.fx_accumulate(xd)
                integ = p.finalize(item["mode"], gi.BANDWIDTH)[0]
                assert abs(integ - ref) < TOL_CONT * abs(ref) + 1e-8 * np.abs(arrays["xcorr_SPECTRUM_0"]).max() / gi.BANDWIDTH


def test_drop_in_run_task(torch, golden):
    """Correlator._run_task() on staged buffers == the reference's _run_task() (all three modes)."""
    from effex_amd.correlator import Correlator, SyntheticSource
    meta, arrays = golden
    iq = gi.xcorr_input()
    cor = Correlator(source=SyntheticSource())
    try:
        for item in meta["xcorr"]:
            cor.mode = item["mode"]
            cor.calibrated_delay = item["delay"]
            cor.gpu_iq_0, cor.gpu_iq_1 = iq[0], iq[1]
            vis = cor._run_task()
            ref = arrays[item["key"]]
            if item["mode"] == "SPECTRUM":
                assert vis.shape == (4096,) and vis.dtype == np.complex128
                assert rel_err(vis, ref) < TOL_VIS
            else:
                assert np.ndim(vis) == 0
                assert abs(vis - ref) < TOL_CONT * abs(ref)
    finally:
        cor.close()


def test_drop_in_nbins_changed_after_construction(torch, golden):
    """cor.nbins = ... after construction (tests/test_effex.py:142-144): like the reference, the drop-in keeps the
    window designed for the constructor's nbins (effex.py:126-127, 287-294) and channelises with len(window) / nbins
    taps of it — against rows the reference's own _run_task produced that way (tests/golden)."""
    from effex_amd.correlator import Correlator, SyntheticSource
    meta, arrays = golden
    for case in meta["stale_nbins"]:
        nbins, num_samp, chunks, delay = case["nbins"], case["num_samp"], case["chunks"], case["delay"]
        x = gi.stale_input(nbins, num_samp, chunks)
        cor = Correlator(source=SyntheticSource())
        try:
            window_before = cor.window.copy()
            cor.nbins = nbins
            cor.num_samp = num_samp
            assert cor.nbins == nbins and np.array_equal(cor.window, window_before)
            cor.calibrated_delay = delay
            for c in range(chunks):
                cor.gpu_iq_0, cor.gpu_iq_1 = x[c, 0], x[c, 1]
                vis = cor._run_task()
                assert vis.shape == (nbins,)
                assert rel_err(vis, arrays[case["key"]][c]) < TOL_VIS, (nbins, c)
            cor.nbins = 2 ** 15        # more bins than window taps: channelize_poly would get zero taps
            with pytest.raises(ValueError):
                cor._run_task()
        finally:
            cor.close()


@pytest.mark.parametrize("nbins", [1000, 1536, 997, 12000])
def test_drop_in_any_resolution(torch, nbins):
    """Correlator(nbins=...) with a channel count that is not a power of two (effex.py:733-739: --resolution is a free
    integer): _run_task() on staged buffers against the oracle's pfb_xcorr, both output modes."""
    from effex_amd.correlator import Correlator, SyntheticSource
    num_samp = 2 ** 16
    x = synth.synth_iq(31 + nbins, 1, 2, num_samp, delays=[0, 3])[0]
    cor = Correlator(source=SyntheticSource(), nbins=nbins, num_samp=num_samp)
    try:
        assert cor.ntaps * nbins == len(cor.window)
        cor.calibrated_delay = 1.25e-6
        cor.gpu_iq_0, cor.gpu_iq_1 = x[0], x[1]
        for mode in ("SPECTRUM", "CONTINUUM"):
            cor.mode = mode
            vis = cor._run_task()
            ref = fx_oracle.pfb_xcorr(x[0], x[1], cor.ntaps, nbins, cor.window, cor.bandwidth, cor.frequency, 1.25e-6, mode)
            if mode == "SPECTRUM":
                assert vis.shape == (nbins,) and rel_err(vis, ref) < TOL_VIS
            else:
                assert abs(vis - ref) < TOL_CONT * abs(ref) + 1e-9 * np.abs(x).max() ** 2
    finally:
        cor.close()


@pytest.mark.parametrize("path", ["tiled", "generic"])
def test_small_multichunk_rows(plan_mod, torch, golden, path):
    _, arrays = golden
    x = gi.small_input()
    with plan_mod.FxPlan(2, gi.SMALL_N, 4, gi.SMALL_S, path=path) as p:
        assert p.path == path
        rows = p.fx_rows(x, "SPECTRUM")                      # host buffers in, host rows out
        assert rows.shape == (gi.SMALL_CHUNKS, 1, gi.SMALL_N)
        assert rel_err(rows[:, 0], arrays["small_rows"]) < TOL_VIS
        p.fx_accumulate(x)
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ[0], arrays["small_rows"].mean(axis=0)) < TOL_VIS


def _dc(x):
    x = x.astype(np.complex128)
    return ((x.real - x.real.mean()) + 1j * (x.imag - x.imag.mean())).astype(np.complex64)      # effex.py:394-395


@pytest.mark.parametrize("calibrate", [True, False])
def test_state_machine_writes_reference_csv(tmp_path, torch, golden, calibrate):
    """End to end: source -> staging (DC removal) -> [calibration] -> HIP path -> csv the reference's reader loads."""
    from effex_amd.correlator import ArraySource, Correlator
    x = gi.small_input()
    path = str(tmp_path / "vis.csv")
    cor = Correlator(num_samp=gi.SMALL_S, nbins=gi.SMALL_N, source=ArraySource(x), output_file=path,
                     calibrate=calibrate)
    first = 1 if calibrate else 0            # the reference's CALIBRATE state consumes the first chunk pair
    assert cor.run_state_machine() == gi.SMALL_CHUNKS - first
    assert cor.state == 'OFF'
    lines = open(path).read().split('\n')
    assert lines[0] == fx_oracle.metadata_header(1, 2.4e6, 1.4204e9, gi.SMALL_S, gi.SMALL_N, 49.6, 'SPECTRUM').strip()
    data = np.loadtxt(path, dtype=np.complex128, delimiter=',', skiprows=2)      # post_process.py:219
    assert data.shape == (gi.SMALL_CHUNKS - first, gi.SMALL_N)
    delay = 0.0
    if calibrate:
        delay = fx_oracle.estimate_delay_gaussian(_dc(x[0, 0]), _dc(x[0, 1]), 2.4e6)
        assert abs(cor.calibrated_delay - delay) * 2.4e6 < 2e-3                  # samples
        assert abs(delay * 2.4e6 - 3) < 0.5          # the synthetic source delays antenna 1 by 3 samples (roll +3)
        delay = cor.calibrated_delay
    window = design_window(4, gi.SMALL_N)
    for c in range(first, gi.SMALL_CHUNKS):
        ref = fx_oracle.pfb_xcorr(_dc(x[c, 0]), _dc(x[c, 1]), 4, gi.SMALL_N, window, 2.4e6, 1.4204e9, delay, 'SPECTRUM')
        assert rel_err(data[c - first], ref) < TOL_VIS


@pytest.mark.parametrize("mode", ["SPECTRUM", "CONTINUUM"])
def test_state_machine_binary_sidecar_equals_csv(tmp_path, torch, mode):
    """Correlator(output_format='bin'): the same run written as a binary sidecar and turned into text
    by rowsink.to_csv gives the bytes of the csv run (effex.py:667-696) -- the rows are the same device results."""
    from effex_amd import rowsink
    from effex_amd.correlator import ArraySource, Correlator
    x = gi.small_input()
    paths = {fmt: str(tmp_path / ("vis." + ext)) for fmt, ext in (("csv", "csv"), ("bin", "fxb"))}
    for fmt in ("csv", "bin"):
        cor = Correlator(num_samp=gi.SMALL_S, nbins=gi.SMALL_N, source=ArraySource(x), output_file=paths[fmt], mode=mode,
                         output_format=fmt)
        assert cor.run_state_machine() == gi.SMALL_CHUNKS - 1
    side = rowsink.RowFile(paths["bin"])
    assert side.rows.shape == (gi.SMALL_CHUNKS - 1, gi.SMALL_N if mode == "SPECTRUM" else 1)
    assert side.row_dtype == (np.complex64 if mode == "SPECTRUM" else np.complex128)
    back = str(tmp_path / "back.csv")
    assert rowsink.to_csv(paths["bin"], back) == gi.SMALL_CHUNKS - 1
    assert open(back, "rb").read() == open(paths["csv"], "rb").read()


def test_pipeline_pops_into_a_mapped_row_file(tmp_path, torch):
    """FxPipeline.pop(out=...) into a window of the sidecar (BinSink.reserve / commit): rows as the blocking path gives."""
    from effex_amd import rowsink
    num_samp, chunks, n_batches = 4096 * 4, 3, 4
    x = synth.synth_iq(18, chunks * n_batches, 2, num_samp).reshape(n_batches, chunks, 2, num_samp)
    from effex_amd import plan as plan_mod
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        ref = np.concatenate([p.fx_rows(x[b], "SPECTRUM") for b in range(n_batches)])[:, 0]
        path = str(tmp_path / "rows.fxb")
        with rowsink.BinSink(path, "run_time:1", None, 4096, np.complex64) as sink, \
                plan_mod.FxPipeline(p, chunks, depth=2, mode="SPECTRUM") as pipe:
            view = sink.reserve(chunks * n_batches)
            pipe.push(x[0])
            for b in range(n_batches):
                if b + 1 < n_batches:
                    pipe.push(x[b + 1])
                pipe.pop(out=view[b * chunks:(b + 1) * chunks])
                sink.commit(chunks)
            with pytest.raises(ValueError):
                pipe.pop(out=np.empty((chunks, 1, 4095), dtype=np.complex64))
    np.testing.assert_array_equal(np.asarray(rowsink.RowFile(path).rows), ref)


def test_state_machine_on_a_byte_source(tmp_path, torch):
    """A source that hands over the receivers' bytes (pyrtlsdr format='bytes') writes the same csv as the same data
    handed over as samples: calibration on the first pair, then one fused convert + de-mean + F+X call per pair."""
    from effex_amd.correlator import ArraySource, Correlator
    n_chunks, num_samp, nbins = 5, 4096 * 6, 4096
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, size=(n_chunks, num_samp + 3, 2), dtype=np.uint8)
    own = rng.integers(0, 256, size=(n_chunks, 2, num_samp, 2), dtype=np.uint8)
    u8 = np.empty((n_chunks, 2, num_samp, 2), np.uint8)
    u8[:, 0] = base[:, 3:] // 2 + own[:, 0] // 2          # common signal, antenna 1 three samples late
    u8[:, 1] = base[:, :-3] // 2 + own[:, 1] // 2
    samples = fx_oracle.u8_to_complex(u8).astype(np.complex128)
    rows = {}
    for name, chunks in (("bytes", u8), ("samples", samples)):
        path = str(tmp_path / (name + ".csv"))
        cor = Correlator(num_samp=num_samp, nbins=nbins, source=ArraySource(chunks), output_file=path)
        assert cor.run_state_machine() == n_chunks - 1
        rows[name] = (np.loadtxt(path, dtype=np.complex128, delimiter=',', skiprows=2), cor.calibrated_delay)
    assert abs(rows["bytes"][1] - rows["samples"][1]) * 2.4e6 < 1e-3          # same calibrated delay (samples)
    assert abs(rows["bytes"][1] * 2.4e6 - 3) < 0.5
    assert rel_err(rows["bytes"][0], rows["samples"][0]) < TOL_VIS


def test_state_machine_on_recorded_files(tmp_path, torch):
    """FileSource: two rtl_sdr-style recordings (raw interleaved uint8 I,Q) and the same data as raw complex64 files
    write the csv an in-memory source of the same chunk pairs writes; a trailing partial chunk is dropped."""
    from effex_amd.correlator import ArraySource, Correlator, FileSource
    n_chunks, num_samp, nbins = 4, 4096 * 5, 4096
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, size=(n_chunks * num_samp + 500 + 2, 2), dtype=np.uint8)
    own = rng.integers(0, 256, size=(2, n_chunks * num_samp + 500, 2), dtype=np.uint8)
    streams = [base[2:] // 2 + own[0] // 2, base[:-2] // 2 + own[1] // 2]          # antenna 1 two samples late, 500 spare samples
    for a in range(2):
        streams[a].tofile(str(tmp_path / ("rx%d.u8" % a)))
        fx_oracle.u8_to_complex(streams[a][None, None])[0, 0].astype(np.complex64).tofile(str(tmp_path / ("rx%d.c64" % a)))
    chunks = np.stack([s[: n_chunks * num_samp].reshape(n_chunks, num_samp, 2) for s in streams], axis=1)
    rows = {}
    sources = {"memory": lambda: ArraySource(chunks),
               "u8": lambda: FileSource(str(tmp_path / "rx0.u8"), str(tmp_path / "rx1.u8"), fmt='u8'),
               "c64": lambda: FileSource(str(tmp_path / "rx0.c64"), str(tmp_path / "rx1.c64"), fmt='c64')}
    for name, make in sources.items():
        path = str(tmp_path / (name + ".csv"))
        src = make()
        cor = Correlator(num_samp=num_samp, nbins=nbins, source=src, output_file=path)
        assert cor.run_state_machine() == n_chunks - 1          # first pair calibrates; the 500 spare samples are dropped
        assert src.closed
        rows[name] = (np.loadtxt(path, dtype=np.complex128, delimiter=',', skiprows=2), cor.calibrated_delay)
    np.testing.assert_array_equal(rows["u8"][0], rows["memory"][0])
    assert rows["u8"][1] == rows["memory"][1] and abs(rows["u8"][1] * 2.4e6 - 2) < 0.5
    # complex64 files: the float32-rounded samples move the calibrated delay by ~1e-14 s, i.e. the rot phase
    # 2 pi f tau (f = 1.42 GHz) by ~1e-4 rad
    assert abs(rows["c64"][1] - rows["memory"][1]) < 1e-12 and rel_err(rows["c64"][0], rows["memory"][0]) < 1e-3


@pytest.mark.parametrize("mode", ["SPECTRUM", "CONTINUUM"])
def test_batched_run_writes_the_rows_of_the_per_pair_run(tmp_path, torch, mode):
    """Correlator(batch=4): the RUN state takes four chunk pairs per device call through the host-fed pipeline -- byte
    recordings, complex64 recordings and an in-memory source, csv and binary sidecar -- and writes the rows the
    one-call-per-pair loop writes (float32 summation order aside), the short last batch and the dropped partial chunk
    included; calibration is the per-pair loop's."""
    from effex_amd import rowsink
    from effex_amd.correlator import ArraySource, Correlator, FileSource
    n_chunks, num_samp, nbins = 11, 4096 * 6, 4096          # 1 calibration pair + 10 rows = two batches of four + two
    rng = np.random.default_rng(23)
    base = rng.integers(0, 256, size=(n_chunks * num_samp + 300 + 3, 2), dtype=np.uint8)
    own = rng.integers(0, 256, size=(2, n_chunks * num_samp + 300, 2), dtype=np.uint8)
    streams = [base[3:] // 2 + own[0] // 2, base[:-3] // 2 + own[1] // 2]
    for a in range(2):
        streams[a].tofile(str(tmp_path / ("rx%d.u8" % a)))
        fx_oracle.u8_to_complex(streams[a][None, None])[0, 0].astype(np.complex64).tofile(str(tmp_path / ("rx%d.c64" % a)))
    chunks_c = np.stack([fx_oracle.u8_to_complex(s[: n_chunks * num_samp].reshape(n_chunks, num_samp, 2)[None])[0]
                         for s in streams], axis=1)
    sources = {"u8": lambda: FileSource(str(tmp_path / "rx0.u8"), str(tmp_path / "rx1.u8"), fmt='u8'),
               "c64": lambda: FileSource(str(tmp_path / "rx0.c64"), str(tmp_path / "rx1.c64"), fmt='c64'),
               "memory": lambda: ArraySource(chunks_c)}
    skip = 2 if mode == "SPECTRUM" else 1
    for name, make in sources.items():
        got = {}
        for batch, fmt in ((1, 'csv'), (4, 'csv'), (4, 'bin'), (16, 'bin')):
            path = str(tmp_path / ("%s_%d.%s" % (name, batch, fmt)))
            src = make()
            cor = Correlator(num_samp=num_samp, nbins=nbins, source=src, output_file=path, mode=mode, output_format=fmt,
                             batch=batch)
            assert cor.run_state_machine() == n_chunks - 1, (name, batch, fmt)
            assert src.closed and cor.state == 'OFF'
            if fmt == 'csv':
                rows = np.loadtxt(path, dtype=np.complex128, delimiter=',', skiprows=skip).reshape(n_chunks - 1, -1)
            else:
                rf = rowsink.RowFile(path)
                assert rf.rows.shape[0] == n_chunks - 1
                rows = np.asarray(rf.rows).astype(np.complex128)
            got[(batch, fmt)] = (rows, cor.calibrated_delay)
        ref_rows, ref_delay = got[(1, 'csv')]
        assert abs(ref_delay * 2.4e6 - 3) < 0.5
        for key, (rows, delay) in got.items():
            assert delay == ref_delay, (name, key)
            assert rows.shape == ref_rows.shape
            for c in range(n_chunks - 1):
                assert rel_err(rows[c], ref_rows[c]) < 2e-5, (name, key, c)


def test_delay_calibration_against_reference(plan_mod, torch, golden):
    """The reference's delay tests (tests/test_effex.py:92-121): 14 cases, |k - est*rate| < 0.5 sample and
    |k/rate - est| < 1e-6 s, plus agreement with the reference's own estimate (golden)."""
    from effex_amd.correlator import Correlator, SyntheticSource
    meta, _ = golden
    cor = Correlator(source=SyntheticSource())
    try:
        for g in meta["delay"]:
            iq_0 = gi.noise_iq(g["num_samp"])
            iq_1 = np.roll(iq_0, g["offset"])
            est = cor._estimate_delay_gaussian(iq_0, iq_1, gi.DELAY_RATE)
            assert abs(g["offset"] - est * gi.DELAY_RATE) < 0.5
            assert abs(est - g["est"]) * gi.DELAY_RATE < 2e-3, g
            est2 = cor._estimate_delay(torch.from_numpy(iq_0.astype(np.complex64)).cuda(),
                                       torch.from_numpy(iq_1.astype(np.complex64)).cuda(), gi.DELAY_RATE)
            assert abs(g["offset"] / gi.DELAY_RATE - est2) < 1e-6
        cor.mode = 'TEST'
        iq_0 = gi.noise_iq(4099)
        est = cor._estimate_delay(iq_0, np.roll(iq_0, 5), gi.DELAY_RATE)
        assert abs(est - (5 / gi.DELAY_RATE - cor.test_delay_offset)) < 1e-6
    finally:
        cor.close()


@pytest.mark.parametrize("n", [3, 100, 2048, 4096, 5000, 8192, 16384, 40000, 1 << 17])
def test_delay_transform_every_pass_mix(plan_mod, n):
    """log2 of the padded length runs through 3 .. 18: every mix of radix-16 passes and a radix-8 / 4 / 2 tail
    (k_delay.h), against the oracle's float64 estimate of the same streams."""
    rng = np.random.default_rng(n)
    iq_0 = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    shift = 1 if n < 8 else 1 + n // 7 % 11
    iq_1 = np.roll(iq_0, shift) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ref = fx_oracle.estimate_delay_gaussian(iq_0.astype(np.complex128), iq_1.astype(np.complex128), gi.DELAY_RATE)
    with plan_mod.FxPlan(2, 512, 4, 4096) as p:
        est = p.estimate_delay(iq_0, iq_1, gi.DELAY_RATE)
    assert abs(est - ref) * gi.DELAY_RATE < 2e-3, (n, est * gi.DELAY_RATE, ref * gi.DELAY_RATE)
    if n >= 100:
        assert abs(est * gi.DELAY_RATE - shift) < 0.5


# --------------------------------------------------------------------------------------------
# batches, ragged sizes, multi-antenna, continuum streaming limit
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path,n_chunks,num_samp", [("fused", 5, 4096 * 7 + 123), ("fused", 300, 4096 * 4),
                                                    ("fused", 3, 4096), ("generic", 5, 4096 * 7 + 123)])
def test_batched_integration_matches_oracle(plan_mod, torch, path, n_chunks, num_samp):
    x = synth.synth_iq(99, n_chunks, 2, num_samp)
    window = design_window(4, 4096)
    rot = plan_mod.rot_table(4096, gi.BANDWIDTH, gi.FREQUENCY, 3e-7)
    ref_chunks = min(n_chunks, 12)            # oracle on a prefix for the rows; integration by property below
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, 4096, 4, num_samp, path=path) as p:
        p.set_rot(rot)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        for c in range(ref_chunks):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], 4, 4096, window, gi.BANDWIDTH, gi.FREQUENCY, 3e-7, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        # integrate in two uneven calls; must equal the mean of the rows (same n_pts per chunk)
        p.fx_accumulate(xd[: n_chunks // 3 + 1])
        p.fx_accumulate(xd[n_chunks // 3 + 1:])
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ[0], rows[:, 0].astype(np.complex128).mean(axis=0)) < 2e-6
        cont_rows = p.fx_rows(xd, "CONTINUUM", gi.BANDWIDTH).cpu().numpy()
        np.testing.assert_allclose(cont_rows[:, 0], rows[:, 0].astype(np.complex128).mean(axis=1) / gi.BANDWIDTH,
                                   rtol=2e-5, atol=1e-7 * np.abs(cont_rows).max())


@pytest.mark.parametrize("n_chunks,frames,extra", [(259, 5, 0), (3, 300, 77), (700, 2, 0), (257, 1, 9), (1, 64, 0),
                                                   (517, 3, 1)])
def test_fused_frame_ranges_ignore_chunk_boundaries(plan_mod, torch, n_chunks, frames, extra):
    """The headline kernel hands every workgroup an equal range of the launch's frames (n_chunks % CUs != 0, ranges
    that start mid-chunk and reload PFB history, several workgroups inside one chunk, chunks shorter than a range):
    per-chunk rows against the oracle and against the generic kernels, integration (rows of several chunks +
    leading-part rows) against the float64 mean of the rows, SPECTRUM and CONTINUUM, complex64 and uint8 input."""
    num_samp = 4096 * frames + extra
    x = synth.synth_iq(4242, n_chunks, 2, num_samp)
    window = design_window(4, 4096)
    rot = plan_mod.rot_table(4096, gi.BANDWIDTH, gi.FREQUENCY, 2e-7)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, 4096, 4, num_samp, path="fused") as p, \
            plan_mod.FxPlan(2, 4096, 4, num_samp, path="generic") as g:
        p.set_rot(rot)
        g.set_rot(rot)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        rows_g = g.fx_rows(xd, "SPECTRUM").cpu().numpy()
        assert rel_err(rows, rows_g) < 4e-6
        for c in sorted({0, 1 % n_chunks, n_chunks // 2, n_chunks - 1}):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], 4, 4096, window, gi.BANDWIDTH, gi.FREQUENCY, 2e-7, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        p.fx_accumulate(xd)
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ[0], rows[:, 0].astype(np.complex128).mean(axis=0)) < 4e-6
        p.fx_accumulate(xd[: n_chunks // 2])          # uneven calls: every launch has its own ranges
        p.fx_accumulate(xd[n_chunks // 2:])
        assert rel_err(p.finalize("SPECTRUM")[0], integ[0]) < 4e-6
        cont = p.fx_rows(xd, "CONTINUUM", gi.BANDWIDTH).cpu().numpy()
        np.testing.assert_allclose(cont[:, 0], rows[:, 0].astype(np.complex128).mean(axis=1) / gi.BANDWIDTH,
                                   rtol=2e-5, atol=1e-7 * np.abs(cont).max())
        # the same ranges on the uint8-ingest variant of the kernel
        b = np.random.default_rng(5).integers(0, 256, size=(min(n_chunks, 300), 2, num_samp, 2), dtype=np.uint8)
        bd = torch.from_numpy(b).cuda()
        rows_b = p.fx_rows_u8(bd, "SPECTRUM", remove_dc=True).cpu().numpy()
        rows_bg = g.fx_rows_u8(bd, "SPECTRUM", remove_dc=True).cpu().numpy()
        assert rel_err(rows_b, rows_bg) < 1e-5
        p.fx_accumulate_u8(bd, remove_dc=True)
        assert rel_err(p.finalize("SPECTRUM")[0], rows_b[:, 0].astype(np.complex128).mean(axis=0)) < 1e-5


def test_randomized_launch_shapes_fused_vs_generic(plan_mod, torch):
    """60 random (chunk count, frames per chunk, ragged tail) draws on the headline kernel — launches that are all
    tail (fewer chunks than CUs), whole rounds plus a tail, chunks of one frame, ranges shorter and longer than a chunk,
    rows of 1 … 64 chunks in the integration — against the generic kernels, which share no code with it."""
    rng = np.random.default_rng(20261003)
    with plan_mod.FxPlan(2, 4096, 4, 4096) as probe:
        n_cu = probe.info["cu_count"]
    for case in range(60):
        frames = int(rng.choice([1, 1, 2, 3, 5, 8, 13, 40]))
        n_chunks = int(rng.choice([1, 2, 7, n_cu - 1, n_cu, n_cu + 1, 2 * n_cu + 3, int(rng.integers(1, 900))]))
        if n_chunks * frames > 6000:
            n_chunks = max(1, 6000 // frames)
        num_samp = 4096 * frames + int(rng.integers(0, 4096))
        x = torch.from_numpy(synth.synth_iq(3000 + case, n_chunks, 2, num_samp)).cuda()
        tag = (case, n_chunks, frames, num_samp)
        with plan_mod.FxPlan(2, 4096, 4, num_samp, path="fused") as f, \
                plan_mod.FxPlan(2, 4096, 4, num_samp, path="generic") as g:
            rf, rg = f.fx_rows(x).cpu().numpy(), g.fx_rows(x).cpu().numpy()
            assert rel_err(rf, rg) < 4e-6, tag
            f.fx_accumulate(x)
            assert rel_err(f.finalize("SPECTRUM")[0], rf[:, 0].astype(np.complex128).mean(axis=0)) < 4e-6, tag
            if case % 4 == 0:
                u8 = torch.from_numpy(rng.integers(0, 256, size=(n_chunks, 2, num_samp, 2), dtype=np.uint8)).cuda()
                assert rel_err(f.fx_rows_u8(u8).cpu().numpy(), g.fx_rows_u8(u8).cpu().numpy()) < TOL_VIS, tag
            if case % 6 == 0 and n_chunks >= 2:                 # 4 antennas: the F-only variant walks the same ranges
                x4 = x[: n_chunks // 2 * 2].reshape(n_chunks // 2, 4, num_samp).contiguous()
                with plan_mod.FxPlan(4, 4096, 4, num_samp) as m, plan_mod.FxPlan(4, 4096, 4, num_samp, path="generic") as mg:
                    assert m.path == "fused", tag
                    assert rel_err(m.fx_rows(x4).cpu().numpy(), mg.fx_rows(x4).cpu().numpy()) < 4e-6, tag


def test_ten_thousand_frame_accumulation(plan_mod, torch):
    """BASELINE configs[1] integrates 10 000 frames: float32 sums of up to 1 024 spectra per raw row, float64 across rows
    (h_launch.h::fused_unit, k_finish.h::fold_partial_kernel / fold_finish_kernel).  A pool of 25 distinct chunk pairs cycled to 10 400 frames of 16
    spectra must equal the float64 mean of the per-chunk rows, and the oracle's mean over the pool, to 1e-5 of max|vis|
    (SURVEY.md §8d 'parity tolerance to state')."""
    num_samp, pool_n, reps = 4096 * 16, 25, 416
    pool = synth.synth_iq(90210, pool_n, 2, num_samp)
    window = design_window(4, 4096)
    xd = torch.from_numpy(pool).cuda().repeat(reps, 1, 1).contiguous()
    assert xd.shape[0] == 10400
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        assert p.path == "fused"
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 1e-6)
        p.fx_accumulate(xd)
        integ = p.finalize("SPECTRUM")[0]
        rows = p.fx_rows(xd[:pool_n], "SPECTRUM").cpu().numpy()[:, 0].astype(np.complex128)
    assert rel_err(integ, rows.mean(axis=0)) < 2e-6
    ref = np.mean([fx_oracle.pfb_xcorr(pool[c, 0], pool[c, 1], 4, 4096, window, gi.BANDWIDTH, gi.FREQUENCY, 1e-6,
                                       "SPECTRUM") for c in range(pool_n)], axis=0)
    assert rel_err(integ, ref) < TOL_VIS


def test_continuum_reference_semantics_full_size(plan_mod, torch):
    """BASELINE configs[2](ii): reference CONTINUUM (effex.py:523-524) at N = 4096 with num_samp = 2^20 (the reference
    clamps num_samp to 2^18, effex.py:282-283; lifted here): oracle on one chunk pair, the rest by property."""
    num_samp, n_chunks = 2 ** 20, 3
    x = synth.synth_iq(1234, n_chunks, 2, num_samp)
    window = design_window(4, 4096)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, 4096, 4, num_samp, path="fused") as p, plan_mod.FxPlan(2, 4096, 4, num_samp) as a:
        for q in (p, a):
            q.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 1e-6)
        cont = p.fx_rows(xd, "CONTINUUM", gi.BANDWIDTH).cpu().numpy()
        cont_a = a.fx_rows(xd, "CONTINUUM", gi.BANDWIDTH).cpu().numpy()       # default plan
        spec = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        p.fx_accumulate(xd)
        integ = p.finalize("CONTINUUM", gi.BANDWIDTH)
    ref = fx_oracle.pfb_xcorr(x[1, 0], x[1, 1], 4, 4096, window, gi.BANDWIDTH, gi.FREQUENCY, 1e-6, "CONTINUUM")
    assert abs(cont[1, 0] - ref) < TOL_CONT * abs(ref)
    assert abs(cont_a[1, 0] - ref) < TOL_CONT * abs(ref)
    np.testing.assert_allclose(cont[:, 0], spec[:, 0].astype(np.complex128).mean(axis=1) / gi.BANDWIDTH, rtol=2e-5)
    np.testing.assert_allclose(integ[0], cont[:, 0].mean(), rtol=2e-5)


def test_eight_antennas_full_size(plan_mod, torch):
    """BASELINE configs[4] at its own size: 8 antennas, 28 baselines, nchan 4096, num_samp 262144: oracle
    (fx_integrate) on one chunk, linearity and conjugate symmetry across the antenna order on the batch."""
    n_ant, num_samp, n_chunks = 8, 2 ** 18, 3
    x = synth.synth_iq(808, n_chunks, n_ant, num_samp)
    window = design_window(4, 4096)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(n_ant, 4096, 4, num_samp) as p:
        assert p.path == "fused" and p.n_baselines == 28
        rows = p.fx_rows(xd).cpu().numpy().astype(np.complex128)
        p.fx_accumulate(xd)
        integ = p.finalize("SPECTRUM")
        np.testing.assert_array_equal(p.fx_rows(xd * 2.0).cpu().numpy().astype(np.complex128), 4.0 * rows)
        rev = p.fx_rows(xd.flip(1).contiguous()).cpu().numpy().astype(np.complex128)
    ref = fx_oracle.fx_integrate(x[1:2], 4096, window)
    assert rel_err(rows[1], ref) < TOL_VIS
    assert rel_err(integ, rows.mean(axis=0)) < 2e-6
    # baseline (a, b) of the reversed antenna order is conj of baseline (7 - b, 7 - a) of the original
    pairs = [(a, b) for a in range(n_ant) for b in range(a + 1, n_ant)]
    for idx, (a, b) in enumerate(pairs):
        assert rel_err(rev[:, idx], np.conj(rows[:, pairs.index((n_ant - 1 - b, n_ant - 1 - a))])) < 1e-6


@pytest.mark.parametrize("n_ant,n_chunks", [(16, 3), (32, 2), (40, 2), (64, 1)])
def test_many_antennas_full_size(plan_mod, torch, n_ant, n_chunks):
    """The matrix-core X-engine (k_xmfma.h) at the headline frame size -- nchan 4096, num_samp 262144 -- with 1, 2, 3 and 4
    antenna tiles: one chunk against the oracle (fx_integrate, every baseline), the integration against the float64 mean of
    the rows, exact linearity, and conjugate symmetry under reversal of the antenna order (which moves every baseline to
    another place of another tile pair)."""
    num_samp = 2 ** 18
    x = synth.synth_iq(900 + n_ant, n_chunks, n_ant, num_samp, delays=np.arange(n_ant) % 11)
    window = design_window(4, 4096)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(n_ant, 4096, 4, num_samp) as p:
        assert p.path == "tiled" and p.n_baselines == n_ant * (n_ant - 1) // 2
        rows = p.fx_rows(xd).cpu().numpy().astype(np.complex128)
        out = plan_mod.pinned_empty((p.n_baselines, 4096), np.complex128)
        p.fx_accumulate(xd)
        p.finalize_async("SPECTRUM", out=out)
        integ = p.finalize_wait()
        assert integ is out
        np.testing.assert_array_equal(p.fx_rows(xd * 2.0).cpu().numpy().astype(np.complex128), 4.0 * rows)
        rev = p.fx_rows(xd[:1].flip(1).contiguous()).cpu().numpy().astype(np.complex128)
    ref = fx_oracle.fx_integrate(x[n_chunks - 1:n_chunks], 4096, window)
    assert rel_err(rows[n_chunks - 1], ref) < TOL_VIS
    assert rel_err(integ, rows.mean(axis=0)) < 2e-6
    pairs = {(a, b): idx for idx, (a, b) in enumerate((a, b) for a in range(n_ant) for b in range(a + 1, n_ant))}
    for (a, b), idx in pairs.items():
        assert rel_err(rev[0, idx], np.conj(rows[0, pairs[(n_ant - 1 - b, n_ant - 1 - a)]])) < 1e-6, (a, b)


@pytest.mark.parametrize("nchan,ntaps,n_chunks,frames,extra", [
    (512, 4, 7, 20, 5), (1024, 4, 3, 9, 0), (2048, 4, 5, 33, 100), (2048, 32, 2, 40, 0), (4096, 8, 2, 11, 7),
    (8192, 4, 3, 6, 1), (4096, 3, 2, 11, 7), (1024, 1, 4, 5, 0), (512, 7, 300, 3, 0), (2048, 4, 1, 128, 0),
    (512, 4, 1, 2049, 0),    # 256 frame ranges of 9 over 2049 frames: the last ranges are empty
    (512, 7, 1, 700, 3), (512, 20, 1, 600, 0), (1024, 20, 2, 150, 5), (2048, 9, 1, 260, 0),   # pre-filter pass, its frame splits
    (8192, 4, 1, 70, 5), (8192, 9, 2, 40, 0), (8192, 16, 1, 130, 77), (8192, 17, 2, 9, 0)])    # 8192 as two 4096-channel problems (17 taps: plain)
def test_tiled_path_matches_oracle(plan_mod, torch, nchan, ntaps, n_chunks, frames, extra):
    """The other --nfft values (effex.py:778) on the tiled fused kernel: rows, ragged tails, frame splits
    (few chunks, many frames), integration in uneven calls, continuum."""
    num_samp = nchan * frames + extra
    x = synth.synth_iq(321, n_chunks, 2, num_samp)
    window = design_window(ntaps, nchan)
    rot = plan_mod.rot_table(nchan, gi.BANDWIDTH, gi.FREQUENCY, -2e-7)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window) as p:
        assert p.path == "tiled"
        p.set_rot(rot)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        for c in range(min(n_chunks, 6)):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, -2e-7,
                                      "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        p.fx_accumulate(xd[: n_chunks // 3 + 1])
        p.fx_accumulate(xd[n_chunks // 3 + 1:])
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ[0], rows[:, 0].astype(np.complex128).mean(axis=0)) < 2e-6
        cont_rows = p.fx_rows(xd, "CONTINUUM", gi.BANDWIDTH).cpu().numpy()
        np.testing.assert_allclose(cont_rows[:, 0], rows[:, 0].astype(np.complex128).mean(axis=1) / gi.BANDWIDTH,
                                   rtol=2e-5, atol=1e-7 * np.abs(cont_rows).max())
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window, path="generic") as g:
        g.set_rot(rot)
        assert rel_err(rows, g.fx_rows(xd, "SPECTRUM").cpu().numpy()) < 4e-6


@pytest.mark.parametrize("ntaps,n_chunks,frames,extra", [(4, 300, 2, 0), (4, 2, 1100, 3), (1, 5, 9, 8191), (3, 1, 1, 0), (2, 17, 33, 100)])
def test_8192_channels_in_two_passes(plan_mod, torch, monkeypatch, ntaps, n_chunks, frames, extra):
    """--nfft 8192 with two antennas and up to four taps: f8192_ring_kernel writes antenna 0's spectra, its XM form runs antenna 1
    through the same stages and multiplies by them as it goes (2 x the algorithmic bytes; the split into two 4096-channel problems
    moves 3 x) -- many chunks of few frames, runs longer than the float32 row limit, a one-frame chunk, a ragged tail of almost a
    frame; against the oracle (effex.py:490-527), the float64 mean of the rows, and the split route (FXC_X8192=0)."""
    nchan = 8192
    num_samp = nchan * frames + extra
    x = synth.synth_iq(8192 + n_chunks, n_chunks, 2, num_samp)
    window = design_window(ntaps, nchan)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window) as p:
        assert p.path == "tiled" and p.info["block"] == 512, p.info
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, -2e-7)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        for c in sorted({0, n_chunks // 2, n_chunks - 1}):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, -2e-7, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        np.testing.assert_array_equal(p.fx_rows(xd, "SPECTRUM").cpu().numpy(), rows)
        p.fx_accumulate(xd[: n_chunks // 3 + 1])
        if n_chunks // 3 + 1 < n_chunks:
            p.fx_accumulate(xd[n_chunks // 3 + 1:])
        assert rel_err(p.finalize("SPECTRUM"), rows.astype(np.complex128).mean(axis=0)) < 2e-6
        # the receivers' bytes (uint8 I, Q; effex.py:391-395 converts and de-means on the host): converted on their way into both
        # passes' rings -- against the conversion pass + the same route, and the oracle on the oracle's conversion
        nb = min(n_chunks, 5)
        u8 = torch.from_numpy(np.random.default_rng(frames).integers(0, 256, size=(nb, 2, num_samp, 2), dtype=np.uint8)).cuda()
        by = p.fx_rows_u8(u8, "SPECTRUM", remove_dc=True).cpu().numpy()
        assert rel_err(by, p.fx_rows(p.convert_u8(u8, remove_dc=True)).cpu().numpy()) < TOL_VIS
        a = fx_oracle.u8_to_complex(u8[:1].cpu().numpy())[0]
        ref = fx_oracle.pfb_xcorr(fx_oracle.remove_dc(a[0]), fx_oracle.remove_dc(a[1]), ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY,
                                  -2e-7, "SPECTRUM")
        assert rel_err(by[0, 0], ref) < TOL_VIS
        p.fx_accumulate_u8(u8, remove_dc=True)
        assert rel_err(p.finalize("SPECTRUM"), by.astype(np.complex128).mean(axis=0)) < 2e-6
    monkeypatch.setenv("FXC_X8192", "0")      # (a route knob of the developer library: the shipped one reads none)
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window, dev=True) as q:
        assert q.info["block"] == 1024
        q.set_delay(gi.BANDWIDTH, gi.FREQUENCY, -2e-7)
        assert rel_err(q.fx_rows(xd, "SPECTRUM").cpu().numpy(), rows) < 2e-6


@pytest.mark.parametrize("nchan,ntaps,n_chunks,frames,extra", [
    (256, 4, 3, 40, 7), (256, 4, 1, 1024, 0), (128, 4, 5, 33, 100), (64, 4, 2, 500, 3), (32, 4, 9, 70, 1), (16, 4, 4, 300, 0),
    (256, 3, 2, 9, 0), (128, 1, 7, 5, 2), (64, 2, 300, 3, 0), (16, 4, 1, 16384, 5), (32, 4, 1, 1, 0), (256, 4, 700, 2, 0),
    (16, 4, 1, 3, 0),
    # more than four taps: the pre-filter pass (streams side by side in a workgroup below 256 channels) + one unit tap
    (256, 8, 3, 40, 7), (64, 16, 5, 300, 3), (16, 32, 4, 100, 1), (128, 5, 2, 33, 0), (32, 9, 70, 20, 2), (256, 32, 1, 2000, 5),
    (64, 8, 1, 3, 0)])
def test_small_channel_counts_match_oracle(plan_mod, torch, nchan, ntaps, n_chunks, frames, extra):
    """--nfft 16 ... 256 (a free integer in the reference, effex.py:733-739) on the wave-local kernel (k_small.h): several
    work items per wave, items beyond the last one, ranges of one frame and empty ranges, ragged tails, integration in
    uneven calls, continuum, bytes in, up to 32 taps -- against the oracle and against the generic kernels."""
    num_samp = nchan * frames + extra
    x = synth.synth_iq(4321, n_chunks, 2, num_samp)
    window = design_window(ntaps, nchan)
    rot = plan_mod.rot_table(nchan, gi.BANDWIDTH, gi.FREQUENCY, -2e-7)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window) as p:
        assert p.path == "tiled"
        p.set_rot(rot)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        for c in range(min(n_chunks, 4)):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, -2e-7,
                                      "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        p.fx_accumulate(xd[: n_chunks // 3 + 1])
        p.fx_accumulate(xd[n_chunks // 3 + 1:])
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ[0], rows[:, 0].astype(np.complex128).mean(axis=0)) < 2e-6
        cont_rows = p.fx_rows(xd, "CONTINUUM", gi.BANDWIDTH).cpu().numpy()
        np.testing.assert_allclose(cont_rows[:, 0], rows[:, 0].astype(np.complex128).mean(axis=1) / gi.BANDWIDTH,
                                   rtol=2e-5, atol=1e-7 * np.abs(cont_rows).max())
        u8 = torch.from_numpy(np.random.default_rng(nchan + frames).integers(
            0, 256, size=(min(n_chunks, 5), 2, num_samp, 2), dtype=np.uint8)).cuda()
        rows_u8 = p.fx_rows_u8(u8).cpu().numpy()
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window, path="generic") as g:
        g.set_rot(rot)
        assert rel_err(rows, g.fx_rows(xd, "SPECTRUM").cpu().numpy()) < 4e-6
        assert rel_err(rows_u8, g.fx_rows_u8(u8).cpu().numpy()) < TOL_VIS


@pytest.mark.parametrize("n_ant,nchan,ntaps,n_chunks,frames,extra", [
    (2, 1000, 4, 3, 40, 7), (2, 3000, 4, 2, 21, 0), (2, 96, 4, 5, 300, 5), (2, 1536, 8, 2, 33, 100), (2, 100, 3, 7, 9, 0),
    (2, 997, 4, 2, 10, 3), (2, 6, 4, 4, 1000, 1), (2, 12, 1, 3, 50, 0), (3, 48, 5, 2, 77, 2), (2, 2310, 2, 1, 12, 0),
    (2, 5000, 4, 1, 9, 11), (4, 10240, 4, 1, 5, 0), (2, 3, 4, 2, 4000, 2), (2, 7, 32, 2, 500, 0), (2, 1001, 4, 1, 3, 0),
    (2, 6561, 4, 1, 4, 0), (2, 250, 4, 1, 1, 0), (5, 360, 4, 2, 30, 1),
    # a prime factor beyond 45 nfft / nchan: chirp-z rows (Bluestein), table in LDS up to 4096 points, from global at 8192
    (2, 1002, 4, 2, 12, 5), (3, 4093, 4, 1, 5, 0), (2, 2049, 2, 2, 7, 1), (2, 97, 5, 3, 200, 3), (2, 127, 4, 1, 1, 0), (2, 67, 4, 2, 50, 0),
    (2, 4096 + 1, 4, 1, 3, 0), (2, 8190 // 2 + 4, 4, 1, 2, 0), (2, 251, 4, 3, 77, 2), (3, 509, 3, 2, 31, 0),
    (2, 5003, 4, 1, 3, 1), (2, 5119, 2, 1, 2, 0),       # 10 080 / 10 240 points: the largest rows that fit
    # 3 .. 64 antennas: spectra antenna-interleaved, then the X-engines of the tiled paths (registers up to 8, blocks of 8 beyond)
    (8, 1000, 4, 3, 30, 3), (11, 96, 4, 2, 200, 1), (16, 250, 3, 2, 40, 0), (7, 2310, 2, 1, 5, 0), (9, 12, 4, 5, 1000, 2),
    # beyond 10240 channels one row is all the LDS holds: the stages alternate between it and the output row
    (2, 12000, 4, 2, 3, 7), (3, 15000, 2, 1, 2, 0), (2, 10241, 4, 1, 2, 0), (2, 16380, 4, 1, 1, 0)])
def test_any_channel_count_matches_oracle(plan_mod, torch, monkeypatch, n_ant, nchan, ntaps, n_chunks, frames, extra):
    """`--resolution` is a free integer (effex.py:733-739): channel counts that are not a power of two run the FIR + mixed-radix
    Stockham kernel (fx_mixed.h: radices 4, 2, 3, 5, 7, 11, 13 in registers, other prime factors from the LDS row up to about 90,
    larger ones as a chirp-z convolution; with two antennas the same kernel multiplies and integrates) -- against the oracle,
    and against the direct O(N^2) DFT kernel it replaced."""
    num_samp = nchan * frames + extra
    x = synth.synth_iq(777 + nchan, n_chunks, n_ant, num_samp, delays=np.arange(n_ant) % 7)
    window = design_window(ntaps, nchan)
    rot = plan_mod.rot_table(nchan, gi.BANDWIDTH, gi.FREQUENCY, -2e-7)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as p:
        assert p.path == "generic"
        p.set_rot(rot)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        for c in range(min(n_chunks, 3)):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, -2e-7, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        p.fx_accumulate(xd[: n_chunks // 3 + 1])
        p.fx_accumulate(xd[n_chunks // 3 + 1:])
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ, rows.astype(np.complex128).mean(axis=0)) < 2e-6
        if n_ant > 2:                           # every baseline, in the order (0,1),(0,2),...: the oracle's integration of chunk 0
            p.set_rot(plan_mod.rot_table(nchan, gi.BANDWIDTH, gi.FREQUENCY, 0.0))
            all_b = p.fx_rows(xd[:1]).cpu().numpy()[0]
            assert rel_err(all_b, fx_oracle.fx_integrate(x[:1], nchan, window)) < TOL_VIS
            p.set_rot(rot)
        spec = p.channelize(xd[0, 0]).cpu().numpy()
        assert rel_err(spec.reshape(-1, nchan), fx_oracle.spectrometer_poly(x[0, 0], ntaps, nchan, window)) < TOL_VIS
        if n_ant == 2 and nchan <= 5120:       # the steps either side of the path: receiver bytes in, DC removal on the device
            u8 = torch.from_numpy(np.random.default_rng(nchan).integers(0, 256, size=(2, 2, num_samp, 2), dtype=np.uint8)).cuda()
            by = p.fx_rows_u8(u8, "SPECTRUM", remove_dc=True).cpu().numpy()      # (bytes converted inside the F + X kernel)
            assert rel_err(by, p.fx_rows(p.convert_u8(u8, remove_dc=True)).cpu().numpy()) < TOL_VIS
            a = fx_oracle.u8_to_complex(u8[:1].cpu().numpy())[0]
            p.set_rot(plan_mod.rot_table(nchan, gi.BANDWIDTH, gi.FREQUENCY, 0.0))
            by0 = p.fx_rows_u8(u8[:1], "SPECTRUM", remove_dc=True).cpu().numpy()
            p.set_rot(rot)
            assert rel_err(by0[0, 0], fx_oracle.pfb_xcorr(fx_oracle.remove_dc(a[0]), fx_oracle.remove_dc(a[1]), ntaps, nchan, window,
                                                          gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")) < TOL_VIS
            p.fx_accumulate_u8(u8, remove_dc=True)
            assert rel_err(p.finalize("SPECTRUM"), by.astype(np.complex128).mean(axis=0)) < 2e-6
            dc = p.fx_rows(xd[:1] + (0.25 - 0.5j), remove_dc=True).cpu().numpy()
            ref = fx_oracle.pfb_xcorr(fx_oracle.remove_dc(x[0, 0] + (0.25 - 0.5j)), fx_oracle.remove_dc(x[0, 1] + (0.25 - 0.5j)),
                                      ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, -2e-7, "SPECTRUM")
            assert rel_err(dc[0, 0], ref) < TOL_VIS
        specialised = p.info["specialised"]
    if specialised & 1:      # the kernel compiled for this channel count ran above: the any-shape kernel must agree with it and the oracle
        monkeypatch.setenv("FXC_RTC", "0")
        with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as a:
            assert a.info["specialised"] == 0
            a.set_rot(rot)
            rows_any = a.fx_rows(xd, "SPECTRUM").cpu().numpy()
            assert rel_err(rows_any, rows) < 2e-6
            assert rel_err(rows_any[0, 0], fx_oracle.pfb_xcorr(x[0, 0], x[0, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, -2e-7,
                                                               "SPECTRUM")) < TOL_VIS
        monkeypatch.delenv("FXC_RTC")
    # a plan is specialised exactly when the library can build the kernel for its shape (fxc_spec_probe: the same search,
    # no plan) -- two antennas only; shapes without one (a large prime factor, more than four taps, rows too long for the
    # registers of a workgroup) keep the any-shape kernel
    from effex_amd import _lib
    can = n_ant == 2 and _lib.load().fxc_spec_probe(nchan, ntaps, 0, None, None, 0) == 0
    assert bool(specialised & 1) == can, (specialised, can)      # (bit 1: the F stage alone, also for 3 and more antennas)
    if (n_ant, nchan, ntaps) in {(2, 1000, 4), (2, 96, 4), (2, 100, 3), (2, 6, 4), (2, 12, 1), (2, 3, 4), (2, 250, 4)}:
        assert specialised & 1
    if (n_ant, nchan, ntaps) in {(2, 997, 4), (2, 1536, 8), (3, 48, 5), (2, 12000, 4), (2, 7, 32)}:
        assert not specialised
    if (n_ant, nchan, ntaps) == (2, 6561, 4):      # 4097 ... 8192 channels: the F stage alone (one stream per workgroup), X from spectra in HBM
        assert specialised == 2 + 4      # (+ 4: antenna 1 through the second-pass build, its last butterfly multiplying with antenna 0's spectra)
    # the direct DFT: a kernel of the developer build only (libfxcorr_dev.so), chosen by a knob read when the plan is built
    monkeypatch.setenv("FXC_GENERIC_FFT", "radix2")
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window, dev=True) as d:
        d.set_rot(rot)
        # (the direct DFT's own float32 sums of nchan terms are the larger share of the difference at the top sizes)
        assert rel_err(rows, d.fx_rows(xd, "SPECTRUM").cpu().numpy()) < (4e-6 if nchan <= 8192 else 1e-5)


@pytest.mark.parametrize("n_ant,ntaps,n_streams,frames,extra", [(1, 4, 3, 7, 5), (1, 4, 600, 9, 0), (1, 3, 1, 40, 8191), (1, 1, 2, 1, 0),
                                                           (3, 4, 2, 11, 3), (1, 4, 1, 300, 1)])
def test_channelize_at_8192_channels(plan_mod, torch, monkeypatch, n_ant, ntaps, n_streams, frames, extra):
    """fxc_channelize (the drop-in's _spectrometer_poly, effex.py:530-555) at 8192 branches and up to four taps: one stream per
    workgroup with the four frames of the FIR in a register ring (k_tiled.h::f8192_ring_kernel; the pair kernel it replaces
    re-read 2.5 frames per frame) -- spectra against the oracle for odd stream counts, runs of one frame, runs cut into several
    splits, and the antenna-interleaved layout of the 3-antenna route; the pair kernel (FXC_F8192=0) agrees."""
    nchan = 8192
    num_samp = nchan * frames + extra
    x = synth.synth_iq(8192 + n_streams, n_streams, n_ant, num_samp, delays=np.arange(n_ant) % 5)
    window = design_window(ntaps, nchan)
    xd = torch.from_numpy(x).cuda()
    flat = xd.reshape(-1, num_samp)
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as p:
        spec = p.channelize(flat).cpu().numpy()
        for s_ in sorted({0, flat.shape[0] // 2, flat.shape[0] - 1}):
            ref = fx_oracle.spectrometer_poly(x.reshape(-1, num_samp)[s_], ntaps, nchan, window)
            assert rel_err(spec[s_], ref) < TOL_SPEC, s_
        if n_ant > 1:
            rows = p.fx_rows(xd).cpu().numpy()
            assert rel_err(rows[0], fx_oracle.fx_integrate(x[:1], nchan, window)) < TOL_VIS
    monkeypatch.setenv("FXC_F8192", "0")      # (a route knob of the developer library: the shipped one reads none)
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window, dev=True) as a:
        assert rel_err(a.channelize(flat).cpu().numpy(), spec) < 2e-6
        if n_ant > 1:
            assert rel_err(a.fx_rows(xd).cpu().numpy(), rows) < 2e-6


@pytest.mark.parametrize("nchan,mode", [(4096, "SPECTRUM"), (4096, "CONTINUUM"), (1000, "SPECTRUM"), (256, "SPECTRUM")])
def test_rows_straight_into_pinned_host_memory(plan_mod, torch, nchan, mode):
    """FXC_MEM_DEVICE_TO_PINNED: samples resident in device memory, rows delivered into fxc_host_alloc memory by the finishing
    kernel itself (what the time-series sink double-buffers, effex.py:402-410, 687-696 at the device's pace): the call returns
    with the work queued, after plan.sync() the pinned array holds exactly the rows of the ordinary device call -- for complex64
    and byte samples and with the DC removal; an `out` that is not pinned memory is refused before anything is queued."""
    num_samp, n_chunks = nchan * 21 + 3, 37
    x = torch.from_numpy(synth.synth_iq(606, n_chunks, 2, num_samp)).cuda()
    u8 = torch.randint(0, 256, (n_chunks, 2, num_samp, 2), dtype=torch.uint8, device="cuda",
                       generator=torch.Generator(device="cuda").manual_seed(884 + nchan))      # (seeded: the bound below is a measured one)
    with plan_mod.FxPlan(2, nchan, 4, num_samp) as p:
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 1e-7)
        ref = p.fx_rows(x, mode, gi.BANDWIDTH).cpu().numpy()
        out = plan_mod.pinned_empty(ref.shape, ref.dtype)
        out[...] = 0
        assert p.fx_rows(x, mode, gi.BANDWIDTH, out=out) is out
        p.sync()
        np.testing.assert_array_equal(out, ref)
        ref_dc = p.fx_rows(x, mode, gi.BANDWIDTH, remove_dc=True).cpu().numpy()
        p.fx_rows(x, mode, gi.BANDWIDTH, remove_dc=True, out=out)
        p.sync()
        np.testing.assert_array_equal(out, ref_dc)
        ref_u8 = p.fx_rows_u8(u8, mode, gi.BANDWIDTH).cpu().numpy()
        p.fx_rows_u8(u8, mode, gi.BANDWIDTH, out=out)
        p.sync()
        np.testing.assert_array_equal(out, ref_u8)
        with pytest.raises(ValueError):
            p.fx_rows(x, mode, gi.BANDWIDTH, out=np.empty(ref.shape, ref.dtype))       # pageable memory: not a device-writable target
        p.sync()


@pytest.mark.parametrize("rtc", ["1", "0"])
def test_long_chunks_of_few_channels_keep_float32_runs_short(plan_mod, torch, monkeypatch, rtc):
    """Few channels with long chunks (12 channels, 2^19 samples: 43 690 spectra per chunk): every kernel that sums s0 conj(s1) in
    float32 registers over a run of frames cuts the chunk into runs of at most 1 024 spectra (h_launch.h::kRowSpectra; the
    any-shape mixed-radix kernel had no such cap: ADVICE r04) -- rows against the oracle, and the integration against the float64
    mean of the rows, with the kernel compiled for the channel count and with the any-shape one."""
    monkeypatch.setenv("FXC_RTC", rtc)
    nchan, ntaps, num_samp, n_chunks = 12, 4, 2 ** 19 + 7, 3
    x = synth.synth_iq(2024, n_chunks, 2, num_samp)
    window = np.random.default_rng(12).standard_normal(ntaps * nchan)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window) as p:
        assert p.info["specialised"] == int(rtc)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        ref = fx_oracle.pfb_xcorr(x[1, 0], x[1, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
        assert rel_err(rows[1, 0], ref) < TOL_VIS
        p.fx_accumulate(xd)
        assert rel_err(p.finalize("SPECTRUM"), rows.astype(np.complex128).mean(axis=0)) < 2e-6


@pytest.mark.parametrize("n_ant,nchan,ntaps,n_streams,frames,extra", [
    (1, 1000, 4, 5, 60, 3), (1, 96, 4, 333, 40, 0), (1, 720, 3, 2, 17, 1), (1, 250, 1, 1, 1, 0), (1, 2000, 4, 3, 9, 0),
    (3, 1000, 4, 4, 30, 5), (5, 96, 2, 3, 100, 0), (1, 3000, 4, 5, 12, 1), (3, 2400, 4, 2, 11, 0), (1, 4000, 4, 3, 8, 0),
    (1, 5000, 4, 5, 9, 2), (2, 6000, 3, 3, 7, 0), (3, 8000, 4, 2, 6, 1), (1, 7168, 4, 300, 5, 0)])
def test_specialised_f_stage_on_the_device(plan_mod, torch, monkeypatch, n_ant, nchan, ntaps, n_streams, frames, extra):
    """fxc_channelize (the drop-in's _spectrometer_poly, effex.py:530-555) at channel counts that are not a power of two runs
    fx_spec.h built as the F stage alone -- a pair of streams per workgroup, odd stream counts included; 3 and more antennas
    take it for their F pass (spectra antenna-interleaved for the X-engines).  Against the oracle, and the any-shape kernel
    (FXC_RTC=0) must agree."""
    num_samp = nchan * frames + extra
    x = synth.synth_iq(99 + nchan, n_streams, n_ant, num_samp, delays=np.arange(n_ant) % 5)
    window = design_window(ntaps, nchan)
    xd = torch.from_numpy(x).cuda()
    flat = xd.reshape(-1, num_samp)
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as p:
        spec = p.channelize(flat).cpu().numpy()
        assert p.info["specialised"] & 2          # the F stage ran the build for this channel count
        for s_ in sorted({0, flat.shape[0] // 2, flat.shape[0] - 1}):
            ref = fx_oracle.spectrometer_poly(x.reshape(-1, num_samp)[s_], ntaps, nchan, window)
            assert rel_err(spec[s_], ref) < TOL_SPEC_ANY, s_
        rows = p.fx_rows(xd).cpu().numpy() if n_ant > 1 else None
        if n_ant > 1:
            assert rel_err(rows[0], fx_oracle.fx_integrate(x[:1], nchan, window)) < TOL_VIS
    monkeypatch.setenv("FXC_RTC", "0")
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as a:
        assert rel_err(a.channelize(flat).cpu().numpy(), spec) < 2e-6 and a.info["specialised"] == 0
        if n_ant > 1:
            assert rel_err(a.fx_rows(xd).cpu().numpy(), rows) < 2e-6


@pytest.mark.parametrize("nchan,ntaps,n_chunks,frames,extra", [
    (1000, 4, 700, 40, 3), (1000, 4, 3, 262, 144), (96, 4, 2100, 25, 0), (1536, 4, 5, 170, 1), (720, 3, 64, 33, 2), (250, 2, 9, 1000, 0),
    (12, 4, 3, 20000, 5), (2000, 4, 1, 131, 0), (1001, 4, 2, 11, 0), (600, 1, 40, 50, 7), (20, 4, 1, 1, 0), (7, 1, 300, 90, 0),
    # above 2048 channels: the lean build (taps and first twiddles from tables in L2)
    (3000, 4, 300, 21, 7), (4000, 4, 2, 65, 0), (2560, 3, 5, 40, 1), (2400, 4, 700, 9, 0), (3072, 2, 3, 1, 0),
    # prime factors 17 ... 23: the lean build on at most 256 threads
    (1020, 4, 40, 30, 3), (34, 4, 5, 700, 1), (1140, 3, 2, 9, 0), (460, 4, 300, 11, 0), (1900, 4, 3, 137, 5), (2040, 4, 200, 7, 0)])
def test_specialised_kernel_on_the_device(plan_mod, torch, nchan, ntaps, n_chunks, frames, extra):
    """The F+X kernel compiled for one channel count when the plan is made (fx_spec.h through hiprtc, h_rtc.h) -- every shape
    class the host emulation covers (tests/test_emul.py), here on the device and at launch sizes that take several rounds of
    workgroups, several splits of a chunk's frames, runs of one frame and runs longer than the float32 row limit: rows against
    the oracle (effex.py:490-527), the integration against the float64 mean of the rows, the byte ingest against the conversion
    pass, and bit-identical results from two calls (no atomics, no order left to chance)."""
    num_samp = nchan * frames + extra
    x = synth.synth_iq(4242 + nchan, n_chunks, 2, num_samp)
    window = design_window(ntaps, nchan)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window) as p:
        assert p.path == "generic" and p.info["specialised"] == 1 and p.info["spec_vgprs"] > 0, p.info
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 3e-7)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        for c in sorted({0, n_chunks // 2, n_chunks - 1}):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 3e-7, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        np.testing.assert_array_equal(p.fx_rows(xd, "SPECTRUM").cpu().numpy(), rows)
        p.fx_accumulate(xd)
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ, rows.astype(np.complex128).mean(axis=0)) < 2e-6
        cont = p.fx_rows(xd[:2], "CONTINUUM", gi.BANDWIDTH).cpu().numpy()
        ref = fx_oracle.pfb_xcorr(x[0, 0], x[0, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 3e-7, "CONTINUUM")
        assert abs(cont[0, 0] - ref) < TOL_CONT * abs(ref) + 1e-9 * np.abs(rows[0]).max() / gi.BANDWIDTH
        u8 = torch.from_numpy(np.random.default_rng(nchan).integers(0, 256, size=(min(n_chunks, 4), 2, num_samp, 2), dtype=np.uint8)).cuda()
        by = p.fx_rows_u8(u8, "SPECTRUM", remove_dc=True).cpu().numpy()
        assert rel_err(by, p.fx_rows(p.convert_u8(u8, remove_dc=True)).cpu().numpy()) < TOL_VIS


@pytest.mark.parametrize("nchan", [8, 64, 1024, 4096, 8192])
def test_mixed_radix_kernel_on_powers_of_two(plan_mod, torch, monkeypatch, nchan):
    """The same kernel with only fours and a two (FXC_GENERIC_FFT=mixed moves the generic path's powers of two onto it)."""
    num_samp = nchan * 23 + 5
    x = synth.synth_iq(55, 2, 2, num_samp)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, nchan, 3, num_samp, path="generic") as g:
        ref = g.fx_rows(xd).cpu().numpy()
    monkeypatch.setenv("FXC_GENERIC_FFT", "mixed")      # (a route knob of the developer library: the shipped one reads none)
    with plan_mod.FxPlan(2, nchan, 3, num_samp, path="generic", dev=True) as m:
        assert rel_err(m.fx_rows(xd).cpu().numpy(), ref) < 4e-6


@pytest.mark.parametrize("n_ant,nchan,ntaps,n_chunks,frames", [(2, 4, 4, 5, 3000), (2, 8, 4, 3, 777), (3, 8, 7, 2, 100), (2, 16384, 4, 2, 3),
                                                               (3, 16384, 2, 1, 2), (2, 2, 4, 3, 500)])
def test_powers_of_two_no_tuned_kernel_takes(plan_mod, torch, n_ant, nchan, ntaps, n_chunks, frames):
    """--nfft 4, 8 and 16384 on the automatic path ride the mixed-radix kernel (two antennas: F and X in one pass; 16384: one LDS
    row, the other in the output); a forced generic path keeps the radix-2 kernels, the reference here besides the oracle."""
    num_samp = nchan * frames + 1
    x = synth.synth_iq(91 + nchan, n_chunks, n_ant, num_samp)
    xd = torch.from_numpy(x).cuda()
    window = design_window(ntaps, nchan)
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as p, \
            plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window, path="generic") as g:
        assert p.path == "generic"
        rows = p.fx_rows(xd).cpu().numpy()
        assert rel_err(rows, g.fx_rows(xd).cpu().numpy()) < 4e-6
        ref = fx_oracle.pfb_xcorr(x[0, 0], x[0, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
        assert rel_err(rows[0, 0], ref) < TOL_VIS
        p.fx_accumulate(xd)
        assert rel_err(p.finalize("SPECTRUM"), rows.astype(np.complex128).mean(axis=0)) < 2e-6
        assert rel_err(p.channelize(xd[0]).cpu().numpy(), g.channelize(xd[0]).cpu().numpy()) < 4e-6


def test_small_channel_counts_outside_the_kernel_fall_back(plan_mod, torch):
    """Fewer than 16 channels, or channel counts that are not 16 x a power of two: the generic kernels, as before (more
    than four taps no longer: the pre-filter pass serves them, test_small_channel_counts_match_oracle)."""
    with plan_mod.FxPlan(2, 8, 4, 8 * 20) as p, plan_mod.FxPlan(3, 48, 5, 48 * 20) as q, \
            plan_mod.FxPlan(9, 8, 4, 8 * 20) as r, plan_mod.FxPlan(2, 256, 8, 256 * 20) as t:
        assert p.path == "generic" and q.path == "generic" and r.path == "generic" and t.path == "tiled"
    with pytest.raises(NotImplementedError):
        plan_mod.FxPlan(2, 48, 8, 48 * 20, path="tiled")


@pytest.mark.parametrize("n_ant,nchan,ntaps,n_chunks,frames,extra", [
    (3, 256, 4, 3, 40, 7), (8, 64, 4, 5, 200, 3), (4, 128, 4, 1, 700, 0), (5, 256, 2, 2, 9, 100), (8, 256, 4, 40, 16, 0),
    (7, 64, 4, 2, 1, 0), (3, 64, 4, 2, 2500, 7), (6, 128, 4, 3, 1025, 0),    # the last two: more than 1024 frames per chunk
    (3, 32, 4, 2, 300, 5), (8, 16, 4, 3, 90, 0), (11, 32, 4, 2, 40, 1),      # part of a wave in the X-engine; more than 8 antennas
    (5, 64, 8, 3, 50, 3), (12, 128, 16, 2, 40, 0), (3, 16, 6, 2, 200, 1)])   # more than four taps: pre-filter pass first
def test_small_channel_counts_multi_antenna(plan_mod, torch, n_ant, nchan, ntaps, n_chunks, frames, extra):
    """3 and more antennas at 16 ... 256 channels: the F-only variant of the wave-local kernel (odd stream counts leave the
    last pair half empty) + the X-engine, against the oracle and the generic kernels."""
    num_samp = nchan * frames + extra
    x = synth.synth_iq(99 + n_ant, n_chunks, n_ant, num_samp, delays=np.arange(n_ant) % 7)
    window = design_window(ntaps, nchan)
    xd = torch.from_numpy(x).cuda()
    pairs = [(a, b) for a in range(n_ant) for b in range(a + 1, n_ant)]
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as p, \
            plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window, path="generic") as g:
        assert p.path == "tiled" and g.path == "generic"
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 0.0)
        g.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 0.0)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        assert rel_err(rows, g.fx_rows(xd, "SPECTRUM").cpu().numpy()) < 4e-6
        for idx in (0, len(pairs) // 2, len(pairs) - 1):
            a, b = pairs[idx]
            ref = fx_oracle.pfb_xcorr(x[0, a], x[0, b], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
            assert rel_err(rows[0, idx], ref) < TOL_VIS, (a, b)
        p.fx_accumulate(xd)
        assert rel_err(p.finalize("SPECTRUM"), rows.astype(np.complex128).mean(axis=0)) < 2e-6


@pytest.mark.parametrize("nchan,ntaps,n_streams,frames,extra", [(256, 4, 5, 33, 9), (128, 4, 2, 900, 0), (64, 3, 1, 50, 3),
                                                                (16, 4, 7, 300, 1), (32, 1, 4, 4, 0),
                                                                (256, 32, 3, 70, 5), (64, 8, 5, 300, 0), (16, 12, 2, 90, 3)])
def test_small_channel_counts_channelize(plan_mod, torch, nchan, ntaps, n_streams, frames, extra):
    """fxc_channelize (the _spectrometer_poly drop-in, effex.py:530-555) at 16 ... 256 branches: natural-order spectra
    from the F-only variant of the wave-local kernel, against the oracle."""
    num_samp = nchan * frames + extra
    x = synth.synth_iq(7, n_streams, 1, num_samp)[:, 0]
    window = design_window(ntaps, nchan)
    with plan_mod.FxPlan(1, nchan, ntaps, num_samp, window=window) as f:
        spec = f.channelize(torch.from_numpy(x).cuda()).cpu().numpy()
    assert spec.shape == (n_streams, num_samp // nchan, nchan)
    for s_ in range(n_streams):
        assert rel_err(spec[s_], fx_oracle.spectrometer_poly(x[s_], ntaps, nchan, window)) < TOL_SPEC, s_


@pytest.mark.parametrize("n_chunks", [1, 2, 7])
def test_headline_shape_small_calls_split_frames(plan_mod, torch, n_chunks):
    """The reference hands over one chunk pair per call (effex.py:497-527): with fewer chunks than CUs the headline
    kernel splits the frames of a chunk over workgroups (frame ranges, leading-part rows); an automatically chosen plan
    and an explicit "fused" plan are the same thing now, and the tiled kernel for 4096 channels must agree with both."""
    num_samp = 262144
    x = synth.synth_iq(77, n_chunks, 2, num_samp)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as a, plan_mod.FxPlan(2, 4096, 4, num_samp, path="fused") as f, \
            plan_mod.FxPlan(2, 4096, 4, num_samp, path="tiled") as t:
        assert a.path == "fused" and f.path == "fused" and t.path == "tiled"
        ra = a.fx_rows(xd).cpu().numpy()
        rf = f.fx_rows(xd).cpu().numpy()
        np.testing.assert_array_equal(ra, rf)
        assert rel_err(t.fx_rows(xd).cpu().numpy(), rf) < 2e-6
        ref = fx_oracle.pfb_xcorr(x[0, 0], x[0, 1], 4, 4096, a.window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
        a.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 0.0)
        assert rel_err(a.fx_rows(xd).cpu().numpy()[0, 0], ref) < TOL_VIS
        a.fx_accumulate(xd)
        f.fx_accumulate(xd)
        assert rel_err(a.finalize("SPECTRUM"), f.finalize("SPECTRUM")) < 2e-6


def test_other_nfft_against_reference_goldens(plan_mod, torch, golden):
    """--nfft 1024 / 2048 / 8192 (tiled kernels) and --resolution 1000 / 96 / 997 / 1536 (not powers of two: the mixed-radix and
    chirp-z kernel) rows produced by the reference's own constructor + _run_task (tests/golden)."""
    meta, arrays = golden
    for case in meta["nfft"]:
        nbins, num_samp, chunks, delay = case["nbins"], case["num_samp"], case["chunks"], case["delay"]
        x = gi.nfft_input(nbins, num_samp, chunks)
        with plan_mod.FxPlan(2, nbins, 4, num_samp) as p:
            assert p.path == ("tiled" if nbins & (nbins - 1) == 0 else "generic")
            p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, delay)
            rows = p.fx_rows(x, "SPECTRUM")                  # host buffers in, host rows out
            assert rel_err(rows[:, 0], arrays[case["key"]]) < TOL_VIS, nbins


def test_randomized_shapes_tiled_vs_generic(plan_mod, torch):
    """150 random (nchan, ntaps, frames, ragged tail, chunk count) draws: the tiled kernels (ring / plain, frame-range
    splits, uint8 ingest where it applies) against the generic kernels, which share no code with them."""
    rng = np.random.default_rng(20261002)
    for case in range(150):
        nchan = int(rng.choice([16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]))
        ntaps = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8, 13, 32])) if nchan >= 512 else int(rng.choice([1, 2, 3, 4, 4]))
        frames = int(rng.integers(1, 70 if nchan <= 1024 else 24)) if nchan >= 512 else int(rng.integers(1, 2000))
        n_chunks = int(rng.choice([1, 1, 2, 3, 7, 19]))
        num_samp = nchan * frames + int(rng.integers(0, nchan))
        x = torch.from_numpy(synth.synth_iq(1000 + case, n_chunks, 2, num_samp)).cuda()
        tag = (case, nchan, ntaps, frames, n_chunks, num_samp)
        with plan_mod.FxPlan(2, nchan, ntaps, num_samp, path="tiled") as t, \
                plan_mod.FxPlan(2, nchan, ntaps, num_samp, path="generic") as g:
            rt, rg = t.fx_rows(x).cpu().numpy(), g.fx_rows(x).cpu().numpy()
            assert rel_err(rt, rg) < 4e-6, tag
            t.fx_accumulate(x)
            g.fx_accumulate(x)
            assert rel_err(t.finalize("SPECTRUM"), g.finalize("SPECTRUM")) < 4e-6, tag
            if case % 3 == 0:
                u8 = torch.from_numpy(rng.integers(0, 256, size=(n_chunks, 2, num_samp, 2), dtype=np.uint8)).cuda()
                assert rel_err(t.fx_rows_u8(u8).cpu().numpy(), g.fx_rows_u8(u8).cpu().numpy()) < TOL_VIS, tag
            if case % 5 == 0:                                        # 3 .. 8 antennas: F-only tiled kernel + X-engine
                n_ant = int(rng.integers(3, 9))
                xm = torch.from_numpy(synth.synth_iq(5000 + case, n_chunks, n_ant, num_samp)).cuda()
                with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp) as m, \
                        plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, path="generic") as mg:
                    assert m.path in ("tiled", "fused") and mg.path == "generic", tag
                    assert rel_err(m.fx_rows(xm).cpu().numpy(), mg.fx_rows(xm).cpu().numpy()) < 4e-6, tag + (n_ant,)
            if case % 4 == 0:
                xs = x.view(n_chunks * 2, num_samp)[: 2 * n_chunks - (case % 8 == 0)]      # sometimes an odd stream count
                with plan_mod.FxPlan(1, nchan, ntaps, num_samp) as f, \
                        plan_mod.FxPlan(1, nchan, ntaps, num_samp, path="generic") as fg:
                    assert rel_err(f.channelize(xs).cpu().numpy(), fg.channelize(xs).cpu().numpy()) < TOL_SPEC, tag


def test_fused_and_generic_agree(plan_mod, torch):
    x = torch.from_numpy(synth.synth_iq(5, 7, 2, 4096 * 10)).cuda()
    with plan_mod.FxPlan(2, 4096, 4, 4096 * 10, path="fused") as a, \
            plan_mod.FxPlan(2, 4096, 4, 4096 * 10, path="generic") as b:
        ra = a.fx_rows(x).cpu().numpy()
        rb = b.fx_rows(x).cpu().numpy()
    assert rel_err(ra, rb) < 2e-6


def test_eight_antennas_28_baselines(plan_mod, torch):
    n_ant, nchan, num_samp, n_chunks = 8, 1024, 1024 * 12, 3
    x = synth.synth_iq(21, n_chunks, n_ant, num_samp)
    window = design_window(4, nchan)
    with plan_mod.FxPlan(n_ant, nchan, 4, num_samp) as p:
        assert p.n_baselines == 28
        p.fx_accumulate(torch.from_numpy(x).cuda())
        integ = p.finalize("SPECTRUM")
        rows = p.fx_rows(x)
    ref = fx_oracle.fx_integrate(x, nchan, window)
    assert integ.shape == (28, nchan)
    assert rel_err(integ, ref) < TOL_VIS
    assert rel_err(rows.astype(np.complex128).mean(axis=0), ref) < TOL_VIS


@pytest.mark.parametrize("n_ant,nchan,ntaps", [(4, 4096, 4), (6, 4096, 4), (8, 4096, 4), (8, 1024, 4), (4, 2048, 4),
                                               (6, 512, 3), (4, 8192, 4), (8, 2048, 8), (6, 4096, 5), (3, 1024, 4), (5, 4096, 4),
                                               (7, 2048, 4)])
def test_multi_antenna_fused_path(plan_mod, torch, n_ant, nchan, ntaps):
    """BASELINE config 5 shape: 3 to 8 antennas run an F-only kernel (fused for even counts at nchan 4096 / ntaps 4, tiled
    otherwise) + the register-resident X-engine."""
    num_samp, n_chunks = nchan * 6 + 11, 3
    x = synth.synth_iq(23, n_chunks, n_ant, num_samp)
    window = design_window(ntaps, nchan)
    rot = plan_mod.rot_table(nchan, gi.BANDWIDTH, gi.FREQUENCY, 2e-7)
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp) as p, \
            plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, path="generic") as g:
        assert p.path == ("fused" if (nchan, ntaps) == (4096, 4) and n_ant % 2 == 0 else "tiled") and g.path == "generic"
        assert p.n_baselines == n_ant * (n_ant - 1) // 2
        p.set_rot(rot)
        g.set_rot(rot)
        xd = torch.from_numpy(x).cuda()
        p.fx_accumulate(xd)
        integ = p.finalize("SPECTRUM")
        rows = p.fx_rows(xd).cpu().numpy()
        rows_g = g.fx_rows(xd).cpu().numpy()
        cont = p.fx_rows(xd, "CONTINUUM", gi.BANDWIDTH).cpu().numpy()
    ref = fx_oracle.fx_integrate(x, nchan, window, rot=rot)
    assert rel_err(integ, ref) < TOL_VIS
    assert rel_err(rows.astype(np.complex128).mean(axis=0), ref) < TOL_VIS
    assert rel_err(rows, rows_g) < 2e-6
    np.testing.assert_allclose(cont, rows.astype(np.complex128).mean(axis=2) / gi.BANDWIDTH, rtol=2e-5,
                               atol=1e-7 * np.abs(cont).max())


@pytest.mark.parametrize("n_ant,nchan,ntaps,frames,n_chunks", [(9, 512, 4, 6, 3), (12, 1024, 4, 5, 2), (16, 4096, 4, 4, 3),
                                                             (17, 256, 4, 30, 2), (24, 2048, 8, 3, 1), (33, 64, 4, 50, 2),
                                                             (64, 512, 4, 3, 1), (10, 8192, 4, 2, 2), (12, 64, 4, 3000, 2)])
def test_more_than_eight_antennas(plan_mod, torch, n_ant, nchan, ntaps, frames, n_chunks):
    """9 ... 64 antennas: an F-only kernel + the X-engine over blocks of 8 antennas (partial last blocks, every pair of
    blocks), baselines in the order (0,1),(0,2)...: against the oracle's integration and the generic kernels' rows."""
    num_samp = nchan * frames + 5
    x = synth.synth_iq(31, n_chunks, n_ant, num_samp, delays=np.arange(n_ant) % 9)
    window = design_window(ntaps, nchan)
    rot = plan_mod.rot_table(nchan, gi.BANDWIDTH, gi.FREQUENCY, 1e-7)
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp) as p, \
            plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, path="generic") as g:
        assert p.path == "tiled" and g.path == "generic"
        assert p.n_baselines == n_ant * (n_ant - 1) // 2
        p.set_rot(rot)
        g.set_rot(rot)
        xd = torch.from_numpy(x).cuda()
        p.fx_accumulate(xd[:1])
        p.fx_accumulate(xd[1:])
        integ = p.finalize("SPECTRUM")
        rows = p.fx_rows(xd).cpu().numpy()
        rows_g = g.fx_rows(xd).cpu().numpy()
    assert rel_err(rows, rows_g) < 2e-6
    assert rel_err(integ, rows.astype(np.complex128).mean(axis=0)) < 2e-6
    ref = fx_oracle.fx_integrate(x, nchan, window, rot=rot)
    assert rel_err(integ, ref) < TOL_VIS


@pytest.mark.parametrize("ntaps,num_samp", [(4, 2 ** 16 + 3), (4, 2 ** 16), (3, 4098), (32, 5000), (1, 2048), (4, 100),
                                            (4, 2 ** 20), (4, 2 ** 20 + 2), (5, 2 ** 20)])   # BASELINE configs[2](i) at full size
def test_continuum_streaming_limit_nchan1(plan_mod, torch, ntaps, num_samp):
    """BASELINE config 3(i): nchan = 1, the PFB degenerates to a T-tap FIR and X to sum y0*conj(y1)."""
    x = synth.synth_iq(31, 3, 2, num_samp)
    h = np.linspace(0.4, 0.1, ntaps)
    with plan_mod.FxPlan(2, 1, ntaps, num_samp, window=h) as p, \
            plan_mod.FxPlan(2, 1, ntaps, num_samp, window=h, path="generic") as g:
        assert p.path == "stream" and g.path == "generic"
        cont = p.fx_rows(x, "CONTINUUM", gi.BANDWIDTH)
        spec = p.fx_rows(torch.from_numpy(x).cuda(), "SPECTRUM").cpu().numpy()
        cont_g = g.fx_rows(x, "CONTINUUM", gi.BANDWIDTH)
        p.fx_accumulate(x)
        integ = p.finalize("CONTINUUM", gi.BANDWIDTH)
    refs = [fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], ntaps, 1, h, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "CONTINUUM")
            for c in range(3)]
    for c in range(3):
        assert abs(cont[c, 0] - refs[c]) < TOL_CONT * abs(refs[c])
        assert abs(cont_g[c, 0] - refs[c]) < TOL_CONT * abs(refs[c])
        assert abs(spec[c, 0, 0] / gi.BANDWIDTH - refs[c]) < TOL_CONT * abs(refs[c])
    assert abs(integ[0] - np.mean(refs)) < TOL_CONT * abs(np.mean(refs))


def test_empty_batches_and_errors(plan_mod, torch):
    with plan_mod.FxPlan(2, 256, 4, 4096) as p:
        empty = torch.empty((0, 2, 4096), dtype=torch.complex64, device="cuda")
        assert p.fx_accumulate(empty) == 0
        assert p.fx_rows(empty).shape == (0, 1, 256)
        from effex_amd import _lib
        with pytest.raises(_lib.FxcError):
            p.finalize("SPECTRUM")                 # nothing accumulated
        with pytest.raises(ValueError):
            p.fx_accumulate(torch.empty((1, 2, 100), dtype=torch.complex64, device="cuda"))
        with pytest.raises(ValueError):
            p.fx_rows(torch.empty((1, 2, 4096), dtype=torch.complex64, device="cuda"), "CONTINUUM", 0.0)
        # byte ingest: empty batch, wrong shape, wrong dtype, bad bandwidth
        assert p.fx_accumulate_u8(torch.empty((0, 2, 4096, 2), dtype=torch.uint8, device="cuda")) == 0
        assert p.fx_rows_u8(np.empty((0, 2, 4096, 2), np.uint8)).shape == (0, 1, 256)
        with pytest.raises(ValueError):
            p.fx_rows_u8(torch.empty((1, 2, 4096), dtype=torch.uint8, device="cuda"))
        with pytest.raises(ValueError):
            p.fx_rows_u8(torch.empty((1, 2, 4096, 2), dtype=torch.int8, device="cuda"))
        with pytest.raises(ValueError):
            p.fx_rows_u8(torch.zeros((1, 2, 4096, 2), dtype=torch.uint8, device="cuda"), "CONTINUUM", 0.0)
    with plan_mod.FxPlan(1, 256, 4, 4096) as p1:
        with pytest.raises(ValueError):                # cross-correlation needs two streams
            p1.fx_rows_u8(torch.zeros((1, 1, 4096, 2), dtype=torch.uint8, device="cuda"))


# --------------------------------------------------------------------------------------------
# size-independent properties at the BASELINE size
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nchan,path,n_chunks", [(4096, "fused", 4), (4096, "fused", 300), (4096, None, 4),
                                                 (2048, None, 4), (1024, None, 300), (8192, None, 3)])
def test_linearity_and_conjugate_symmetry_full_size(plan_mod, torch, nchan, path, n_chunks):
    """BASELINE.json's num_samp through size-independent properties: the headline kernel (explicit "fused": one
    workgroup per chunk pair, also with more chunk pairs than CUs), the default plan, and the
    tiled kernels."""
    num_samp = 2 ** 18
    x = torch.from_numpy(synth.synth_iq(77777, n_chunks, 2, num_samp)).cuda()
    with plan_mod.FxPlan(2, nchan, 4, num_samp, path=path) as p:
        assert p.path == ("fused" if nchan == 4096 else "tiled")
        base = p.fx_rows(x).cpu().numpy().astype(np.complex128)
        scaled = p.fx_rows(x * 2.0).cpu().numpy().astype(np.complex128)          # exact: power of two
        np.testing.assert_array_equal(scaled, 4.0 * base)
        swapped = p.fx_rows(x.flip(1).contiguous()).cpu().numpy().astype(np.complex128)
        # V_10 = conj(V_01) bin for bin (rot = 1)
        assert rel_err(swapped, np.conj(base)) < 1e-6
        # auto-correlation is real and non-negative
        auto = p.fx_rows(torch.stack([x[:, 0], x[:, 0]], dim=1).contiguous()).cpu().numpy()
        assert np.abs(auto.imag).max() <= 1e-6 * np.abs(auto.real).max()
        assert auto.real.min() >= 0.0


@pytest.mark.parametrize("nchan,ntaps,path", [(4096, 4, "fused"), (2048, 4, "tiled"), (2048, 32, "tiled"), (8192, 4, "tiled")])
def test_largest_chunk_the_fast_paths_take(plan_mod, torch, nchan, ntaps, path):
    """num_samp = 2^27 (1 GiB per stream): the 32-bit byte offsets of the fast kernels' buffer loads and stores at their
    limit (fused kernel, tiled ring kernel, pre-filter pass, 8192 split), against the generic kernels (64-bit indexing,
    no shared code); one frame more per stream and the plan leaves the fast path."""
    from effex_amd.plan import synth_fill
    num_samp = 2 ** 27
    x = torch.empty((1, 2, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 31337)
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp) as f, plan_mod.FxPlan(2, nchan, ntaps, num_samp, path="generic") as g:
        assert f.path == path and g.path == "generic"
        rf, rg = f.fx_rows(x).cpu().numpy(), g.fx_rows(x).cpu().numpy()
        assert rel_err(rf, rg) < TOL_VIS
        f.fx_accumulate(x)
        assert rel_err(f.finalize("SPECTRUM"), rg.astype(np.complex128)) < TOL_VIS
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp + nchan) as p:
        assert p.path == "generic"


@pytest.mark.parametrize("ntaps,num_samp", [(4, 2 ** 29 + 6), (7, 2 ** 29 + 5)])
def test_streaming_path_on_streams_beyond_4_gib(plan_mod, torch, ntaps, num_samp):
    """nchan = 1 has no size limit of its own: 4 GiB per stream (64-bit indexing in stream1_t4_kernel / stream1_kernel)
    against the generic kernels."""
    from effex_amd.plan import synth_fill
    x = torch.empty((1, 2, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 99)
    w = np.linspace(0.4, 0.1, ntaps)
    with plan_mod.FxPlan(2, 1, ntaps, num_samp, window=w) as f, plan_mod.FxPlan(2, 1, ntaps, num_samp, window=w, path="generic") as g:
        assert f.path == "stream" and g.path == "generic"
        a = f.fx_rows(x, "CONTINUUM", 2.4e6).cpu().numpy()
        b = g.fx_rows(x, "CONTINUUM", 2.4e6).cpu().numpy()
    assert rel_err(a, b) < TOL_VIS


@pytest.mark.parametrize("nchan,ntaps,n_ant", [(4096, 4, 2), (2048, 4, 2), (2048, 32, 2), (8192, 4, 2), (4096, 4, 8), (64, 4, 2)])
def test_nan_samples_poison_their_chunk_only(plan_mod, torch, nchan, ntaps, n_ant):
    """numpy semantics of the reference (effex.py:508-527): one NaN sample makes every bin of its chunk's visibilities
    NaN (FIR -> all bins of the frames it touches -> the mean over frames) and leaves the other chunks' rows alone; an
    integration over the batch is NaN."""
    num_samp = nchan * 40 + 7
    x = torch.from_numpy(synth.synth_iq(4242, 5, n_ant, num_samp)).cuda()
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp) as p:
        clean = p.fx_rows(x).cpu().numpy()
        x[3, n_ant - 1, 17 * nchan + 5] = float("nan")
        rows = p.fx_rows(x).cpu().numpy()
        hit = [b for b in range(rows.shape[1])]              # baselines that include the last antenna are poisoned
        last = [i for i, (a, b) in enumerate((a, b) for a in range(n_ant) for b in range(a + 1, n_ant)) if b == n_ant - 1]
        assert np.isnan(rows[3][last]).all()
        others = [c for c in range(5) if c != 3]
        np.testing.assert_array_equal(rows[others], clean[others])
        rest = [i for i in hit if i not in last]
        if rest:
            np.testing.assert_array_equal(rows[3][rest], clean[3][rest])
        p.fx_accumulate(x)
        integ = p.finalize("SPECTRUM")
        assert np.isnan(integ[last]).all()


def test_caller_threads_with_a_plan_each(plan_mod, torch):
    """include/fxcorr.h: one caller thread per plan.  Four threads drive a plan each (fused and tiled kernels, a torch
    stream per thread) at the same time -- ctypes releases the GIL inside every fxc_* call -- and every result equals the one
    the same plan gave alone."""
    import threading
    shapes = [(4096, 4, 37), (2048, 4, 29), (4096, 4, 5), (1024, 8, 11)]
    jobs = []
    for k, (nchan, ntaps, n_chunks) in enumerate(shapes):
        num_samp = nchan * 24 + 3 * k
        x = torch.from_numpy(synth.synth_iq(600 + k, n_chunks, 2, num_samp)).cuda()
        plan = plan_mod.FxPlan(2, nchan, ntaps, num_samp)     # follows the calling thread's current torch stream
        want = plan.fx_rows(x).cpu().numpy()
        plan.fx_accumulate(x)
        want_int = plan.finalize("SPECTRUM")
        jobs.append((plan, x, want, want_int))
    errors = []

    def work(plan, x, want, want_int):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(25):
                    got = plan.fx_rows(x).cpu().numpy()
                    np.testing.assert_array_equal(got, want)
                    plan.fx_accumulate(x)
                    np.testing.assert_array_equal(plan.finalize("SPECTRUM"), want_int)
        except Exception as exc:      # surfaced in the main thread below
            errors.append(exc)

    threads = [threading.Thread(target=work, args=job) for job in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for plan, *_ in jobs:
        plan.close()
    assert not errors, errors


@pytest.mark.parametrize("nchan,ntaps,n_chunks,frames,extra", [(4096, 8, 19, 23, 424), (1024, 8, 300, 24, 10), (512, 9, 600, 31, 0),
                                                              (2048, 16, 5, 33, 100), (1024, 8, 7, 40, 1)])
def test_prefilter_with_more_workgroups_than_cus(plan_mod, torch, nchan, ntaps, n_chunks, frames, extra):
    """The pre-filter pass with 16-byte accesses (even num_samp, 8-frame block) and 8-byte ones, on launches of more than
    256 workgroups: the shapes on which a 16-byte buffer store with an SGPR offset lost its data to the next VALU write
    (tests/test_isa_hazards.py) -- spectra of every stream and frame against the generic kernels, then the rows."""
    num_samp = nchan * frames + extra
    x = torch.from_numpy(synth.synth_iq(7, n_chunks, 2, num_samp)).cuda()
    with plan_mod.FxPlan(1, nchan, ntaps, num_samp) as f, plan_mod.FxPlan(1, nchan, ntaps, num_samp, path="generic") as g:
        xs = x.reshape(-1, num_samp)
        sf, sg = f.channelize(xs).cpu().numpy(), g.channelize(xs).cpu().numpy()
        worst = np.abs(sf - sg).max(axis=2) / np.abs(sg).max()
        assert worst.max() < TOL_SPEC, np.argwhere(worst >= TOL_SPEC)[:10].tolist()
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp) as f, plan_mod.FxPlan(2, nchan, ntaps, num_samp, path="generic") as g:
        assert f.path == "tiled"
        assert rel_err(f.fx_rows(x).cpu().numpy(), g.fx_rows(x).cpu().numpy()) < TOL_VIS


def test_sharded_integration_equals_single_rank(plan_mod, torch):
    """SURVEY.md §8e on one GPU: two 'ranks' integrate disjoint chunk ranges, their exported sums are
    added (what the RCCL all-reduce does) and finalised once."""
    from effex_amd import sharding
    num_samp, n_chunks = 4096 * 8, 21
    x = torch.from_numpy(synth.synth_iq(3, n_chunks, 2, num_samp)).cuda()
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as whole:
        whole.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 1e-6)
        whole.fx_accumulate(x)
        ref = whole.finalize("SPECTRUM")
        total = None
        for rank in range(2):
            lo, hi = sharding.chunk_range(rank, 2, n_chunks)
            with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
                p.fx_accumulate(x[lo:hi])
                sums = p.acc_export(p.new_sums()).clone()
                torch.cuda.synchronize()
            total = sums if total is None else total + sums
        out = whole.finalize_sums(total, "SPECTRUM")
    assert rel_err(out, ref) < 1e-12


def test_reduce_through_rccl_on_the_plan_stream(plan_mod, torch):
    """fxc_comm_* / fxc_reduce (include/fxcorr.h): libfxcorr binds librccl itself, builds a
    communicator and enqueues ncclReduce / ncclAllReduce of the exported sums on the plan's stream.  One GPU here, so a
    world of one: what must hold is that the path runs and leaves the integration unchanged."""
    from effex_amd import sharding
    num_samp, n_chunks = 4096 * 8, 9
    x = torch.from_numpy(synth.synth_iq(5, n_chunks, 2, num_samp)).cuda()
    uid = plan_mod.RcclComm.unique_id()
    assert len(uid) == 128
    with plan_mod.RcclComm(0, 0, 1, uid) as comm, plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 1e-6)
        p.fx_accumulate(x)
        ref = p.finalize("SPECTRUM", reset=False)
        for root in (0, None):                         # ncclReduce to rank 0, ncclAllReduce
            p.reduce(comm, root)
            np.testing.assert_array_equal(p.finalize_sums(None, "SPECTRUM"), ref)
        p.reduce(None, 0)                              # no communicator: export only
        np.testing.assert_array_equal(p.finalize_sums(None, "SPECTRUM"), ref)
        # a CONTINUUM finalize of the accumulator in between neither clobbers the reduced sums nor makes them valid
        p.fx_accumulate(x[:2])
        cont = p.finalize("CONTINUUM", gi.BANDWIDTH, reset=False)
        assert np.isfinite(cont).all()
        np.testing.assert_array_equal(p.finalize_sums(None, "SPECTRUM"), ref)
        with plan_mod.FxPlan(2, 4096, 4, num_samp) as q:
            q.fx_accumulate(x)
            q.finalize("CONTINUUM", gi.BANDWIDTH, reset=False)
            with pytest.raises(plan_mod._lib.FxcError):
                q.finalize_sums(None, "SPECTRUM")       # no fxc_reduce has run on this plan
        p.acc_reset()
        integ = sharding.ShardedIntegrator(p, 0, 1, comm=comm)
        with pytest.raises(plan_mod._lib.FxcError):
            integ.finalize_wait()                      # nothing queued
        integ.accumulate(x)
        integ.finalize_async("SPECTRUM", gi.BANDWIDTH)
        with pytest.raises(plan_mod._lib.FxcError):
            integ.finalize("SPECTRUM", gi.BANDWIDTH)   # an asynchronous result is outstanding (as fxc_finalize refuses)
        np.testing.assert_array_equal(integ.finalize_wait(), ref)
        integ.accumulate(x)
        np.testing.assert_array_equal(integ.finalize("SPECTRUM", gi.BANDWIDTH), ref)
        with pytest.raises(ValueError):
            plan_mod.RcclComm(0, 1, 1, uid)            # rank outside the world


def test_plan_follows_the_callers_stream_and_device(plan_mod, torch):
    """A plan made without a stream issues every call on torch's *current* stream (fxc_set_stream), so inputs produced
    and outputs consumed under ``with torch.cuda.stream(s)`` are ordered with the kernels; and no ABI entry leaves
    the process on another current device."""
    num_samp, n_chunks = 4096 * 6, 40
    x = torch.from_numpy(synth.synth_iq(6, n_chunks, 2, num_samp)).cuda()
    side = torch.cuda.Stream()
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        base = p.fx_rows(x).cpu().numpy()
        default_stream = p._stream
        for _ in range(3):
            with torch.cuda.stream(side):
                big = torch.empty((64, 1024, 1024), device="cuda").normal_()      # keeps `side` busy ahead of the input
                xs = x * (2.0 + 0.0 * big[0, 0, 0])                                # produced on `side`, after `big`
                rows = p.fx_rows(xs)
                assert p._stream == side.cuda_stream
                got = rows.cpu().numpy()                                           # consumed on `side`
            np.testing.assert_array_equal(got, 4.0 * base)
            np.testing.assert_array_equal(p.fx_rows(x).cpu().numpy(), base)        # back on the default stream
            assert p._stream == default_stream
        assert torch.cuda.current_device() == 0
    with plan_mod.FxPlan(2, 4096, 4, num_samp, stream="owned") as q:
        with pytest.raises(Exception):
            q.set_stream(side.cuda_stream)                                         # a plan that owns its stream keeps it


def test_pipes_are_closed_before_their_plan(plan_mod, torch):
    """An fxc_pipe holds a pointer to its plan: fxc_plan_destroy refuses while one is alive, and FxPlan.close()
    closes its pipes first."""
    from effex_amd import _lib
    p = plan_mod.FxPlan(2, 512, 4, 4096)
    pipe = plan_mod.FxPipeline(p, 2, depth=2)
    lib = _lib.load()
    assert lib.fxc_plan_destroy(p._h) == _lib.FXC_ERR_STATE
    x = synth.synth_iq(8, 2, 2, 4096)
    pipe.push(x)
    assert pipe.pop().shape == (2, 1, 512)
    p.close()
    assert not pipe._h                        # closed by the plan
    pipe.close()                              # idempotent


def test_input_conditioning_on_device(plan_mod, torch):
    """uint8 ingestion and per-chunk DC removal (effex.py:394-395) without leaving HBM."""
    num_samp, n_chunks = 10007, 3
    rng = np.random.default_rng(12)
    u8 = rng.integers(0, 256, size=(n_chunks, 2, num_samp, 2), dtype=np.uint8)
    u8[0, 0, :, 0] = np.clip(u8[0, 0, :, 0].astype(int) // 2 + 100, 0, 255)      # a stream with a strong DC offset
    ud = torch.from_numpy(u8).cuda()
    with plan_mod.FxPlan(2, 256, 4, num_samp) as p:
        plain = p.convert_u8(ud, remove_dc=False).cpu().numpy()
        ref_plain = fx_oracle.u8_to_complex(u8)
        assert plain.shape == (n_chunks, 2, num_samp)
        np.testing.assert_allclose(plain, ref_plain, rtol=0, atol=6e-8)
        nodc = p.convert_u8(ud, remove_dc=True).cpu().numpy()
        x = torch.from_numpy(ref_plain.astype(np.complex64)).cuda()
        nodc2 = p.remove_dc(x.clone()).cpu().numpy()
        for c in range(n_chunks):
            for a in range(2):
                ref = fx_oracle.remove_dc(ref_plain[c, a])
                np.testing.assert_allclose(nodc[c, a], ref, rtol=0, atol=1e-7)
                np.testing.assert_allclose(nodc2[c, a], fx_oracle.remove_dc(ref_plain[c, a].astype(np.complex64)),
                                           rtol=0, atol=1.5e-7)
                assert abs(nodc[c, a].mean()) < 1e-7
        # conditioned input straight into the path
        rows = p.fx_rows(p.convert_u8(ud, remove_dc=True)).cpu().numpy()
        window = design_window(4, 256)
        ref0 = fx_oracle.pfb_xcorr(fx_oracle.remove_dc(ref_plain[0, 0]), fx_oracle.remove_dc(ref_plain[0, 1]), 4, 256,
                                   window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
        assert rel_err(rows[0, 0], ref0) < TOL_VIS


@pytest.mark.parametrize("nchan,frames,n_chunks,remove_dc", [(4096, 9, 3, True), (4096, 5, 300, True), (4096, 6, 2, False),
                                                             (1024, 12, 4, True), (2048, 7, 3, True), (512, 300, 1, True),
                                                             (8192, 3, 2, True), (64, 40, 2, True),
                                                             (1000, 12, 4, True), (96, 50, 3, False), (360, 9, 300, True), (6, 2000, 2, True)])
def test_fx_straight_from_rtlsdr_bytes(plan_mod, torch, nchan, frames, n_chunks, remove_dc):
    """fxc_fx_rows_u8 / fxc_fx_accumulate_u8: the byte stream of the reference's receivers (pyrtlsdr conversion behind
    effex.py:652, DC removal of effex.py:394-395) straight into F+X.  The headline shape, the tiled ring kernels and the mixed-radix
    F + X kernel (channel counts that are not a power of two) read the bytes inside the kernel (also across frame-range splits);
    other plans convert first.  Checked against the oracle chain and against the two-step device path."""
    num_samp = nchan * frames + 37
    rng = np.random.default_rng(2024 + nchan)
    u8 = rng.integers(0, 256, size=(n_chunks, 2, num_samp, 2), dtype=np.uint8)
    u8[0, 0, :, 0] = np.clip(u8[0, 0, :, 0].astype(int) // 2 + 90, 0, 255)        # a stream with a strong DC offset
    u8[:, 1, 5:, :] = (u8[:, 0, :-5, :] // 2 + u8[:, 1, 5:, :] // 2)              # common signal, 5 samples late
    ud = torch.from_numpy(u8).cuda()
    window = design_window(4, nchan)
    with plan_mod.FxPlan(2, nchan, 4, num_samp) as p:
        rows = p.fx_rows_u8(ud, "SPECTRUM", remove_dc=remove_dc).cpu().numpy()
        two_step = p.fx_rows(p.convert_u8(ud, remove_dc=remove_dc)).cpu().numpy()
        assert rel_err(rows, two_step) < TOL_VIS       # the fused ingest rounds b / 127.5 + off once in float32
        ref_plain = fx_oracle.u8_to_complex(u8[:2])
        for c in range(min(n_chunks, 2)):
            a0, a1 = ref_plain[c, 0], ref_plain[c, 1]
            if remove_dc:
                a0, a1 = fx_oracle.remove_dc(a0), fx_oracle.remove_dc(a1)
            ref = fx_oracle.pfb_xcorr(a0, a1, 4, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        rows_host = p.fx_rows_u8(u8, "SPECTRUM", remove_dc=remove_dc)            # host bytes in, host rows out
        np.testing.assert_array_equal(rows_host, rows)
        cont = p.fx_rows_u8(ud, "CONTINUUM", gi.BANDWIDTH, remove_dc=remove_dc).cpu().numpy()
        np.testing.assert_allclose(cont[:, 0], rows[:, 0].astype(np.complex128).mean(axis=1) / gi.BANDWIDTH,
                                   rtol=2e-5, atol=1e-7 * np.abs(cont).max())
        p.fx_accumulate_u8(ud[: n_chunks // 2], remove_dc=remove_dc)
        p.fx_accumulate_u8(ud[n_chunks // 2:], remove_dc=remove_dc)
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ[0], rows[:, 0].astype(np.complex128).mean(axis=0)) < 2e-6


def _offset_iq(seed, n_chunks, num_samp, dtype):
    """Synthetic chunk pairs with a different complex DC offset on every stream (the receivers' I/Q imbalance that
    effex.py:394-395 removes), in the source's sample type."""
    x = synth.synth_iq(seed, n_chunks, 2, num_samp).astype(np.complex128)
    rng = np.random.default_rng(seed)
    x += (rng.uniform(-0.3, 0.3, size=(n_chunks, 2, 1)) + 1j * rng.uniform(-0.3, 0.3, size=(n_chunks, 2, 1)))
    if dtype == np.complex128:
        x += 1e-9 * rng.standard_normal(x.shape)          # bits a complex64 does not hold
    return x.astype(dtype)


@pytest.mark.parametrize("nchan,frames,n_chunks,dtype", [(4096, 9, 3, np.complex64), (4096, 5, 300, np.complex64),
                                                         (4096, 7, 2, np.complex128), (1024, 12, 4, np.complex64),
                                                         (64, 40, 2, np.complex128), (96, 21, 2, np.complex64),
                                                         (1, 3000, 2, np.complex64)])
def test_fx_with_dc_removal_on_the_device(plan_mod, torch, nchan, frames, n_chunks, dtype):
    """fxc_fx_rows_iq / fxc_fx_accumulate_iq: effex.py:394-395 (per chunk, per antenna: x - mean(x.real) - 1j mean(x.imag))
    in front of _pfb_xcorr, on the device, for complex64 and complex128 samples, host and device buffers -- against the
    oracle chain remove_dc -> pfb_xcorr.  Tolerance: 1e-5 of max|vis| (TOL_VIS)."""
    num_samp = nchan * frames + (37 if nchan > 1 else 0)
    x = _offset_iq(77 + nchan, n_chunks, num_samp, dtype)
    c128 = dtype == np.complex128
    rng = np.random.default_rng(nchan)
    window = design_window(4, nchan) if nchan > 1 else rng.standard_normal(4)
    xd = torch.from_numpy(x).cuda()
    keep = xd.clone()
    with plan_mod.FxPlan(2, nchan, 4, num_samp, window=window) as p:
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 1e-7)
        rows = p.fx_rows(xd, "SPECTRUM", remove_dc=True, c128=c128).cpu().numpy()
        assert torch.equal(xd, keep)                      # the caller's samples are never written
        rows_host = p.fx_rows(x, "SPECTRUM", remove_dc=True, c128=c128)
        np.testing.assert_array_equal(rows_host, rows)
        pinned_in = plan_mod.pinned_empty(x.shape, dtype)
        pinned_in[...] = x
        pinned_out = plan_mod.pinned_empty(rows.shape, np.complex64)
        got = p.fx_rows(pinned_in, "SPECTRUM", remove_dc=True, c128=c128, out=pinned_out)
        assert got is pinned_out
        np.testing.assert_array_equal(pinned_out, rows)   # written by the device through the mapping
        for c in range(min(n_chunks, 2)):
            ref = fx_oracle.pfb_xcorr(fx_oracle.remove_dc(x[c, 0]), fx_oracle.remove_dc(x[c, 1]), 4, nchan, window,
                                      gi.BANDWIDTH, gi.FREQUENCY, 1e-7, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        with_mean = p.fx_rows(xd, "SPECTRUM", c128=c128).cpu().numpy()
        ref_mean = fx_oracle.pfb_xcorr(x[0, 0], x[0, 1], 4, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 1e-7, "SPECTRUM")
        assert rel_err(with_mean[0, 0], ref_mean) < TOL_VIS
        assert rel_err(with_mean[0, 0], rows[0, 0]) > 1e-3     # the offset matters: the flag is not a no-op
        cont = p.fx_rows(xd, "CONTINUUM", gi.BANDWIDTH, remove_dc=True, c128=c128).cpu().numpy()
        np.testing.assert_allclose(cont[:, 0], rows[:, 0].astype(np.complex128).mean(axis=1) / gi.BANDWIDTH,
                                   rtol=2e-5, atol=1e-7 * np.abs(cont).max())
        p.fx_accumulate(xd[: n_chunks // 2], remove_dc=True, c128=c128)
        p.fx_accumulate(x[n_chunks // 2:], remove_dc=True, c128=c128)
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ[0], rows[:, 0].astype(np.complex128).mean(axis=0)) < 2e-6


def test_dc_removal_with_a_small_workspace(plan_mod, torch):
    """Device buffers are de-meaned into a staging buffer in passes bounded by the workspace target: many passes give the
    rows of one pass (float32 summation order aside)."""
    num_samp, n_chunks = 4096 * 3, 37
    x = _offset_iq(5, n_chunks, num_samp, np.complex64)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        one = p.fx_rows(xd, remove_dc=True).cpu().numpy()
    import subprocess, sys, os      # the workspace target is read once per process: the 1 MiB run is a child
    code = ("import numpy as np, torch, sys; sys.path.insert(0, %r); from effex_amd import plan\n"
            "x = torch.from_numpy(np.load(sys.argv[1])).cuda()\n"
            "with plan.FxPlan(2, 4096, 4, x.shape[2]) as p: np.save(sys.argv[2], p.fx_rows(x, remove_dc=True).cpu().numpy())\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "x.npy"), x)
        subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "x.npy"), os.path.join(tmp, "r.npy")], check=True,
                       env=dict(os.environ, FXC_WS_MB="1"), timeout=600)
        many = np.load(os.path.join(tmp, "r.npy"))
    # (not bit for bit: a pass of 5 chunk pairs splits its frames over the workgroups differently from one of 37)
    assert many.shape == one.shape and rel_err(many, one) < TOL_VIS


def test_resolution_1000_at_full_size(plan_mod, torch):
    """BASELINE configs[1]'s size at a channel count that is not a power of two (nchan 1000, 1 024 chunk pairs of 2^18 samples,
    4.3 GB resident, generated on the device): properties that do not need the oracle at that size -- the integration is the
    mean of the rows, doubling the input quadruples every row bit for bit, swapping the antennas conjugates it -- and chunks
    0 and 1023 against the oracle."""
    from effex_amd.plan import synth_fill
    num_samp, n_chunks, nchan = 2 ** 18, 1024, 1000
    x = torch.empty((n_chunks, 2, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 4242)
    window = design_window(4, nchan)
    with plan_mod.FxPlan(2, nchan, 4, num_samp, window=window) as p:
        assert p.path == "generic"
        rows = p.fx_rows(x).cpu().numpy()
        p.fx_accumulate(x[:300])
        p.fx_accumulate(x[300:])
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ, rows.astype(np.complex128).mean(axis=0)) < 2e-6
        sub = x[500:516]
        rows16 = p.fx_rows(sub).cpu().numpy()          # (16 chunk pairs split their frames differently from 1 024: not bit for bit)
        assert rel_err(rows16, rows[500:516]) < 1e-6
        np.testing.assert_array_equal(p.fx_rows(sub * 2.0).cpu().numpy(), 4.0 * rows16)
        swapped = p.fx_rows(sub.flip(1).contiguous()).cpu().numpy()
        assert rel_err(swapped, np.conj(rows16)) < 1e-6
        for c in (0, n_chunks - 1):
            xc = x[c].cpu().numpy()
            ref = fx_oracle.pfb_xcorr(xc[0], xc[1], 4, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c


@pytest.mark.parametrize("nchan", [8192, 3000])
def test_new_routes_at_full_size(plan_mod, torch, nchan):
    """BASELINE configs[1]'s size (1 024 chunk pairs of 2^18 samples, 4.3 GB resident, generated on the device) on this round's routes
    -- --nfft 8192 in two passes, --resolution 3000 on the lean build of the kernel per channel count -- through properties that do
    not need the oracle at that size: the integration is the mean of the rows, doubling the input quadruples every row bit for
    bit, swapping the antennas conjugates it, the receivers' bytes agree with their conversion; chunks 0 and 1023 against the oracle."""
    from effex_amd.plan import synth_fill
    num_samp, n_chunks = 2 ** 18, 1024
    x = torch.empty((n_chunks, 2, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 777 + nchan)
    window = design_window(4, nchan)
    with plan_mod.FxPlan(2, nchan, 4, num_samp, window=window) as p:
        assert p.info["block"] == 512 and (nchan == 8192 or p.info["specialised"] & 1), p.info
        rows = p.fx_rows(x).cpu().numpy()
        p.fx_accumulate(x[:300])
        p.fx_accumulate(x[300:])
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ, rows.astype(np.complex128).mean(axis=0)) < 2e-6
        sub = x[500:516]
        rows16 = p.fx_rows(sub).cpu().numpy()          # (16 chunk pairs split their frames differently from 1 024: not bit for bit)
        assert rel_err(rows16, rows[500:516]) < 1e-6
        np.testing.assert_array_equal(p.fx_rows(sub * 2.0).cpu().numpy(), 4.0 * rows16)
        swapped = p.fx_rows(sub.flip(1).contiguous()).cpu().numpy()
        assert rel_err(swapped, np.conj(rows16)) < 1e-6
        u8 = torch.randint(0, 256, (64, 2, num_samp, 2), dtype=torch.uint8, device="cuda",
                           generator=torch.Generator(device="cuda").manual_seed(1729))      # (seeded: the bound below is a measured one)
        by = p.fx_rows_u8(u8, "SPECTRUM", remove_dc=True).cpu().numpy()
        assert rel_err(by, p.fx_rows(p.convert_u8(u8, remove_dc=True)).cpu().numpy()) < TOL_VIS
        for c in (0, n_chunks - 1):
            xc = x[c].cpu().numpy()
            ref = fx_oracle.pfb_xcorr(xc[0], xc[1], 4, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c


@pytest.mark.parametrize("n_ant", [2, 3])
def test_any_channel_count_with_a_small_workspace(plan_mod, torch, n_ant):
    """The mixed-radix path in passes bounded by the workspace target (1 MiB in a child process: raw sums only with two
    antennas, spectra + raw sums with three): rows and the integration equal the one-pass run's."""
    nchan, num_samp, n_chunks = 1000, 1000 * 40 + 3, 23
    x = synth.synth_iq(17, n_chunks, n_ant, num_samp)
    xd = torch.from_numpy(x).cuda()
    with plan_mod.FxPlan(n_ant, nchan, 4, num_samp) as p:
        one = p.fx_rows(xd).cpu().numpy()
        p.fx_accumulate(xd)
        integ = p.finalize("SPECTRUM")
    import subprocess, sys, os, tempfile      # the workspace target is read once per process
    code = ("import numpy as np, torch, sys; sys.path.insert(0, %r); from effex_amd import plan\n"
            "x = torch.from_numpy(np.load(sys.argv[1])).cuda()\n"
            "with plan.FxPlan(x.shape[1], 1000, 4, x.shape[2]) as p:\n"
            "    rows = p.fx_rows(x).cpu().numpy(); p.fx_accumulate(x); np.savez(sys.argv[2], rows=rows, integ=p.finalize('SPECTRUM'))\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "x.npy"), x)
        subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "x.npy"), os.path.join(tmp, "r.npz")], check=True,
                       env=dict(os.environ, FXC_WS_MB="1"), timeout=600)
        many = np.load(os.path.join(tmp, "r.npz"))
        assert many["rows"].shape == one.shape and rel_err(many["rows"], one) < 2e-6
        assert rel_err(many["integ"], integ) < 2e-6


def test_pinned_host_memory(plan_mod, torch):
    """fxc_host_alloc / fxc_host_free (effex.py:109-110: cusignal.get_shared_mem): buffers are ordinary host memory to the
    caller, results are bit-identical to the pageable route, a pointer the library did not hand out is refused."""
    import ctypes
    from effex_amd import _lib
    lib = _lib.load()
    num_samp = 4096 * 6
    x = synth.synth_iq(3, 2, 2, num_samp)
    a = plan_mod.pinned_empty(x.shape, np.complex64)
    a[...] = x
    np.testing.assert_array_equal(a, x)
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        ref = p.fx_rows(x, "CONTINUUM", gi.BANDWIDTH)
        out = plan_mod.pinned_empty(ref.shape, np.complex128)
        np.testing.assert_array_equal(p.fx_rows(a, "CONTINUUM", gi.BANDWIDTH, out=out), ref)
        inner = plan_mod.pinned_empty((4, 1, 4096), np.complex64)      # rows into the middle of a pinned block
        inner[...] = 7
        p.fx_rows(a, "SPECTRUM", out=inner[1:3])
        np.testing.assert_array_equal(inner[1:3], p.fx_rows(x, "SPECTRUM"))
        assert (inner[0] == 7).all() and (inner[3] == 7).all()
        with pytest.raises(ValueError):
            p.fx_rows(a, "SPECTRUM", out=np.empty((2, 1, 4095), np.complex64))
    assert lib.fxc_host_free(None) == 0
    assert lib.fxc_host_free(ctypes.c_void_p(a.ctypes.data + 64)) == _lib.FXC_ERR_ARG   # inside a block, not its start
    ptr = ctypes.c_void_p()
    assert lib.fxc_host_alloc(ctypes.byref(ptr), 0) == _lib.FXC_ERR_ARG
    assert lib.fxc_host_alloc(ctypes.byref(ptr), 1 << 20) == 0 and ptr.value
    assert lib.fxc_host_free(ptr) == 0
    assert lib.fxc_host_free(ptr) == _lib.FXC_ERR_ARG                                    # freed once


@pytest.mark.parametrize("fmt,dtype", [("c64", np.complex64), ("c128", np.complex128)])
def test_host_fed_pipeline_with_dc_removal(plan_mod, torch, fmt, dtype):
    """fxc_pipe_create_iq: complex recordings through the double-buffered front end with effex.py:394-395 on the device
    == the blocking call, batch for batch."""
    num_samp, chunks, n_batches = 4096 * 5, 3, 4
    x = _offset_iq(21, chunks * n_batches, num_samp, dtype).reshape(n_batches, chunks, 2, num_samp)
    c128 = fmt == "c128"
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        ref = [p.fx_rows(x[b], remove_dc=True, c128=c128) for b in range(n_batches)]
        with plan_mod.FxPipeline(p, chunks, depth=2, fmt=fmt, remove_dc=True) as pipe:
            got = []
            pipe.push(x[0])
            for b in range(1, n_batches):
                view = pipe.acquire()
                assert view.dtype == dtype
                view[...] = x[b]
                pipe.submit()
                got.append(pipe.pop())
            got.append(pipe.pop())
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g, r)
    ref0 = fx_oracle.pfb_xcorr(fx_oracle.remove_dc(x[0, 0, 0]), fx_oracle.remove_dc(x[0, 0, 1]), 4, 4096,
                               design_window(4, 4096), gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
    assert rel_err(got[0][0, 0], ref0) < TOL_VIS


@pytest.mark.parametrize("mode", ["SPECTRUM", "CONTINUUM"])
def test_drop_in_stage_then_run_task(torch, mode):
    """The drop-in's own per-pair call path (effex.py:391-395, 402-410): _stage narrows the source's complex128 chunk pair
    into the pinned gpu_iq buffers, _run_task hands them over and the mean comes off on the device -- equal to the
    reference's host de-mean followed by _pfb_xcorr (oracle chain), and unchanged by a second call or by rebinding."""
    from effex_amd.correlator import Correlator, SyntheticSource
    num_samp = 2 ** 16
    x = _offset_iq(31, 3, num_samp, np.complex128)
    cor = Correlator(num_samp=num_samp, source=SyntheticSource(), mode=mode)
    try:
        assert cor._pinned
        cor._state = 'RUN'
        for c in range(3):
            cor._stage((x[c, 0], x[c, 1]))
            assert cor._dc_pending and cor._gpu_iq[0].dtype == np.complex64 and np.shares_memory(cor._gpu_iq[0], cor._pair_buf)
            vis = cor._run_task()                       # nobody looked at the buffers: the device takes the mean off
            ref = fx_oracle.pfb_xcorr(fx_oracle.remove_dc(x[c, 0]), fx_oracle.remove_dc(x[c, 1]), 4, 4096, cor.window,
                                      cor.bandwidth, cor.frequency, 0.0, mode)
            assert rel_err(vis, ref) < TOL_VIS
            np.testing.assert_array_equal(cor._run_task(), vis)
            # looking at them through the public names shows what the reference holds there (effex.py:394-395: de-meaned
            # samples), and from then on _run_task computes exactly what they hold -- the same visibility
            held = np.array(cor.gpu_iq_0)
            assert not cor._dc_pending and np.shares_memory(cor.gpu_iq_0, cor._pair_buf)
            assert abs(held.mean()) < 1e-6 and rel_err(held, fx_oracle.remove_dc(x[c, 0])) < 1e-6
            assert rel_err(cor._run_task(), ref) < TOL_VIS
        # effex.py:391's idiom right after a staged pair: an in-place write is handed over as written, mean and all
        cor._stage((x[0, 0], x[0, 1]))
        cor.gpu_iq_0[:] = x[1, 0]
        cor.gpu_iq_1[:] = x[1, 1]
        ref = fx_oracle.pfb_xcorr(x[1, 0].astype(np.complex64), x[1, 1].astype(np.complex64), 4, 4096, cor.window,
                                  cor.bandwidth, cor.frequency, 0.0, mode)
        assert rel_err(cor._run_task(), ref) < TOL_VIS
        plain = cor._plan().fx_rows(np.stack([x[1, 0], x[1, 1]]).astype(np.complex64)[None], mode, cor.bandwidth)[0, 0]
        assert rel_err(cor._run_task(), plain) < 1e-6
        # rebinding (effex.py:394-395 style) hands over exactly what was bound: no DC removal behind the caller's back
        cor.gpu_iq_0, cor.gpu_iq_1 = x[0, 0], x[0, 1]
        ref = fx_oracle.pfb_xcorr(x[0, 0], x[0, 1], 4, 4096, cor.window, cor.bandwidth, cor.frequency, 0.0, mode)
        assert rel_err(cor._run_task(), ref) < TOL_VIS
        # ... and so does writing into the staging buffers in place after a rebind to them
        cor.gpu_iq_0, cor.gpu_iq_1 = cor._pair_buf[0, 0], cor._pair_buf[0, 1]
        cor.gpu_iq_0[:] = x[1, 0]
        cor.gpu_iq_1[:] = x[1, 1]
        ref = fx_oracle.pfb_xcorr(x[1, 0].astype(np.complex64), x[1, 1].astype(np.complex64), 4, 4096, cor.window,
                                  cor.bandwidth, cor.frequency, 0.0, mode)
        assert rel_err(cor._run_task(), ref) < TOL_VIS
        cor.remove_dc = False
        cor._stage((x[2, 0], x[2, 1]))
        ref = fx_oracle.pfb_xcorr(x[2, 0], x[2, 1], 4, 4096, cor.window, cor.bandwidth, cor.frequency, 0.0, mode)
        assert rel_err(cor._run_task(), ref) < TOL_VIS
    finally:
        cor.close()


@pytest.mark.parametrize("mode", ["SPECTRUM", "CONTINUUM"])
def test_host_fed_pipeline(plan_mod, torch, mode):
    """double-buffered host-fed front end == the blocking host path, batch for batch."""
    from effex_amd import _lib
    num_samp, chunks, n_batches = 4096 * 5, 3, 5
    x = synth.synth_iq(8, chunks * n_batches, 2, num_samp).reshape(n_batches, chunks, 2, num_samp)
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 1e-7)
        ref = [p.fx_rows(x[b], mode, gi.BANDWIDTH) for b in range(n_batches)]
        with plan_mod.FxPipeline(p, chunks, depth=2, mode=mode, bandwidth=gi.BANDWIDTH) as pipe:
            with pytest.raises(_lib.FxcError):
                pipe.pop()                                   # nothing in flight
            got = []
            pipe.push(x[0])
            for b in range(1, n_batches):
                pipe.push(x[b])                              # batch b in flight while b-1 completes
                assert pipe.in_flight == 2
                got.append(pipe.pop())
            with pytest.raises(ValueError):
                pipe.push(x[0][:2])
            got.append(pipe.pop())
            assert pipe.in_flight == 0
            pipe.acquire()[...] = x[2]                       # zero-copy producer path
            pipe.submit()
            np.testing.assert_array_equal(pipe.pop(), ref[2])
        for b in range(n_batches):
            np.testing.assert_array_equal(got[b], ref[b])


def test_host_fed_pipeline_on_bytes(plan_mod, torch):
    """fxc_pipe_create_u8: the double-buffered front end fed with the receivers' bytes equals fxc_fx_rows_u8."""
    num_samp, chunks, n_batches = 4096 * 5, 3, 4
    rng = np.random.default_rng(9)
    batches = [rng.integers(0, 256, size=(chunks, 2, num_samp, 2), dtype=np.uint8) for _ in range(n_batches)]
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        ref = [p.fx_rows_u8(b) for b in batches]
        with plan_mod.FxPipeline(p, chunks, depth=2, u8=True) as pipe:
            got = []
            pipe.push(batches[0])
            for b in batches[1:]:
                pipe.acquire()[...] = b
                pipe.submit()
                got.append(pipe.pop())
            got.append(pipe.pop())
            assert pipe.in_flight == 0
            with pytest.raises(ValueError):
                pipe.push(batches[0][:, :, :, 0])
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g, r)


def test_device_synth_is_bit_identical(plan_mod, torch):
    n_chunks, n_ant, num_samp = 3, 3, 5000
    x = torch.empty((n_chunks, n_ant, num_samp), dtype=torch.complex64, device="cuda")
    plan_mod.synth_fill(x, 424242, first_chunk=5)
    torch.cuda.synchronize()
    ref = synth.synth_iq(424242, n_chunks, n_ant, num_samp, first_chunk=5)
    np.testing.assert_array_equal(x.cpu().numpy(), ref)


def test_timers_and_kernel_profiling(plan_mod, torch):
    x = torch.from_numpy(synth.synth_iq(1, 4, 2, 4096 * 4)).cuda()
    with plan_mod.FxPlan(2, 4096, 4, 4096 * 4) as p:
        p.kernel_profiling(True)
        p.timer_start()
        p.fx_accumulate(x)
        p.fx_accumulate(x)
        ms = p.timer_stop()
        kms, n = p.kernel_time()
        assert n == 2 and 0 < kms <= ms * 1.05
        p.finalize("SPECTRUM")


def test_bench_two_ranks_share_one_gpu():
    """bench.py's real multi-rank flow (torch.distributed.run, per-rank frames of the synthetic stream, FxPlan +
    ShardedIntegrator, barrier / max-over-ranks timing, the all-reduced float64 check after the timed region) with two
    ranks on this one GPU over gloo.  RCCL wants a GPU per rank, so the run also takes bench.py's fall-back from
    fxc_reduce to the torch.distributed transport.  Timing is meaningless here; the checks are not."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2",
           "--warmup", "1", "--frames", "600"]
    proc = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["frames_per_gpu"] == 600 and line["config"]["path"] == "fused"
    assert line["config"]["reduce_transport"].startswith("torch.distributed; ranks share GPUs")
    # the line says which ranks RCCL saw: none here (gloo, shared GPU), with the reason, and not under the strict default
    rccl = line["rccl"]
    assert rccl["transport"] == "torch.distributed" and rccl["ranks_seen"] is None and rccl["ranks_summed"] is None
    assert "ranks share GPUs" in rccl["fallback_reason"] and rccl["strict"] is False and rccl["all_ranks_agree"] is False
    assert rccl["torch_backend"] == "gloo" and rccl["version"] >= 20000 and "rccl" in rccl["library"]
    assert all(r["rccl"]["ranks_seen"] is None and r["rccl"]["reduces_queued"] == 0 for r in line["ranks"]["per_rank"])
    assert line["verify"]["frames"] == 1200 and line["verify"]["integration_vs_float64_mean_of_rows"] < 1e-6
    assert line["value"] > 0 and line["roofline"]["launches"] == 2 and line["scaling"] == "weak"
    ranks = line["ranks"]
    assert [r["rank"] for r in ranks["per_rank"]] == [0, 1] and [r["first_frame"] for r in ranks["per_rank"]] == [0, 600]
    assert ranks["avg_kernel_ms"]["min"] > 0 and ranks["reduce_us"]["max"] > 0


def _launch_bench(n_ranks, *extra, timeout=900):
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(n_ranks), "--dist-backend",
           "gloo"] + list(extra)
    proc = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=timeout)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]                       # rank 0 alone prints, every child exited cleanly
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling,frames", [("weak", 64), ("strong", 530)])
def test_bench_eight_ranks_share_one_gpu(scaling, frames):
    """The launch the driver makes for the scaling curve -- torch.distributed.run with 8 ranks -- rehearsed on this one GPU
    over gloo (ranks share the GPU, so RCCL is out and the run takes the torch.distributed transport): the per-rank table,
    the frame bookkeeping, the all-reduced check after the timed region and a clean exit of all eight children."""
    line = _launch_bench(8, "--steps", "2", "--warmup", "1", "--frames", str(frames), "--scaling", scaling, "--no-power")
    total = frames * 8 if scaling == "weak" else frames
    assert line["n_gpus"] == 8 and line["scaling"] == scaling and line["config"]["frames_total"] == total
    assert line["verify"]["frames"] == total and line["verify"]["integration_vs_float64_mean_of_rows"] < 1e-6
    ranks = line["ranks"]["per_rank"]
    assert [r["rank"] for r in ranks] == list(range(8)) and sum(r["frames"] for r in ranks) == total
    assert [r["first_frame"] for r in ranks] == [sum(q["frames"] for q in ranks[:k]) for k in range(8)]
    assert max(r["frames"] for r in ranks) - min(r["frames"] for r in ranks) <= (0 if scaling == "weak" else 1)
    assert all(r["avg_kernel_ms"] > 0 and r["launches"] == 2 for r in ranks) and line["ranks"]["reduce_us"]["max"] > 0
    assert line["config"]["reduce_transport"].startswith("torch.distributed; ranks share GPUs")
    # the line says which ranks RCCL saw: none here (gloo, shared GPU), with the reason, and not under the strict default
    rccl = line["rccl"]
    assert rccl["transport"] == "torch.distributed" and rccl["ranks_seen"] is None and rccl["ranks_summed"] is None
    assert "ranks share GPUs" in rccl["fallback_reason"] and rccl["strict"] is False and rccl["all_ranks_agree"] is False
    assert rccl["torch_backend"] == "gloo" and rccl["version"] >= 20000 and "rccl" in rccl["library"]
    assert all(r["rccl"]["ranks_seen"] is None and r["rccl"]["reduces_queued"] == 0 for r in ranks)
    assert line["value"] > 0 and line["roofline"]["launches"] == 2


def test_bench_rows_mode_two_ranks_write_the_single_rank_file(tmp_path):
    """bench.py --rows (the time-series mode; effex.py:687-696): two ranks sharing this GPU fill disjoint
    windows of one row file through ShardedRows; the file is byte-identical to the one a single rank writes, and
    tools/rows_to_csv.py turns it into the csv the reference's writer produces for those rows."""
    import subprocess
    import sys
    import os
    from effex_amd import rowsink
    two, one = str(tmp_path / "two.fxb"), str(tmp_path / "one.fxb")
    line = _launch_bench(2, "--rows", "--rows-batch", "16", "--rows-file", two, "--steps", "1", "--warmup", "0", "--frames", "72",
                         "--scaling", "strong")
    assert line["unit"] == "rows/s" and line["n_gpus"] == 2 and line["verify"]["rows_in_file"] == 72
    assert [r["frames"] for r in line["ranks"]["per_rank"]] == [32, 40]          # whole batches of 16: 2 + 3 (the last one short)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", "--rows-batch", "16", "--rows-file", one,
                           "--steps", "1", "--warmup", "0", "--frames", "72", "--no-cpu-baseline"], cwd=root,
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert open(two, "rb").read() == open(one, "rb").read()
    back = rowsink.RowFile(two)
    out_csv, ref_csv = str(tmp_path / "two.csv"), str(tmp_path / "ref.csv")
    subprocess.run([sys.executable, os.path.join(root, "tools", "rows_to_csv.py"), two, out_csv], check=True, capture_output=True)
    with rowsink.CsvSink(ref_csv, back.header, back.freqs) as sink:          # the reference's writer (golden-tested in test_host.py)
        sink.write_rows(np.asarray(back.rows))
    assert open(out_csv, "rb").read() == open(ref_csv, "rb").read()
    # the rows themselves: frame 3 and the last one against the oracle
    window = design_window(4, 4096)
    for f in (3, 71):
        x = synth.synth_iq(1234, 1, 2, 262144, first_chunk=f)[0]
        ref = fx_oracle.pfb_xcorr(x[0], x[1], 4, 4096, window, 2.4e6, 1.4204e9, 0.0, "SPECTRUM")
        assert rel_err(np.asarray(back.rows[f]), ref) < TOL_VIS


@pytest.mark.gpu
@pytest.mark.parametrize("stream", ["owned", "raw", None])
def test_sharded_rows_wait_for_the_plans_own_stream(tmp_path, torch, stream):
    """ShardedRows over device-resident samples with a plan that does NOT follow torch's current stream (one it owns, or a raw
    stream handed to set_stream): the rows of a batch are written into a pinned slot by a kernel on the PLAN's stream, so the
    slot may only be handed to the file writers once that stream has passed it -- every row of the file against the oracle, and
    every batch read exactly once (effex.py:687-696 is the row-per-task writer this replaces)."""
    from effex_amd import rowsink, sharding
    from effex_amd.plan import FxPlan
    nchan, ntaps, num_samp, n_chunks, batch = 1024, 4, 1024 * 64, 40, 8
    window = design_window(ntaps, nchan)
    x = synth.synth_iq(4321, n_chunks, 2, num_samp)
    xd = torch.from_numpy(x).cuda()
    side = torch.cuda.Stream()
    plan = FxPlan(2, nchan, ntaps, num_samp, window=window, stream="owned" if stream == "owned" else None)
    try:
        if stream == "raw":
            plan.set_stream(side.cuda_stream)
        reads = []

        def read_chunks(lo, hi):
            reads.append((lo, hi))
            return xd[lo:hi]

        path = str(tmp_path / "rows.fxb")
        header = rowsink.header_line(1, 2.4e6, 1.4204e9, num_samp, nchan, 49.6, "SPECTRUM")
        freqs = rowsink.spectrum_freqs(nchan, 2.4e6, 1.4204e9)
        torch.cuda.synchronize()
        sharding.ShardedRows(plan, batch=batch).run(path, header, freqs, read_chunks, n_chunks, "SPECTRUM", 2.4e6)
        assert reads == [(lo, min(n_chunks, lo + batch)) for lo in range(0, n_chunks, batch)]
        back = np.asarray(rowsink.RowFile(path).rows)
        assert back.shape == (n_chunks, nchan)
        for c in range(n_chunks):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], ntaps, nchan, window, 2.4e6, 1.4204e9, 0.0, "SPECTRUM")
            assert rel_err(back[c], ref) < TOL_VIS, c
    finally:
        plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nchan,ntaps,n_chunks,frames,extra", [(6000, 4, 40, 9, 17), (5000, 4, 3, 70, 0), (4500, 2, 300, 2, 4499), (6561, 4, 5, 4, 0),
                                                                (7000, 3, 2, 1100, 5)])
def test_above_4096_channels_in_two_passes(plan_mod, torch, monkeypatch, nchan, ntaps, n_chunks, frames, extra):
    """--resolution 4097 ... 8192 off the powers of two, two antennas (effex.py:733-739, :508-521): antenna 0 through the F-only kernel
    built for the channel count, antenna 1 through its second-pass build whose last butterfly multiplies with antenna 0's spectra
    (fx_spec.h, FXM_XM; 2 x the algorithmic bytes) -- rows against the oracle, the integration against the float64 mean of the rows,
    runs longer than the float32 row limit, many chunks of few frames, a ragged tail, several workspace passes; and the route it
    replaces (both antennas' spectra + xmul_kernel: FXC_XM=0, developer library) agrees."""
    num_samp = nchan * frames + extra
    x = synth.synth_iq(nchan + n_chunks, n_chunks, 2, num_samp)
    xd = torch.from_numpy(x).cuda()
    window = design_window(ntaps, nchan)
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window) as p:
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 3e-7)
        rows = p.fx_rows(xd, "SPECTRUM").cpu().numpy()
        assert p.info["specialised"] & 4, p.info
        for c in sorted({0, n_chunks // 2, n_chunks - 1}):
            ref = fx_oracle.pfb_xcorr(x[c, 0], x[c, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 3e-7, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        p.fx_accumulate(xd[: n_chunks // 2])
        p.fx_accumulate(xd[n_chunks // 2:])
        assert rel_err(p.finalize("SPECTRUM"), rows.astype(np.complex128).mean(axis=0)) < 2e-6
        cont = p.fx_rows(xd[:1], "CONTINUUM", gi.BANDWIDTH).cpu().numpy()[0, 0]
        ref_c = fx_oracle.pfb_xcorr(x[0, 0], x[0, 1], ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 3e-7, "CONTINUUM")
        assert abs(cont - ref_c) < TOL_CONT * abs(ref_c) + 1e-8 * np.abs(rows[0, 0]).max() / gi.BANDWIDTH
    monkeypatch.setenv("FXC_XM", "0")      # (a route knob of the developer library: the shipped one reads none)
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window, dev=True) as q:
        q.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 3e-7)
        three = q.fx_rows(xd, "SPECTRUM").cpu().numpy()
        assert not (q.info["specialised"] & 4)
        assert rel_err(three, rows) < 2e-6


@pytest.mark.gpu
def test_byte_ingest_build_is_warmed_at_plan_creation(plan_mod, torch, tmp_path, monkeypatch):
    """A plan builds the byte-ingest twin of its kernel per channel count lazily, inside the first byte call (seconds of hiprtc in a
    live stream when the shape is neither pre-built nor cached).  ``FxPlan.warm_bytes()`` -- which the drop-in calls at plan
    creation for byte sources (pyrtlsdr's packed bytes, effex.py:652) -- runs the same search ahead of time: afterwards the code
    object is in the run-time cache and the first byte call agrees with the oracle."""
    monkeypatch.setenv("FXC_RTC_CACHE", str(tmp_path / "rtc"))
    nchan, ntaps, num_samp = 1040, 4, 1040 * 9 + 3          # (a channel count outside the pre-built list)
    window = design_window(ntaps, nchan)
    u8 = np.random.default_rng(1040).integers(0, 256, size=(2, 2, num_samp, 2), dtype=np.uint8)
    with plan_mod.FxPlan(2, nchan, ntaps, num_samp, window=window) as p:
        assert p.info["specialised"] & 1
        before = set(os.listdir(str(tmp_path / "rtc")))
        assert p.warm_bytes()
        assert set(os.listdir(str(tmp_path / "rtc"))) - before          # the byte-ingest build landed in the cache
        rows = p.fx_rows_u8(torch.from_numpy(u8).cuda(), "SPECTRUM", remove_dc=True).cpu().numpy()
    a = fx_oracle.u8_to_complex(u8[:1])[0]
    ref = fx_oracle.pfb_xcorr(fx_oracle.remove_dc(a[0]), fx_oracle.remove_dc(a[1]), ntaps, nchan, window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
    assert rel_err(rows[0, 0], ref) < TOL_VIS
