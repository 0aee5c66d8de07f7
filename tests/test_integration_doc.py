"""INTEGRATION.md's binding, executed: the ctypes stub a reference maintainer would paste into effex/effex.py is cut out
of the document, attached to a bare stand-in for the reference's Correlator (attributes only, effex.py:43-130) and
checked against the reference-executed goldens.  What is tested is the document's text, not effex_amd's own wrappers."""
import os
import re
import types

import numpy as np
import pytest

import golden_inputs as gi
from effex_amd import _lib
from effex_amd.window import design_window

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tolerances import TOL_CONT, TOL_SPEC, TOL_VIS


def _python_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def _stub_namespace():
    """Module-level part of the first block (imports, CDLL, argtypes, _fxc) plus its method definitions; the two
    statements meant for Correlator.__init__ become `_init_plan(self)`."""
    if not os.path.isfile(_lib.LIB_PATH):     # a fresh checkout running this file alone (never rebuilt under a loaded library)
        from effex_amd import build
        build.build()
    blocks = _python_blocks()
    assert len(blocks) >= 4, "INTEGRATION.md lost its code blocks"
    first = blocks[0].replace('"/path/to/effex_amd/csrc/libfxcorr.so"', repr(_lib.LIB_PATH))
    parts = re.split(r"^# --- ", first, flags=re.M)
    src = []
    for part in parts:
        if part.startswith("Correlator.__init__"):
            body = part.split("\n", 1)[1]
            src.append("def _init_plan(self):\n" + "".join("    " + line + "\n" for line in body.splitlines()))
        else:
            src.append(part if not part or part[0] in "#\n" else "# " + part)
    src += blocks[1:4]            # _estimate_delay_gaussian, _pfb_xcorr_bytes, pinned staging + _pfb_xcorr_staged
    ns = {}
    exec(compile("\n".join(src), "INTEGRATION.md", "exec"), ns)
    return ns


def _bare_correlator(ns, nbins, ntaps, num_samp, mode):
    cor = types.SimpleNamespace(nbins=nbins, ntaps=ntaps, num_samp=num_samp, mode=mode, bandwidth=gi.BANDWIDTH,
                                frequency=gi.FREQUENCY, calibrated_delay=0.0, window=design_window(ntaps, nbins))
    for name in ("_init_plan", "_pfb_xcorr", "_spectrometer_poly", "_estimate_delay_gaussian", "_pfb_xcorr_bytes", "_init_staging",
                 "_pfb_xcorr_staged"):
        setattr(cor, name, types.MethodType(ns[name], cor))
    return cor


def test_integration_md_parses_without_a_gpu():
    """CPU half: the document still has its blocks and they still compile against the in-tree library's symbols."""
    ns = _stub_namespace()
    for name in ("_fx", "_fxc", "_init_plan", "_pfb_xcorr", "_spectrometer_poly", "_estimate_delay_gaussian",
                 "_pfb_xcorr_bytes", "_pinned", "_init_staging", "_pfb_xcorr_staged"):
        assert name in ns, name


@pytest.mark.gpu
def test_integration_md_stub_against_reference_goldens(golden):
    meta, arrays = golden
    ns = _stub_namespace()
    iq = gi.xcorr_input()
    for item in meta["xcorr"]:
        cor = _bare_correlator(ns, 4096, 4, 2 ** 18, item["mode"])
        cor._init_plan()
        cor.calibrated_delay = item["delay"]
        cor.gpu_iq_0, cor.gpu_iq_1 = iq[0], iq[1]
        got = cor._pfb_xcorr()
        ref = arrays[item["key"]]
        if item["mode"] == "SPECTRUM":
            assert np.abs(got - ref).max() < TOL_VIS * np.abs(ref).max(), item
        else:
            floor = 1e-3 * 1e-5 * np.abs(arrays["xcorr_SPECTRUM_0"]).max() / gi.BANDWIDTH
            assert abs(got - ref) < TOL_CONT * abs(ref) + floor, item
        ns["_fx"].fxc_plan_destroy(cor._plan)


@pytest.mark.gpu
def test_integration_md_spectrometer_and_delay(golden):
    meta, arrays = golden
    ns = _stub_namespace()
    cor = _bare_correlator(ns, 4096, 4, 2 ** 18, "SPECTRUM")
    cor._init_plan()
    # the reference's first tone case through the patched _spectrometer_poly (tests/test_effex.py:62-89)
    num_samp, rate, freq, taps, branches = gi.tone_cases()[0]
    x = gi.tone_iq(num_samp, rate, freq)
    spec = cor._spectrometer_poly(x, taps, branches, design_window(taps, branches))
    import fx_oracle
    ref = fx_oracle.spectrometer_poly(x.astype(np.complex64), taps, branches, design_window(taps, branches))
    assert spec.shape == ref.shape
    assert np.abs(spec - ref).max() < TOL_SPEC * np.abs(ref).max()
    # delay: a 37-sample shift of a noise record (effex.py:583-627)
    rng = np.random.default_rng(5)
    n = 2 ** 16
    a = (rng.standard_normal(n + 100) + 1j * rng.standard_normal(n + 100)).astype(np.complex64)
    d = cor._estimate_delay_gaussian(a[37:37 + n], a[:n], 2.4e6)
    want = fx_oracle.estimate_delay_gaussian(a[37:37 + n], a[:n], 2.4e6)
    assert abs(d - want) < 2e-3 / 2.4e6
    # pinned staging + the DC removal of effex.py:394-395 on the device: the staged pair with an offset on each stream
    cor._init_staging()
    iq = gi.xcorr_input()
    cor.gpu_iq_0[:] = iq[0] + (0.2 - 0.1j)
    cor.gpu_iq_1[:] = iq[1] - (0.05 + 0.3j)
    got = cor._pfb_xcorr_staged()
    a0 = fx_oracle.remove_dc(np.asarray(cor.gpu_iq_0))
    a1 = fx_oracle.remove_dc(np.asarray(cor.gpu_iq_1))
    ref = fx_oracle.pfb_xcorr(a0, a1, 4, 4096, cor.window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
    assert np.abs(got - ref).max() < TOL_VIS * np.abs(ref).max()
    ns["_fx"].fxc_plan_destroy(cor._plan)
