"""ORACLE support — run the *unmodified* reference ``effex.py`` in the build container.

TEST INFRASTRUCTURE ONLY; works only where ``/root/reference`` exists (never on the GPU box).

The reference imports ``cupy``, ``cusignal`` and ``rtlsdr`` at module level
(``/root/reference/effex/effex.py:13-16``) and opens two RTL-SDR dongles in its constructor
(``:81-82``); none of that exists here.  This module injects stand-ins into ``sys.modules``:

* ``cupy``      -> numpy's namespace + ``asnumpy``            (cupy mirrors the numpy API by design)
* ``cusignal``  -> ``get_window``/``firwin`` = scipy.signal's (cusignal's are ports of them),
                   ``get_shared_mem`` = plain zeros, ``filtering.channelize_poly`` = the restated
                   definition in ``oracle/fx_oracle.py`` (the one function whose source is not in
                   the reference tree)
* ``rtlsdr``    -> a fake ``RtlSdr`` with settable attributes and ``close()``
* ``post_process`` is the reference's own module (needs matplotlib, present here)

No stand-in replaces anything that *is* in the reference tree: ``Correlator`` and every method
on it run as written.  No reference source is copied: the module is located by path at run time.
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np

REFERENCE_DIR = "/root/reference/effex"


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "effex.py"))


class _FakeRtlSdr(object):
    def __init__(self, device_index=0, dithering_enabled=False):
        self.device_index = device_index
        self.rs = None
        self.fc = None
        self.gain = None
        self.closed = False

    def close(self):
        self.closed = True


def _make_cupy():
    cp = types.ModuleType("cupy")
    for name in dir(np):
        if not name.startswith("__"):
            try:
                setattr(cp, name, getattr(np, name))
            except Exception:
                pass
    cp.asnumpy = np.asarray
    cp.fft = np.fft
    cp.random = np.random
    return cp


def _make_cusignal():
    import scipy.signal
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import fx_oracle

    cs = types.ModuleType("cusignal")
    cs.get_window = scipy.signal.get_window
    cs.firwin = scipy.signal.firwin
    cs.get_shared_mem = lambda n, dtype=np.float64: np.zeros(n, dtype=dtype)
    filt = types.ModuleType("cusignal.filtering")
    filt.channelize_poly = fx_oracle.channelize_poly
    cs.filtering = filt
    return cs, filt


def load_reference():
    """Import /root/reference/effex/effex.py through the stand-ins; returns the module.

    The constructor writes ``log_effex.log`` into the cwd (effex.py:62), so callers should
    ``chdir`` somewhere disposable first; ``make_correlator`` below does.
    """
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_DIR)
    cs, filt = _make_cusignal()
    rt = types.ModuleType("rtlsdr")
    rt.RtlSdr = _FakeRtlSdr
    sys.modules["cupy"] = _make_cupy()
    sys.modules["cusignal"] = cs
    sys.modules["cusignal.filtering"] = filt
    sys.modules["rtlsdr"] = rt
    import matplotlib
    matplotlib.use("Agg")
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    sys.modules.pop("effex", None)
    return importlib.import_module("effex")


def make_correlator(**kwargs):
    """Construct the reference's Correlator in a temp dir (it logs to cwd); returns (module, cor)."""
    fx = load_reference()
    tmp = tempfile.mkdtemp(prefix="effex_ref_")
    prev = os.getcwd()
    os.chdir(tmp)
    try:
        kwargs.setdefault("loglevel", "ERROR")
        cor = fx.Correlator(**kwargs)
    finally:
        os.chdir(prev)
    cor._oracle_tmpdir = tmp
    return fx, cor
