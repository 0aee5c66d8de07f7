import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# kernels compiled at run time (fx_spec.h through hiprtc) are kept on disk between processes: inside the repository for the tests
os.environ.setdefault("FXC_RTC_CACHE", os.path.join(ROOT, "build", "rtc_cache"))


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build what is MISSING -- the shipped library and the developer
    build the A/B tests load -- so that the suite does not depend on __graft_entry__.build() having run first.  (Existing builds are
    left alone: the GPU box gets them with the snapshot.)"""
    try:
        from effex_amd import build as fx_build
        if not os.path.isfile(fx_build.LIB):
            fx_build.build()
        if not os.path.isfile(fx_build.LIB_DEV):
            fx_build.build(dev=True)
    except Exception as exc:      # (no hipcc: the tests that need the library say so themselves)
        print("conftest: could not build libfxcorr: %s" % exc)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _name_the_running_test(request):
    """tests/tolerances.py looks the bound of a comparison up by the test function that makes it."""
    import tolerances
    tolerances._current[0] = request.node.originalname or request.node.name
    yield
    tolerances._current[0] = None


def pytest_sessionfinish(session, exitstatus):
    """Every comparison that went through a tests/tolerances.py bound, with the bound it met: gpurun_out/observed_errors.json
    (what tools/make_tolerances.py turns into tests/golden/tolerances.json and the table of DESIGN.md)."""
    try:
        import tolerances
        if tolerances.observed and _has_gpu():
            tolerances.dump(os.path.join(ROOT, "gpurun_out", "observed_errors.json"))
    except Exception as exc:        # a reporting aid must never fail a run
        sys.stderr.write("observed_errors.json not written: %s\n" % exc)


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    gdir = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gdir, "reference_outputs.json")) as fh:
        meta = json.load(fh)
    arrays = dict(np.load(os.path.join(gdir, "reference_outputs.npz")))
    return meta, arrays
