"""The C-ABI library loads and exports every symbol include/fxcorr.h declares (no GPU compute)."""
import ctypes
import os
import re

import pytest

from effex_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from effex_amd import build
    build.build()           # cross-compiles for gfx950 here; on the GPU box the prebuilt .so is current
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "fxcorr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fxc_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, "no ctypes signature for " + name
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_status_strings(lib):
    assert lib.fxc_version() == 106
    assert lib.fxc_status_string(0) == b"ok"
    assert b"unsupported" in lib.fxc_status_string(_lib.FXC_ERR_UNSUPPORTED)


def test_argument_validation_needs_no_device(lib):
    h = ctypes.c_void_p()
    win = (ctypes.c_double * (64 * 33))()
    # ntaps > 32: cusignal raises NotImplementedError (SURVEY.md §8b Errors)
    rc = lib.fxc_plan_create(ctypes.byref(h), 0, 2, 64, 33, 64 * 64, win, None, -1)
    assert rc == _lib.FXC_ERR_UNSUPPORTED
    with pytest.raises(NotImplementedError):
        _lib.check(rc, None)
    rc = lib.fxc_plan_create(ctypes.byref(h), 0, 2, 64, 4, 32, win, None, -1)     # shorter than one frame
    assert rc == _lib.FXC_ERR_ARG
    with pytest.raises(ValueError):
        _lib.check(rc, None)
    assert lib.fxc_plan_create(ctypes.byref(h), 0, 0, 64, 4, 4096, win, None, -1) == _lib.FXC_ERR_ARG
    assert lib.fxc_plan_create(ctypes.byref(h), 0, 2, 64, 4, 4096, None, None, -1) == _lib.FXC_ERR_ARG
    assert lib.fxc_plan_create(None, 0, 2, 64, 4, 4096, win, None, -1) == _lib.FXC_ERR_ARG
    assert lib.fxc_sync(None) == _lib.FXC_ERR_ARG
    assert lib.fxc_set_stream(None, None) == _lib.FXC_ERR_ARG
    assert lib.fxc_reduce(None, None, 0) == _lib.FXC_ERR_ARG
    assert lib.fxc_comm_unique_id(None) == _lib.FXC_ERR_ARG
    assert lib.fxc_comm_create(None, 0, 0, 1, None) == _lib.FXC_ERR_ARG
    assert lib.fxc_comm_destroy(None) == _lib.FXC_OK
    assert lib.fxc_comm_info(None, None) == _lib.FXC_ERR_ARG and lib.fxc_comm_probe(None, None) == _lib.FXC_ERR_ARG
    assert lib.fxc_rccl_version(None, None, 0) == _lib.FXC_ERR_ARG
    assert b"RCCL" in lib.fxc_status_string(_lib.FXC_ERR_COMM)
    assert lib.fxc_plan_destroy(None) == _lib.FXC_OK
    assert lib.fxc_fx_rows_iq(None, None, None, 1, 0, 0, 1.0, _lib.FXC_IQ_C128, 1) == _lib.FXC_ERR_ARG
    assert lib.fxc_fx_rows_iq(None, None, None, 1, 0, 0, 1.0, 7, 0) == _lib.FXC_ERR_ARG          # no such sample format
    assert lib.fxc_fx_accumulate_iq(None, None, 1, 0, _lib.FXC_IQ_C64, 1) == _lib.FXC_ERR_ARG
    assert lib.fxc_pipe_create_iq(None, None, 1, 2, 0, 1.0, _lib.FXC_IQ_C64, 1) == _lib.FXC_ERR_ARG
    assert lib.fxc_host_alloc(None, 64) == _lib.FXC_ERR_ARG
    assert lib.fxc_host_free(None) == _lib.FXC_OK


def test_no_cpu_backend(lib):
    """Without a HIP device the product path fails loudly instead of computing on the CPU."""
    n = ctypes.c_int(-1)
    assert lib.fxc_device_count(ctypes.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    win = (ctypes.c_double * (64 * 4))()
    rc = lib.fxc_plan_create(ctypes.byref(h), 0, 2, 64, 4, 4096, win, None, -1)
    assert rc == _lib.FXC_ERR_NODEVICE
    assert b"no CPU backend" in lib.fxc_last_error(None)
    from effex_amd.plan import FxPlan, pinned_empty
    with pytest.raises(_lib.FxcError):
        FxPlan(2, 64, 4, 4096)
    # pinned staging memory is the HIP runtime's to give: without a device there is none, and the drop-in class falls back
    # to ordinary arrays for its *buffers* (never for arithmetic)
    ptr = ctypes.c_void_p()
    assert lib.fxc_host_alloc(ctypes.byref(ptr), 4096) == _lib.FXC_ERR_NODEVICE and not ptr.value
    with pytest.raises(_lib.FxcError):
        pinned_empty((16,), "complex64")
    from effex_amd.correlator import Correlator, SyntheticSource
    cor = Correlator(source=SyntheticSource())
    assert not cor._pinned and cor.gpu_iq_0.dtype == "complex64" and len(cor.gpu_iq_0) == 2 ** 18
    with pytest.raises(_lib.FxcError):
        cor._run_task()


def test_bench_refuses_a_foreign_library(monkeypatch, tmp_path):
    """bench.py measures the in-tree build only: FXCORR_LIB (the developer A/B override) makes it exit."""
    import subprocess
    import sys
    env = dict(os.environ, FXCORR_LIB=str(tmp_path / "libother.so"))
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "1"],
                          env=env, capture_output=True, text=True, timeout=120)
    assert proc.returncode != 0 and "in-tree library only" in proc.stderr
    assert _lib.is_in_tree()


def test_header_and_library_from_plain_c(lib, tmp_path):
    """include/fxcorr.h is a C header (C99, -pedantic) and libfxcorr.so links from a C program: what a cgo / JNI / FFI
    binding of the reference's language would rely on."""
    _run_c_program(tmp_path)


@pytest.mark.gpu
def test_plan_from_plain_c_on_the_gpu(lib, tmp_path):
    """The same C program on a GPU box: creates a plan, reads its info, destroys it."""
    _run_c_program(tmp_path)


def _hip_env():
    env = dict(os.environ)
    # the process needs a HIP runtime for libfxcorr's own dependency: torch's copy or ROCm's
    try:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    except ImportError:
        pass
    return env


def _c_array(name, ctype, values):
    return "static const %s %s[] = {%s};\n" % (ctype, name, ", ".join(repr(float(v)) for v in values))


def _build_compute_program(tmp_path):
    """tests/cabi/compute_fxcorr.c against a header and a data file generated here from the committed reference goldens
    (tests/golden/reference_outputs.npz) and the seeded inputs they were made from (oracle/golden_inputs.py)."""
    import shutil
    import subprocess
    import sys
    import numpy as np
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import golden_inputs as gi
    from effex_amd.plan import rot_table
    from effex_amd.window import design_window
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    arrays = np.load(os.path.join(ROOT, "tests", "golden", "reference_outputs.npz"))
    x, h = gi.kat_input()
    kat = arrays["kat_spec"]
    x32 = x.astype(np.complex64)
    with open(str(tmp_path / "golden_kat.h"), "w") as fh:
        fh.write("/* generated by tests/test_abi.py from tests/golden/reference_outputs.npz (kat_spec) and oracle/golden_inputs.py::kat_input */\n")
        fh.write("#define KAT_NCHAN 4\n#define KAT_NTAPS 2\n#define KAT_NUM_SAMP %d\n#define KAT_FRAMES %d\n" % (len(x), kat.shape[0]))
        fh.write(_c_array("kat_window", "double", h))
        fh.write(_c_array("kat_x", "float", x32.view(np.float32)))
        fh.write(_c_array("kat_spec", "double", kat.astype(np.complex128).ravel().view(np.float64)))
    nchan, ntaps, num_samp = 4096, 4, 2 ** 18
    iq = np.ascontiguousarray(gi.xcorr_input(), dtype=np.complex64)
    blob = str(tmp_path / "headline.bin")
    with open(blob, "wb") as fh:
        np.array([num_samp, nchan, ntaps], dtype=np.int64).tofile(fh)
        np.ascontiguousarray(design_window(ntaps, nchan), dtype=np.float64).tofile(fh)
        np.ascontiguousarray(rot_table(nchan, gi.BANDWIDTH, gi.FREQUENCY, 0.0), dtype=np.complex128).tofile(fh)
        iq.tofile(fh)
        np.ascontiguousarray(arrays["xcorr_SPECTRUM_0"], dtype=np.complex128).tofile(fh)
    exe = str(tmp_path / "compute_fxcorr")
    libdir = os.path.dirname(_lib.IN_TREE_LIB)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
                    os.path.join(ROOT, "tests", "cabi", "compute_fxcorr.c"), "-o", exe, "-L", libdir, "-lfxcorr", "-lm",
                    "-Wl,-rpath," + libdir], check=True)
    return exe, blob


def test_computing_c_program_builds(lib, tmp_path):
    """The computing C consumer compiles as C99 -pedantic against the generated golden header and links (it needs a GPU to run:
    without one it says so and exits 77)."""
    import subprocess
    exe, blob = _build_compute_program(tmp_path)
    proc = subprocess.run([exe, blob], env=_hip_env(), capture_output=True, text=True, timeout=120)
    assert proc.returncode in (0, 77), (proc.returncode, proc.stdout, proc.stderr)


@pytest.mark.gpu
def test_c_program_computes_on_the_gpu(lib, tmp_path):
    """From plain C, no Python in the process: fxc_channelize on the N = 4 / T = 2 known-answer case against the reference-executed
    `kat_spec` (effex.py:530-555), and fxc_set_rot + fxc_fx_rows (FXC_MEM_HOST) on the headline shape against `xcorr_SPECTRUM_0`
    (_run_task, effex.py:490-521); tolerance 1e-5 of max|expected|, stated in the C source."""
    import subprocess
    exe, blob = _build_compute_program(tmp_path)
    proc = subprocess.run([exe, blob], env=_hip_env(), capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, (proc.returncode, proc.stdout, proc.stderr)
    assert "c-abi compute ok" in proc.stdout and "kat_spec err" in proc.stdout and "xcorr_SPECTRUM_0 err" in proc.stdout


def _run_c_program(tmp_path):
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "use_fxcorr")
    libdir = os.path.dirname(_lib.IN_TREE_LIB)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cabi", "use_fxcorr.c"), "-o", exe, "-L", libdir, "-lfxcorr",
                    "-Wl,-rpath," + libdir], check=True)
    env = dict(os.environ)
    # the process needs a HIP runtime for libfxcorr's own dependency: torch's copy or ROCm's
    try:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    except ImportError:
        pass
    proc = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, (proc.returncode, proc.stdout, proc.stderr)
    assert "c-abi ok" in proc.stdout
