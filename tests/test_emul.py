"""Host emulation of the fused kernel's phases (tests/emul/emul_fused.cpp compiles the kernel's own
header with g++) against the oracle: validates the FFT decomposition, LDS layouts, ring rotation and
permlane pairing on a machine without a GPU.  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import fx_oracle
from effex_amd import synth
from effex_amd.window import design_window

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(HERE, "emul", "emul_fused.cpp")
    lib = os.path.join(HERE, "emul", "libemul_fused.so")
    deps = [src, os.path.join(HERE, "..", "effex_amd", "csrc", "fx_fused4096.h"),
            os.path.join(HERE, "..", "effex_amd", "csrc", "fx_math.h")]
    if not os.path.isfile(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", lib, src], check=True)
    return ctypes.CDLL(lib)


def test_dft16_sign_and_order(emul):
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(16) + 1j * rng.standard_normal(16)).astype(np.complex64)
    out = np.zeros(16, np.complex64)
    emul.emul_dft16(a.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    ref = np.fft.ifft(a.astype(np.complex128)) * 16          # kernel exp(+2 pi i n k / 16)
    assert np.abs(out - ref).max() < 2e-6


def test_output_layouts_are_consistent(emul):
    assert emul.emul_layout_check() == 0


@pytest.mark.parametrize("num_samp", [4096, 4096 * 3 + 17, 4096 * 9])
def test_fused_phases_match_oracle(emul, num_samp):
    x = synth.synth_iq(1234, 1, 2, num_samp)[0]
    w = design_window(4, 4096)
    out = np.zeros(4096, np.complex128)
    rc = emul.emul_fused4096(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(num_samp),
                             w.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    f0 = fx_oracle.spectrometer_poly(x[0], 4, 4096, w)
    f1 = fx_oracle.spectrometer_poly(x[1], 4, 4096, w)
    ref = (f0 * np.conj(f1)).sum(axis=0)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-6


@pytest.fixture(scope="module")
def emul_tiled():
    src = os.path.join(HERE, "emul", "emul_tiled.cpp")
    lib = os.path.join(HERE, "emul", "libemul_tiled.so")
    deps = [src] + [os.path.join(HERE, "..", "effex_amd", "csrc", h) for h in ("fx_tiled.h", "fx_fused4096.h", "fx_math.h")]
    if not os.path.isfile(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", lib, src], check=True)
    return ctypes.CDLL(lib)


@pytest.mark.parametrize("nchan,ntaps,frames,ring", [
    (512, 4, 9, 0), (512, 4, 9, 1), (1024, 4, 6, 1), (2048, 4, 5, 1), (2048, 3, 5, 1), (1024, 1, 3, 1), (4096, 8, 4, 0), (4096, 4, 6, 1),
    (8192, 4, 3, 0), (2048, 32, 3, 0), (512, 7, 12, 0)])
def test_tiled_phases_match_oracle(emul_tiled, nchan, ntaps, frames, ring):
    """fx_tiled.h (the other --nfft values, effex.py:778): decomposition, padded exchange layout, bin mapping,
    and the ntaps <= 4 ring variant, on the host."""
    num_samp = nchan * frames + 13
    x = synth.synth_iq(99, 1, 2, num_samp)[0]
    w = design_window(ntaps, nchan)
    out = np.zeros(nchan, np.complex128)
    rc = emul_tiled.emul_tiled(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(num_samp), nchan, ntaps,
                               w.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ring)
    assert rc == 0
    f0 = fx_oracle.spectrometer_poly(x[0], ntaps, nchan, w)
    f1 = fx_oracle.spectrometer_poly(x[1], ntaps, nchan, w)
    ref = (f0 * np.conj(f1)).sum(axis=0)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-6


@pytest.fixture(scope="module")
def emul_small():
    src = os.path.join(HERE, "emul", "emul_small.cpp")
    lib = os.path.join(HERE, "emul", "libemul_small.so")
    deps = [src] + [os.path.join(HERE, "..", "effex_amd", "csrc", h) for h in ("fx_small.h", "fx_tiled.h", "fx_fused4096.h", "fx_math.h")]
    if not os.path.isfile(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", lib, src], check=True)
    return ctypes.CDLL(lib)


@pytest.mark.parametrize("nchan,ntaps,frames", [(16, 4, 40), (32, 4, 33), (64, 4, 21), (128, 4, 12), (256, 4, 9), (256, 3, 5),
                                                (64, 1, 7), (32, 2, 6)])
def test_small_phases_match_oracle(emul_small, nchan, ntaps, frames):
    """fx_small.h (--nfft 16 ... 256 inside one wave): radix-16, twiddle, transposition rows inside the item's lanes,
    the transforms of P points and the bin mapping, on the host."""
    num_samp = nchan * frames + 5
    x = synth.synth_iq(77, 1, 2, num_samp)[0]
    w = design_window(ntaps, nchan)
    out = np.zeros(nchan, np.complex128)
    rc = emul_small.emul_small(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(num_samp), nchan, ntaps,
                               w.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    f0 = fx_oracle.spectrometer_poly(x[0], ntaps, nchan, w)
    f1 = fx_oracle.spectrometer_poly(x[1], ntaps, nchan, w)
    ref = (f0 * np.conj(f1)).sum(axis=0)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-6


@pytest.fixture(scope="module")
def emul_sched():
    src = os.path.join(HERE, "emul", "emul_sched.cpp")
    lib = os.path.join(HERE, "emul", "libemul_sched.so")
    deps = [src] + [os.path.join(HERE, "..", "effex_amd", "csrc", h) for h in ("fx_fused4096.h", "fx_math.h")]
    if not os.path.isfile(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", lib, src], check=True)
    return ctypes.CDLL(lib)


@pytest.mark.parametrize("n_chunks,n_pts,grid,seg,unit", [
    (10000, 64, 256, 1, 4), (10000, 64, 256, 1, 1), (10000, 64, 256, 4, 4), (10240, 64, 256, 1, 4), (257, 64, 256, 1, 1),
    (255, 64, 256, 1, 4), (1, 64, 16, 1, 1), (1, 64, 16, 1, 4), (3, 16, 12, 1, 1), (7, 5, 8, 1, 1), (23, 5, 8, 1, 3),
    (100, 1, 25, 1, 1), (103, 1, 25, 2, 64), (5, 3, 256, 1, 1), (2, 1, 256, 1, 4), (1000, 256, 256, 1, 1), (33, 7, 31, 2, 2),
    (700, 2, 256, 3, 64), (517, 3, 256, 1, 64)])
@pytest.mark.parametrize("rows_are_chunks", [1, 0])
def test_fused_work_split(emul_sched, n_chunks, n_pts, grid, seg, unit, rows_are_chunks):
    """fx_fused4096_kernel's work split (whole chunks round-robin, then equal frame ranges of the tail that ignore
    chunk boundaries; rows + leading-part rows; ring history at range starts and across segment jumps): every frame
    once, per-chunk and total sums rebuilt as the finishing kernels do."""
    rng = np.random.default_rng(n_chunks * 131 + n_pts * 7 + grid + unit)
    w = rng.integers(1, 1000, size=n_chunks * n_pts).astype(np.float64)
    per_chunk = np.zeros(n_chunks, np.float64)
    total = ctypes.c_double()
    fmin, fmax, n_rows = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    rc = emul_sched.emul_fused_schedule(n_chunks, n_pts, grid, seg, unit, rows_are_chunks,
                                        w.ctypes.data_as(ctypes.c_void_p), per_chunk.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.byref(total), ctypes.byref(fmin), ctypes.byref(fmax), ctypes.byref(n_rows))
    assert rc == 0
    assert total.value == w.sum()
    assert fmax.value - fmin.value <= 1                      # balanced whatever n_chunks % grid is
    if rows_are_chunks:
        assert n_rows.value == n_chunks + grid
        assert np.array_equal(per_chunk, w.reshape(n_chunks, n_pts).sum(axis=1))
