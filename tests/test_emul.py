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
def emul_mixed():
    src = os.path.join(HERE, "emul", "emul_mixed.cpp")
    lib = os.path.join(HERE, "emul", "libemul_mixed.so")
    deps = [src] + [os.path.join(HERE, "..", "effex_amd", "csrc", h) for h in ("fx_mixed.h", "fx_math.h")]
    if not os.path.isfile(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib, src], check=True)
    return ctypes.CDLL(lib)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 30, 96, 100, 125, 127, 210, 500, 997, 1000, 1001, 1536, 2000,
                               2310, 2431, 3000, 4096, 5000, 6561, 8190, 10240, 13122, 16383, 16384])
def test_mixed_radix_stages_match_numpy(emul_mixed, n):
    """The generic path's FFT for any channel count (fx_mixed.h): the factorisation multiplies back to n, and the Stockham
    stages run thread by thread give numpy's transform (kernel exp(+2 pi i n k / N), natural order out) -- for the register
    radices, for prime factors walked from the row (127, 997, 17) and for the row-sharing thread counts of small n."""
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ref = np.fft.ifft(x.astype(np.complex128)) * n
    radices = (ctypes.c_int * 16)()
    for tpr in sorted({emul_mixed.emul_mixed_threads_per_row(n), 1, 256}):
        out = np.zeros(n, np.complex64)
        stages = emul_mixed.emul_mixed_fft(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), n, tpr, radices)
        assert stages >= 0 and int(np.prod(list(radices)[:stages], dtype=np.int64)) == n
        assert np.abs(out - ref).max() <= 4e-6 * max(np.abs(ref).max(), 1e-30), (n, tpr)
    # two rows a slot carries through the stages together (one index computation, one set of twiddles)
    x2 = np.stack([x, x[::-1] * (0.5 - 0.25j)]).astype(np.complex64)
    out2 = np.zeros_like(x2)
    assert emul_mixed.emul_mixed_fft_two_rows(x2.ctypes.data_as(ctypes.c_void_p), out2.ctypes.data_as(ctypes.c_void_p), n,
                                              emul_mixed.emul_mixed_threads_per_row(n)) >= 0
    ref2 = np.fft.ifft(x2.astype(np.complex128), axis=1) * n
    assert np.abs(out2 - ref2).max() <= 4e-6 * max(np.abs(ref2).max(), 1e-30), n


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


def test_emulation_sources_under_address_and_undefined_sanitizers(tmp_path):
    """Sanitizers belong on the CPU build (GPU AddressSanitizer is not available on the pool): the kernels' own phase headers
    (fx_math.h, fx_fused4096.h, fx_tiled.h, fx_small.h, fx_mixed.h), compiled by g++ for the host emulation, run a representative set of
    shapes under -fsanitize=address,undefined in a child process (libasan preloaded).  Index maps that step outside an LDS
    image, a ring slot or a raw row show up here as reports, not as silently wrong spectra."""
    import shutil
    import sys
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    libasan = subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan for this g++")
    libs = {}
    for name in ("emul_fused", "emul_tiled", "emul_small", "emul_sched", "emul_mixed"):
        out = str(tmp_path / ("lib%s_san.so" % name))
        subprocess.run([gxx, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-fPIC",
                        "-o", out, os.path.join(HERE, "emul", name + ".cpp")], check=True)
        libs[name] = out
    driver = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, %(root)r)
from effex_amd import synth
from effex_amd.window import design_window
libs = %(libs)r
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
f = ctypes.CDLL(libs["emul_fused"])
a = (np.arange(16) + 1j * np.arange(16)[::-1]).astype(np.complex64); o = np.zeros(16, np.complex64)
f.emul_dft16(vp(a), vp(o))
assert f.emul_layout_check() == 0
for num_samp in (4096, 4096 * 3 + 17):
    x = synth.synth_iq(1234, 1, 2, num_samp)[0]; w = design_window(4, 4096); out = np.zeros(4096, np.complex128)
    assert f.emul_fused4096(vp(x), ctypes.c_int64(num_samp), vp(w), vp(out)) == 0
t = ctypes.CDLL(libs["emul_tiled"])
for nchan, ntaps, frames, ring in ((512, 4, 9, 1), (1024, 4, 6, 1), (2048, 32, 3, 0), (4096, 4, 6, 1), (8192, 4, 3, 0), (512, 7, 12, 0)):
    num_samp = nchan * frames + 13
    x = synth.synth_iq(99, 1, 2, num_samp)[0]; w = design_window(ntaps, nchan); out = np.zeros(nchan, np.complex128)
    assert t.emul_tiled(vp(x), ctypes.c_int64(num_samp), nchan, ntaps, vp(w), vp(out), ring) == 0
s = ctypes.CDLL(libs["emul_small"])
for nchan, ntaps, frames in ((16, 4, 40), (32, 4, 33), (64, 4, 21), (128, 4, 12), (256, 3, 5)):
    num_samp = nchan * frames + 5
    x = synth.synth_iq(77, 1, 2, num_samp)[0]; w = design_window(ntaps, nchan); out = np.zeros(nchan, np.complex128)
    assert s.emul_small(vp(x), ctypes.c_int64(num_samp), nchan, ntaps, vp(w), vp(out)) == 0
c = ctypes.CDLL(libs["emul_sched"])
for n_chunks, n_pts, grid, seg, unit, rac in ((10000, 64, 256, 1, 4, 0), (257, 64, 256, 1, 1, 1), (23, 5, 8, 1, 3, 0), (5, 3, 256, 1, 1, 1), (103, 1, 25, 2, 64, 0)):
    w = np.arange(1, n_chunks * n_pts + 1, dtype=np.float64); per = np.zeros(n_chunks); tot = ctypes.c_double()
    fmin, fmax, nr = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    assert c.emul_fused_schedule(n_chunks, n_pts, grid, seg, unit, rac, vp(w), vp(per), ctypes.byref(tot), ctypes.byref(fmin),
                                 ctypes.byref(fmax), ctypes.byref(nr)) == 0
    assert tot.value == w.sum()
m = ctypes.CDLL(libs["emul_mixed"])
for n in (1, 2, 6, 96, 100, 127, 1000, 1001, 3000, 6561, 10240):
    a = (np.arange(n) + 1j * np.arange(n)[::-1]).astype(np.complex64); o = np.zeros(n, np.complex64)
    assert m.emul_mixed_fft(vp(a), vp(o), n, m.emul_mixed_threads_per_row(n), None) >= 0
print("sanitized emulation ok")
""" % {"root": os.path.dirname(HERE), "libs": libs}
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    proc = subprocess.run([sys.executable, "-c", driver], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert "sanitized emulation ok" in proc.stdout
    assert "AddressSanitizer" not in proc.stderr and "runtime error" not in proc.stderr, proc.stderr[-3000:]


def _spec_shape(nchan, ntaps, u8=False, fonly=False, xm=False):
    """The library's own cut of fx_spec.h for this shape (fxc_spec_probe compiles it through hiprtc -- no GPU needed -- and
    reports threads per slot, slots and stage order), as the -D options the host emulation is built with."""
    import re
    from effex_amd import _lib
    lib = _lib.load(dev=bool(os.environ.get("FXC_RTC_U")))      # (the knob that forces the frames per step exists in the developer library only)
    buf = ctypes.create_string_buffer(1024)
    rc = lib.fxc_spec_probe(nchan, ntaps, 3 if xm else (2 if fonly else int(u8)), b"gfx950", buf, len(buf))
    if rc != 0:
        return rc, None
    fonly = fonly or xm
    rep = dict(kv.split("=") for kv in buf.value.decode().split())
    stages = rep["stages"]
    flags = ["-DFXM_N=%d" % nchan, "-DFXM_T=%d" % ntaps, "-DFXM_TPR=%s" % rep["tpr"], "-DFXM_SLOTS=%s" % rep["slots"],
             "-DFXM_NST=%d" % len(stages.split(",")), "-DFXM_RADICES=%s" % stages, "-DFXM_U8=%d" % int(u8),
             "-DFXM_U=%s" % rep["frames_per_step"], "-DFXM_FONLY=%d" % int(fonly), "-DFXM_LEAN=%s" % rep["lean"], "-DFXM_ROWS=%s" % rep["rows"],
             "-DFXM_GROUPS=%s" % rep["groups"], "-DFXM_PADS=%s" % rep["pads"], "-DFXM_PLANE0=%s" % rep["plane0"], "-DFXM_TWFULL=%s" % rep["twfull"], "-DFXM_XM=%d" % int(xm)]
    assert re.fullmatch(r"[0-9,]+", stages) and int(rep["code_bytes"]) > 1000
    return 0, (flags, int(rep["tpr"]), int(rep["slots"]))


def test_specialised_kernel_shapes():
    """Which channel counts get a kernel of their own (h_rtc.h::spec_shape): every prime factor a register butterfly, up to
    four taps, the first stage's points within the ring; the others keep the any-shape kernel (FXC_ERR_UNSUPPORTED here)."""
    from effex_amd import _lib
    for nchan, ntaps in ((997, 4), (1000, 5), (2 * 29, 4), (16384, 4), (2 * 1700, 4)):      # (3400 = 8 x 25 x 17: a 17-point butterfly has room on 256 threads, 2048 channels at most)
        assert _spec_shape(nchan, ntaps)[0] == _lib.FXC_ERR_UNSUPPORTED, (nchan, ntaps)
    rc, (flags, tpr, slots) = _spec_shape(1000, 4)
    stages = [int(v) for v in [f for f in flags if f.startswith("-DFXM_RADICES=")][0].split("=")[1].split(",")]
    # 1000 = 4 x 2 x 5 x 5 x 5 in prime factors: composite radices (10 = 2 x 5 in registers) take fewer trips through LDS
    assert rc == 0 and (tpr, slots) == (256, 1) and int(np.prod(stages)) == 1000 and len(stages) < 5 and stages[0] in (4, 5), stages
    rc, (flags, tpr, slots) = _spec_shape(96, 4)
    assert tpr * slots == 256 and tpr in (32, 64)
    rc, (flags, tpr, slots) = _spec_shape(2 * 17, 4)      # prime factors 17 ... 23: the lean build (registers for the butterfly)
    assert rc == 0 and "-DFXM_LEAN=1" in flags and "-DFXM_RADICES=2,17" in flags


@pytest.mark.parametrize("nchan,ntaps,n_pts,wg_splits,u8", [
    (1000, 4, 11, 2, False), (1000, 4, 5, 1, True), (96, 4, 37, 1, False), (12, 4, 150, 2, False), (6, 2, 9, 1, False),
    (7, 1, 5, 1, False), (250, 4, 13, 2, False), (720, 3, 7, 2, False), (1001, 4, 6, 1, False), (1536, 4, 6, 1, False),
    (20, 4, 33, 1, True), (4, 4, 40, 1, False),
    # above 2048 channels: the lean build (taps and first twiddles from tables, two first-stage butterflies a thread at 4000)
    (3000, 4, 6, 2, False), (4000, 4, 5, 1, False), (2560, 3, 5, 1, False), (2400, 4, 7, 1, False),
    (340, 4, 9, 1, False), (2 * 19, 2, 40, 1, False), (460, 4, 5, 2, True), (1700, 4, 5, 1, False)])      # prime factors 17, 19, 23
@pytest.mark.parametrize("frames_per_step", [1, 2])
def test_specialised_kernel_matches_oracle(tmp_path, monkeypatch, nchan, ntaps, n_pts, wg_splits, u8, frames_per_step):
    """fx_spec.h -- the two-antenna F+X kernel compiled per channel count -- run on the host (tests/emul/emul_spec.cpp: one
    thread per GPU thread, a real barrier) with the options the library would hand hiprtc: the sums over each slot's run of
    frames, added up over the slots, are the oracle's sum_i spec0[i] conj(spec1[i]) of the chunk (effex.py:508-521 before the
    mean).  Covers odd and even stage counts (buffer swap), one-stage shapes (no LDS), several slots per workgroup with runs of
    different lengths and empty runs, partial lanes in every stage, three taps, and the byte ingest."""
    monkeypatch.setenv("FXC_RTC_U", str(frames_per_step))      # (developer knob: one or two frames per step where the LDS allows)
    rc, shape = _spec_shape(nchan, ntaps, u8)
    assert rc == 0
    flags, tpr, slots = shape
    if frames_per_step == 2 and "-DFXM_U=2" not in flags:
        pytest.skip("one frame per step for this shape (one stage only)")
    lib_path = str(tmp_path / "libemul_spec.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-pthread"] + flags +
                   ["-o", lib_path, os.path.join(HERE, "emul", "emul_spec.cpp")], check=True)
    lib = ctypes.CDLL(lib_path)
    assert lib.emul_spec_threads() == tpr * slots and lib.emul_spec_slots() == slots
    n_chunks, num_samp = 2, nchan * n_pts + min(3, nchan - 1)
    rng = np.random.default_rng(nchan * 7 + n_pts)
    window = rng.standard_normal(ntaps * nchan) if nchan < 16 else design_window(ntaps, nchan)
    if u8:
        xb = rng.integers(0, 256, size=(n_chunks, 2, num_samp, 2), dtype=np.uint8)
        xb[:, 1, 2:] = xb[:, 0, :-2] // 2 + xb[:, 1, 2:] // 2
        dc = (rng.standard_normal((n_chunks, 2, 2)) * 0.1).astype(np.float32)            # conversion offsets [chunk][antenna] (re, im)
        x = (xb.astype(np.float32) / np.float32(127.5) + dc[:, :, None, :]).view(np.complex64)[..., 0]
        x_in, dc_in = xb, dc
    else:
        x = synth.synth_iq(nchan, n_chunks, 2, num_samp)
        x_in, dc_in = x, None
    tw = np.exp(2j * np.pi * np.arange(nchan) / nchan).astype(np.complex64)
    h32 = np.ascontiguousarray(window, dtype=np.float32)
    E = wg_splits * slots
    out = np.full((E, n_chunks, nchan), np.nan + 0j, dtype=np.complex64)
    lib.emul_spec_run.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_longlong] * 3 + [ctypes.c_int] * 2
    assert lib.emul_spec_run(x_in.ctypes.data, h32.ctypes.data, out.ctypes.data, tw.ctypes.data,
                             dc_in.ctypes.data if u8 else None, num_samp, n_pts, n_chunks, wg_splits, 1) == 0
    assert np.isfinite(out).all()
    got = out.astype(np.complex128).sum(axis=0)
    for c in range(n_chunks):
        s0 = fx_oracle.spectrometer_poly(x[c, 0], ntaps, nchan, window)
        s1 = fx_oracle.spectrometer_poly(x[c, 1], ntaps, nchan, window)
        ref = (s0 * np.conj(s1)).sum(axis=0)
        assert np.abs(got[c] - ref).max() <= 1e-5 * np.abs(ref).max(), (nchan, c)
    # a slot's row is the sum over ITS run of frames: slot e of E takes frames [e n_pts / E, (e + 1) n_pts / E)
    e = E - 1
    lo, hi = e * n_pts // E, (e + 1) * n_pts // E
    s0 = fx_oracle.spectrometer_poly(x[0, 0], ntaps, nchan, window)[lo:hi]
    s1 = fx_oracle.spectrometer_poly(x[0, 1], ntaps, nchan, window)[lo:hi]
    ref = (s0 * np.conj(s1)).sum(axis=0)
    assert np.abs(out[e, 0] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize("nchan,ntaps,n_pts,wg_splits,n_streams,ant", [
    (1000, 4, 9, 2, 3, 1), (96, 4, 21, 1, 6, 3), (250, 2, 7, 1, 1, 1), (7, 1, 5, 1, 4, 2), (720, 3, 6, 1, 5, 5), (12, 4, 70, 2, 2, 1),
    (3000, 4, 5, 1, 3, 1), (3584, 2, 4, 1, 2, 2),
    # above 4096 channels: one stream per workgroup, up to sixteen points a thread
    (5000, 4, 5, 1, 3, 1), (6000, 2, 4, 2, 4, 2), (8000, 4, 3, 1, 1, 1)])
def test_specialised_f_stage_matches_oracle(tmp_path, nchan, ntaps, n_pts, wg_splits, n_streams, ant):
    """fx_spec.h built as the F stage alone (FXM_FONLY: what fxc_channelize and the F pass of 3 and more antennas run off the
    powers of two): a workgroup carries a pair of streams, the last butterfly's outputs are the spectra -- against the oracle's
    _spectrometer_poly (effex.py:530-555), natural bin order, for an odd number of streams (the last pair has one) and for the
    antenna-interleaved layout the X-engines read ([chunk][frame][antenna][nchan])."""
    rc, shape = _spec_shape(nchan, ntaps, fonly=True)
    assert rc == 0
    flags, tpr, slots = shape
    lib_path = str(tmp_path / "libemul_spec_f.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-pthread"] + flags +
                   ["-o", lib_path, os.path.join(HERE, "emul", "emul_spec.cpp")], check=True)
    lib = ctypes.CDLL(lib_path)
    assert lib.emul_spec_fonly() == 1
    num_samp = nchan * n_pts + min(2, nchan - 1)
    rng = np.random.default_rng(nchan + n_streams)
    window = rng.standard_normal(ntaps * nchan) if nchan < 16 else design_window(ntaps, nchan)
    x = synth.synth_iq(31 + nchan, n_streams, 1, num_samp)[:, 0]
    tw = np.exp(2j * np.pi * np.arange(nchan) / nchan).astype(np.complex64)
    h32 = np.ascontiguousarray(window, dtype=np.float32)
    assert n_streams % ant == 0
    out = np.full((n_streams // ant, n_pts, ant, nchan), np.nan + 0j, dtype=np.complex64)
    lib.emul_spec_run.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_longlong] * 3 + [ctypes.c_int] * 2
    assert lib.emul_spec_run(x.ctypes.data, h32.ctypes.data, out.ctypes.data, tw.ctypes.data, None, num_samp, n_pts, n_streams,
                             wg_splits, ant) == 0
    assert np.isfinite(out).all()
    for s_ in range(n_streams):
        ref = fx_oracle.spectrometer_poly(x[s_], ntaps, nchan, window)
        got = out[s_ // ant, :, s_ % ant, :]
        assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max(), (nchan, s_)


def test_specialised_kernels_are_cached_on_disk(tmp_path, monkeypatch):
    """A build of fx_spec.h is kept as a code object under FXC_RTC_CACHE, keyed by sources, options, architecture and compiler: the
    second process (here: the second search, with the in-process cache out of the picture -- fxc_spec_probe has none) loads it
    instead of compiling; a damaged file is ignored and replaced; FXC_RTC_CACHE=0 writes nothing."""
    import time
    from effex_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(1024)
    cache = tmp_path / "rtc"
    monkeypatch.setenv("FXC_RTC_CACHE", str(cache))

    def shape_of(report):          # everything but where the code object came from
        return report.split(b" source=")[0]

    t0 = time.time()
    assert lib.fxc_spec_probe(360, 4, 0, b"gfx950", buf, len(buf)) == 0
    cold, first = time.time() - t0, buf.value
    assert first.endswith(b"source=built")
    files = sorted(cache.glob("*.co"))
    assert files and all(f.read_bytes()[:4] == b"\x7fELF" for f in files)
    t0 = time.time()
    assert lib.fxc_spec_probe(360, 4, 0, b"gfx950", buf, len(buf)) == 0 and shape_of(buf.value) == shape_of(first)
    assert buf.value.endswith(b"source=cache")
    warm = time.time() - t0
    assert warm < 0.5 * cold, (cold, warm)
    for f in files:
        f.write_bytes(b"not a code object")
    assert lib.fxc_spec_probe(360, 4, 0, b"gfx950", buf, len(buf)) == 0 and shape_of(buf.value) == shape_of(first)
    assert files[0].read_bytes()[:4] == b"\x7fELF"
    off = tmp_path / "off"
    monkeypatch.setenv("FXC_RTC_CACHE", "0")
    assert lib.fxc_spec_probe(360, 4, 2, b"gfx950", buf, len(buf)) == 0
    assert not off.exists() and len(sorted(cache.glob("*.co"))) == len(files)


def test_prebuilt_code_objects_answer_before_hiprtc(monkeypatch):
    """The stated list of channel counts of effex_amd/build.py (PREBUILT_CHANNELS) ships as code objects beside the library
    (csrc/rtc_prebuilt/, keyed by source + options + architecture, not by compiler): the shipped library finds them with the
    run-time cache switched off, for every variant the shape has, and says so (fxc_spec_probe's source=, fxc_info.spec_source);
    a channel count outside the list is built by hiprtc."""
    from effex_amd import _lib
    from effex_amd import build as fx_build
    fx_build.prebuild()
    lib = _lib.load()
    monkeypatch.setenv("FXC_RTC_CACHE", "0")
    buf = ctypes.create_string_buffer(1024)
    for nchan in fx_build.PREBUILT_CHANNELS:
        for variant in (0, 1, 2):
            rc = lib.fxc_spec_probe(nchan, 4, variant, b"gfx950", buf, len(buf))
            if rc == _lib.FXC_ERR_UNSUPPORTED:
                assert nchan > 4096 and variant < 2
                continue
            assert rc == 0 and buf.value.endswith(b"source=prebuilt") and b"scratch=0" in buf.value, (nchan, variant, buf.value)
    assert lib.fxc_spec_probe(1080, 4, 0, b"gfx950", buf, len(buf)) == 0 and buf.value.endswith(b"source=built")


@pytest.mark.parametrize("nchan,ntaps,n_pts,wg_splits", [(6000, 4, 5, 2), (4500, 2, 4, 1)])
def test_second_pass_kernel_matches_oracle(tmp_path, nchan, ntaps, n_pts, wg_splits):
    """Two antennas above 4096 channels off the powers of two, two passes (h_launch.h::two_pass_raw_sums) on the host emulation:
    antenna 0 of every chunk pair through the F-only build (streams two chunks apart: Args::stride), then antenna 1 through the
    second-pass build (FXM_XM) whose last butterfly multiplies with antenna 0's spectra -- the sums over the slots' runs are the
    oracle's sum_i spec0[i] conj(spec1[i]) (effex.py:508-521 before the mean)."""
    libs = []
    for tag, xm in (("f", False), ("x", True)):
        rc, shape = _spec_shape(nchan, ntaps, fonly=not xm, xm=xm)
        assert rc == 0
        flags, tpr, slots = shape
        assert "-DFXM_ROWS=1" in flags and slots == 1
        lib_path = str(tmp_path / ("libemul_spec_%s.so" % tag))
        subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-pthread"] + flags +
                       ["-o", lib_path, os.path.join(HERE, "emul", "emul_spec.cpp")], check=True)
        lib = ctypes.CDLL(lib_path)
        lib.emul_spec_run2.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_longlong] * 3 + [ctypes.c_int] * 2 + [ctypes.c_longlong, ctypes.c_void_p]
        assert lib.emul_spec_xm() == int(xm)
        libs.append(lib)
    f_lib, x_lib = libs
    n_chunks, num_samp = 3, nchan * n_pts + 11
    window = design_window(ntaps, nchan)
    x = synth.synth_iq(nchan + 1, n_chunks, 2, num_samp)
    tw = np.exp(2j * np.pi * np.arange(nchan) / nchan).astype(np.complex64)
    h32 = np.ascontiguousarray(window, dtype=np.float32)
    spec0 = np.full((n_chunks, n_pts, nchan), np.nan + 0j, dtype=np.complex64)
    assert f_lib.emul_spec_run2(x.ctypes.data, h32.ctypes.data, spec0.ctypes.data, tw.ctypes.data, None, num_samp, n_pts, n_chunks,
                                wg_splits, 1, 2 * num_samp, None) == 0
    for c in range(n_chunks):
        ref = fx_oracle.spectrometer_poly(x[c, 0], ntaps, nchan, window)
        assert np.abs(spec0[c] - ref).max() <= 1e-5 * np.abs(ref).max()
    out = np.full((wg_splits, n_chunks, nchan), np.nan + 0j, dtype=np.complex64)
    ant1 = x.reshape(-1)[num_samp:]
    assert x_lib.emul_spec_run2(ant1.ctypes.data, h32.ctypes.data, out.ctypes.data, tw.ctypes.data, None, num_samp, n_pts, n_chunks,
                                wg_splits, 1, 2 * num_samp, spec0.ctypes.data) == 0
    assert np.isfinite(out).all()
    got = out.astype(np.complex128).sum(axis=0)
    for c in range(n_chunks):
        s0 = fx_oracle.spectrometer_poly(x[c, 0], ntaps, nchan, window)
        s1 = fx_oracle.spectrometer_poly(x[c, 1], ntaps, nchan, window)
        ref = (s0 * np.conj(s1)).sum(axis=0)
        assert np.abs(got[c] - ref).max() <= 1e-5 * np.abs(ref).max(), (nchan, c)
