"""Build libfxcorr.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libfxcorr.so")
LIB_DEV = os.path.join(CSRC, "libfxcorr_dev.so")      # -DFXC_DEV_KERNELS=1: tests and tools only (effex_amd/_lib.py::load(dev=True))
# one translation unit: fxcorr.hip (the C ABI) includes the kernel (k_*.h), phase (fx_*.h) and host (h_*.h) parts


def sources():
    parts = sorted(f for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".hip"))
    return parts + [os.path.join("..", "..", "include", "fxcorr.h")]



# -fno-slp-vectorize: packed f32 VALU runs at the scalar-f32 rate on gfx950 and the v_pk_* forms cost
# operand-shuffling moves, so SLP packing of the butterflies is a net loss (MI355X_MICROARCH.md).
# -Bsymbolic-functions: calls between the library's own exported functions stay inside it (the developer build is loaded
# beside the shipped one in the test process: the first-loaded copy must not answer for the other)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC", "-Wl,-Bsymbolic-functions"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def up_to_date(lib=LIB):
    if not os.path.isfile(lib):
        return False
    t = os.path.getmtime(lib)
    return all(os.path.getmtime(os.path.join(CSRC, s)) <= t for s in sources())


def build(force=False, verbose=False, dev=False):
    lib = LIB_DEV if dev else LIB
    if not force and up_to_date(lib):
        return lib
    cmd = [hipcc_path()] + FLAGS + (["-DFXC_DEV_KERNELS=1"] if dev else []) + ["-o", lib, "fxcorr.hip"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return lib


# Channel counts whose run-time-compiled kernels (fx_spec.h through hiprtc) are ALSO built here, at build time, into csrc/rtc_prebuilt/ --
# code objects keyed by source + options + architecture (not by compiler) that libfxcorr looks up before it touches hiprtc, so that a plan
# for one of them costs a file read instead of 1 - 14 s of compiling (fxc_info.spec_source == 3).  Every variant that exists for the shape:
# 0 F + X from complex64, 1 F + X from the receivers' bytes, 2 the F stage alone (fxc_channelize; 3+ antennas; above 4096 channels).
PREBUILT_CHANNELS = (1000, 1200, 1250, 1280, 1440, 1500, 1536, 1600, 1800, 1920, 2000, 2400, 2500, 2560, 3000, 3072, 3600, 4000, 5000, 6000)
PREBUILT_TAPS = (4,)
PREBUILT_DIR = os.path.join(CSRC, "rtc_prebuilt")


def prebuild(force=False, verbose=False):
    """Fill csrc/rtc_prebuilt/ (needs no GPU: hiprtc cross-compiles for gfx950).  The developer library does the writing
    (FXC_RTC_PREBUILD_DIR is one of its knobs); the shipped one only reads the directory."""
    import ctypes
    import hashlib
    h = hashlib.sha256()
    for name in ("fx_spec.h", "fx_mixed.h", "fx_math.h", "h_rtc.h", "spec_tuned.h"):
        path = os.path.join(CSRC, name)
        if os.path.isfile(path):
            h.update(open(path, "rb").read())
    h.update(repr((PREBUILT_CHANNELS, PREBUILT_TAPS)).encode())
    stamp, want = os.path.join(PREBUILT_DIR, "stamp.txt"), h.hexdigest()
    if not force and os.path.isfile(stamp) and open(stamp).read().strip() == want:
        return PREBUILT_DIR
    shutil.rmtree(PREBUILT_DIR, ignore_errors=True)
    os.makedirs(PREBUILT_DIR)
    lib = ctypes.CDLL(build(dev=True))
    lib.fxc_spec_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    saved = {k: os.environ.get(k) for k in ("FXC_RTC_PREBUILD_DIR", "FXC_RTC_CACHE")}
    os.environ["FXC_RTC_PREBUILD_DIR"] = PREBUILT_DIR
    os.environ["FXC_RTC_CACHE"] = "0"          # (compile: a cache hit of another compiler's build is not what should ship)
    try:
        for nchan in PREBUILT_CHANNELS:
            for taps in PREBUILT_TAPS:
                for variant in (0, 1, 2) + ((3,) if nchan > 4096 else ()):      # (3: the second pass of two antennas above 4096 channels)
                    report = ctypes.create_string_buffer(1024)
                    rc = lib.fxc_spec_probe(nchan, taps, variant, b"gfx950", report, len(report))
                    if verbose:
                        print("prebuilt", nchan, taps, variant, rc, report.value.decode()[:160])
                    if rc not in (0, -2):      # (-2: the shape has no such kernel, e.g. F + X above 4096 channels)
                        raise RuntimeError("pre-building the kernel for %d channels, variant %d failed (%d)" % (nchan, variant, rc))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    with open(stamp, "w") as fh:
        fh.write(want + "\n")
    return PREBUILT_DIR


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, dev=True))
    if "--prebuilt" in sys.argv:
        print(prebuild(force="--force" in sys.argv, verbose=True))
