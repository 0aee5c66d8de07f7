#!/usr/bin/env python3
"""Developer aid: build tests/emul/emul_fused.cpp with the given -D flags and compare with the oracle."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fx_oracle
from effex_amd import synth, window

def main():
    flags = sys.argv[1:]
    so = os.path.join(tempfile.mkdtemp(), "libemul.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC"] + flags + ["-o", so, os.path.join(ROOT, "tests", "emul", "emul_fused.cpp")], check=True)
    lib = ctypes.CDLL(so)
    S = 4096 * 9 + 5
    x = synth.synth_iq(1234, 1, 2, S)[0]
    w = window.design_window(4, 4096)
    out = np.zeros(4096, np.complex128)
    rc = lib.emul_fused4096(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(S), w.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    f0 = fx_oracle.spectrometer_poly(x[0], 4, 4096, w); f1 = fx_oracle.spectrometer_poly(x[1], 4, 4096, w)
    ref = (f0 * np.conj(f1)).sum(axis=0)
    err = abs(out - ref).max() / abs(ref).max()
    print(flags, "rc", rc, "max rel err %.3g" % err)
    sys.exit(0 if (rc == 0 and err < 1e-6) else 1)

if __name__ == "__main__":
    main()
