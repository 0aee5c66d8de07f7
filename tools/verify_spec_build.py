#!/usr/bin/env python3
"""Does a stage list of fx_spec.h build without scratch under hipcc too?  The library builds its kernels per channel count with hiprtc (ROCm's or
PyTorch's, whichever the process has); tools/tune_spec.py measured them under PyTorch's.  hipcc is a third opinion on the same source: a
table entry that spills under it sits at the edge of the register file and is not worth shipping (tools/make_spec_table.py --verify).

    python tools/verify_spec_build.py NCHAN VARIANT U STAGES      -> prints "ok vgprs" or "spill N"
"""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "effex_amd", "csrc")


def check(nchan, variant, u, stages):
    from effex_amd import _lib
    lib = _lib.load(dev=True)
    os.environ["FXC_RTC_RADICES"] = stages
    os.environ["FXC_RTC_U"] = str(u)
    buf = ctypes.create_string_buffer(1024)
    rc = lib.fxc_spec_probe(nchan, 4, variant, b"gfx950", buf, len(buf))
    if rc != 0:
        return "probe %d" % rc
    rep = dict(kv.split("=") for kv in buf.value.decode().split())
    if rep["stages"] != stages or int(rep["frames_per_step"]) != u:
        return "probe chose %s u=%s" % (rep["stages"], rep["frames_per_step"])
    fonly = variant in (2, 3)
    flags = ["-DFXM_N=%d" % nchan, "-DFXM_T=4", "-DFXM_TPR=" + rep["tpr"], "-DFXM_SLOTS=" + rep["slots"], "-DFXM_NST=%d" % len(stages.split(",")),
             "-DFXM_RADICES=" + stages, "-DFXM_U8=%d" % int(variant == 1), "-DFXM_FONLY=%d" % int(fonly), "-DFXM_XM=%d" % int(variant == 3),
             "-DFXM_U=%d" % u, "-DFXM_LEAN=" + rep["lean"], "-DFXM_ROWS=" + rep["rows"], "-DFXM_GROUPS=" + rep["groups"], "-DFXM_PADS=" + rep["pads"],
             "-DFXM_PLANE0=" + rep["plane0"], "-DFXM_TWFULL=" + rep["twfull"], "-DFXM_WAVES=" + rep["waves"]]
    with tempfile.TemporaryDirectory() as tmp:
        src, asm = os.path.join(tmp, "k.hip"), os.path.join(tmp, "k.s")
        open(src, "w").write('#include "fx_spec.h"\n')
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", CSRC] + flags +
                       ["-S", "--cuda-device-only", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    scratch = int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", text).group(1))
    vgprs = int(re.search(r"\.vgpr_count:\s*(\d+)", text).group(1))
    return "ok %d" % vgprs if scratch == 0 else "spill %d" % scratch


if __name__ == "__main__":
    print(check(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]))
