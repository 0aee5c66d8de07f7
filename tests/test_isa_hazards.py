"""Compiled-code check (no GPU): no wide buffer store with a scalar offset in the library.

A buffer store of more than 8 bytes reads its data registers over several cycles, and a VALU instruction that overwrites
them needs wait states in between.  LLVM's hazard recognizer inserts the `s_nop` -- except when the store carries an SGPR
offset ("this hazard only exists if the instruction is not using a register in the soffset field").  Round 2's pre-filter
kernel with 16-byte accesses compiled to `buffer_store_dwordx4 v[0:3], v40, s[8:11], s0 offen` directly followed by
`v_mov_b64 v[0:1], ...` and stored wrong frames (7 and 13 of every stream of workgroups >= 256), deterministically; with the
offset in the VGPR the compiler pads the pair and every shape is right (k_prepass.h, profiles/r02/round2_experiments.md).
Round 3's stand-alone reproducer (tools/ubench/store_hazard.hip, profiles/r03/gfx950_store_hazard.md) shows the hazard
itself on MI355X -- the un-padded pair WITHOUT a scalar offset corrupts a quarter of the records -- but the pair WITH a
scalar offset stores correctly there, as LLVM assumes: the root cause of round 2's failure is therefore not pinned down, and
the defensive rule stays: wide buffer stores keep their offset in the VGPR."""
import os
import re
import shutil
import subprocess

import pytest

from effex_amd import build as fx_build

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "effex_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")
def test_no_wide_buffer_store_with_a_scalar_offset(tmp_path):
    asm = tmp_path / "fxcorr.s"
    flags = [f for f in fx_build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([fx_build.hipcc_path()] + flags + ["-S", "--cuda-device-only", "-o", str(asm), "fxcorr.hip"],
                   cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    wide = [line.strip() for line in open(asm) if re.search(r"\bbuffer_store_(dwordx3|dwordx4)\b", line)]
    # operands: vdata, vaddr, srsrc, soffset [modifiers]
    bad = [line for line in wide if re.match(r"s\d+|m0|s\[", line.split(",")[3].split()[0])]
    assert not bad, "wide buffer stores with an SGPR offset:\n" + "\n".join(bad[:8])
