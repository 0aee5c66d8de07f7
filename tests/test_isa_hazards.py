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


needs_hipcc = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not found")


@pytest.fixture(scope="module")
def asm_listing(tmp_path_factory):
    """Device-only assembly of the library with the build's own flags (one compile for the whole module)."""
    asm = tmp_path_factory.mktemp("asm") / "fxcorr.s"
    flags = [f for f in fx_build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([fx_build.hipcc_path()] + flags + ["-S", "--cuda-device-only", "-o", str(asm), "fxcorr.hip"],
                   cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    return open(asm).read()


@needs_hipcc
def test_no_wide_buffer_store_with_a_scalar_offset(asm_listing):
    wide = [line.strip() for line in asm_listing.split("\n") if re.search(r"\bbuffer_store_(dwordx3|dwordx4)\b", line)]
    # operands: vdata, vaddr, srsrc, soffset [modifiers]
    bad = [line for line in wide if re.match(r"s\d+|m0|s\[", line.split(",")[3].split()[0])]
    assert not bad, "wide buffer stores with an SGPR offset:\n" + "\n".join(bad[:8])


def kernel_resources(asm_text):
    """{mangled kernel name: (VGPRs, AGPR offset, SGPRs, scratch bytes, static LDS bytes)} from the .amdhsa_kernel blocks."""
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm_text, re.S):
        body = m.group(2)

        def field(name, default=0):
            f = re.search(r"\.amdhsa_" + name + r" (\d+)", body)
            return int(f.group(1)) if f else default
        out[m.group(1)] = (field("next_free_vgpr"), field("accum_offset"), field("next_free_sgpr"),
                           field("private_segment_fixed_size"), field("group_segment_fixed_size"))
    return out


@needs_hipcc
def test_no_kernel_spills_to_scratch(asm_listing):
    """Round 3 shipped three kernels with scratch (the uint8 kernel with in-kernel DC removal 56 B per lane, the 16-channel
    F-only kernel 24 B, the 1024-thread 8192-channel kernels 168 / 172 B): every kernel of the library must compile to
    .amdhsa_private_segment_fixed_size 0 -- a spill in a frame loop is a scratch round trip per step, and the ones outside
    cost the wave its scratch set-up.  Also: nothing uses the dynamic-stack or the flat-scratch path."""
    res = kernel_resources(asm_listing)
    assert len(res) >= 80, len(res)
    spilled = {name: r[3] for name, r in res.items() if r[3] != 0}
    assert not spilled, "kernels with scratch: {}".format(spilled)
    assert not re.search(r"\bscratch_(load|store)", asm_listing)
    # the two-waves-per-SIMD kernels stay within the 256 registers that occupancy leaves them
    assert max(r[0] for r in res.values()) <= 256


@needs_hipcc
def test_no_buffer_access_inside_a_readfirstlane_loop(asm_listing):
    """A buffer descriptor the compiler takes for divergent is served by a "waterfall": v_readfirstlane x 4, two 64-bit
    compares, s_and_saveexec, the access, s_cbranch_execnz -- a dozen instructions and a loop around every single load.
    Round 4 found the F-only tiled kernels written that way (520 v_readfirstlane per kernel: the descriptor's size word was
    `2 c + 1 >= n_streams ? 1 : 2` times the stream size, which the compiler tied to a per-lane test and formed with a
    v_cndmask) and, for one commit, the pre-filter pass (a per-lane stream index in the descriptor base: the 32-tap pass
    1.7 -> 2.15 ms).  Every descriptor of this library is wave-uniform by construction: no kernel may contain the pattern."""
    bad = {}
    kernels = list(re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)s_endpgm", asm_listing, re.S | re.M))
    assert len(kernels) >= 80, len(kernels)
    for m in kernels:
        lines = [ln.strip() for ln in m.group(2).split("\n")]
        hits = 0
        for i, ln in enumerate(lines):
            if ln.startswith(("buffer_load", "buffer_store")):
                before, after = lines[max(0, i - 12):i], lines[i + 1:i + 4]
                if any(b.startswith("v_readfirstlane_b32") for b in before) and any(a.startswith("s_cbranch_execnz") for a in after):
                    hits += 1
        if hits:
            bad[m.group(1)] = hits
    assert not bad, "buffer accesses inside readfirstlane loops: {}".format(bad)


@needs_hipcc
def test_no_flat_memory_instructions(asm_listing):
    """Every pointer this library dereferences on the device is global or LDS, and the compiler must know which: a `flat_`
    access counts against BOTH the vector-memory and the LDS counters, so a wait for LDS operands becomes a wait for HBM
    (round 4: a pointer passed through an asm barrier, or selected from an array of pointers, lost its address space --
    23 flat loads per frame in the 8192-channel kernel, 35 in the first matrix-core X-engine)."""
    flat = [ln.strip() for ln in asm_listing.split("\n") if re.match(r"\s+flat_(load|store|atomic)", ln)]
    assert not flat, flat[:8]


@needs_hipcc
@pytest.mark.parametrize("nchan,variant", [(1000, 0), (96, 0), (720, 0), (1000, 1), (1000, 2), (1536, 0), (3000, 0), (4000, 0), (4000, 2)])
def test_specialised_kernels_keep_their_registers(tmp_path, nchan, variant):
    """fx_spec.h as the library builds it for one channel count (the options fxc_spec_probe reports; variant 0 / 1 / 2 = complex64,
    bytes, F only), compiled here with hipcc and -- by the probe itself -- with the hiprtc the process has: no scratch under
    either compiler (the ring, the taps and the twiddles are registers or -- the lean build above 2048 channels -- loads from tables, never the stack), at most 256 registers where the build means to
    keep two workgroups of 256 threads on a CU, no flat or scratch memory instruction, the samples through buffer loads, and one
    barrier per LDS round trip (S - 1 per step)."""
    import ctypes
    from effex_amd import _lib
    buf = ctypes.create_string_buffer(1024)
    assert _lib.load().fxc_spec_probe(nchan, 4, variant, b"gfx950", buf, len(buf)) == 0
    rep = dict(kv.split("=") for kv in buf.value.decode().split())
    assert int(rep["scratch"]) == 0 and 0 < int(rep["vgprs"]) <= 512 // max(1, (int(rep["tpr"]) * int(rep["slots"]) // 64 + 3) // 4 * int(rep["resident"]))
    stages = rep["stages"].split(",")
    flags = ["-DFXM_N=%d" % nchan, "-DFXM_T=4", "-DFXM_TPR=" + rep["tpr"], "-DFXM_SLOTS=" + rep["slots"], "-DFXM_NST=%d" % len(stages),
             "-DFXM_RADICES=" + rep["stages"], "-DFXM_U8=%d" % int(variant == 1), "-DFXM_FONLY=%d" % int(variant == 2), "-DFXM_U=" + rep["frames_per_step"],
             "-DFXM_LEAN=" + rep["lean"], "-DFXM_ROWS=" + rep["rows"], "-DFXM_GROUPS=" + rep["groups"], "-DFXM_PADS=" + rep["pads"],
             "-DFXM_PLANE0=" + rep["plane0"], "-DFXM_TWFULL=" + rep["twfull"], "-DFXM_WAVES=" + rep["waves"]]
    src = tmp_path / "spec.hip"
    src.write_text('#include "fx_spec.h"\n')
    asm = tmp_path / "spec.s"
    subprocess.run([fx_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", CSRC] + flags +
                   ["-S", "--cuda-device-only", "-o", str(asm), str(src)], check=True, stderr=subprocess.DEVNULL)
    text = asm.read_text()
    assert re.search(r"\.private_segment_fixed_size:\s*0\b", text) and re.search(r"\.vgpr_spill_count:\s*0\b", text)
    body = [ln.split(";")[0].strip() for ln in text.split("\n")]
    ops = [ln.split()[0] for ln in body if ln and not ln.startswith((".", ";")) and not ln.endswith(":")]
    assert not [o for o in ops if o.startswith(("flat_", "scratch_"))]
    assert any(o.startswith("buffer_load_") for o in ops)
    if int(rep["tpr"]) > 64 and len(stages) >= 2:
        unrolled = (4 + int(rep["frames_per_step"]) - 1)
        unrolled //= __import__("math").gcd(unrolled, int(rep["frames_per_step"]))
        # (+ 2 once per launch where the last stage's items split a step's frames over threads: their sums meet in LDS at the end)
        rows_per_step = int(rep["rows"]) * int(rep["frames_per_step"])
        last_group = int(rep["groups"].split(",")[-1]) or rows_per_step
        tail = 2 if (variant != 2 and last_group < rows_per_step) else 0
        assert ops.count("s_barrier") == unrolled * (len(stages) - 1) + tail, (ops.count("s_barrier"), unrolled, stages, tail)
