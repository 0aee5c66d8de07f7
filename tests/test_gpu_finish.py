"""The end of an integration on the GPU: the fold of the kernels' raw rows into the accumulator, export, finalize and
reset together (k_finish.h::fold_partial_kernel + fold_finish_kernel), the asynchronous finalize (include/fxcorr.h fxc_finalize_async /
fxc_finalize_wait) and BASELINE configs[1] at its own size.

Reference semantics: mean over all spectra of f0 * conj(f1 * rot), fft-shifted (effex/effex.py:516-521); CONTINUUM:
mean over the bins / bandwidth (:523-524).  Tolerance as in test_gpu_parity.py: 1e-5 of max|vis| against the float64
oracle; results that must be the same arithmetic in the same order are compared bit for bit.
"""
import numpy as np
import pytest

import fx_oracle
import golden_inputs as gi
from effex_amd import _lib, synth
from effex_amd.window import design_window

pytestmark = pytest.mark.gpu

from tolerances import TOL_VIS


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def plan_mod(torch):
    from effex_amd import plan
    return plan


def rel_err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


SHAPES = [  # n_ant, nchan, ntaps, num_samp, n_chunks, path
    (2, 4096, 4, 4096 * 6, 7, "fused"), (2, 4096, 4, 4096 * 3, 300, "fused"), (2, 2048, 4, 2048 * 9 + 5, 11, "tiled"),
    (2, 2048, 32, 2048 * 40, 3, "tiled"), (2, 8192, 4, 8192 * 5, 4, "tiled"), (8, 4096, 4, 4096 * 4, 3, "fused"),
    (2, 1, 4, 5000, 6, "stream"), (3, 8, 4, 8 * 20, 5, "generic"), (3, 64, 4, 64 * 20, 5, "tiled"), (2, 256, 4, 256 * 20, 5, "tiled")]


@pytest.mark.parametrize("n_ant,nchan,ntaps,num_samp,n_chunks,path", SHAPES)
def test_async_finalize_equals_blocking_finalize(plan_mod, torch, n_ant, nchan, ntaps, num_samp, n_chunks, path):
    """Every path, both modes: the queued finalize gives the bits of the blocking one, `reset=False` keeps integrating,
    and the integration equals the float64 mean of the per-chunk rows."""
    x = torch.from_numpy(synth.synth_iq(11, n_chunks, n_ant, num_samp)).cuda()
    window = np.array([0.4, 0.3, 0.2, 0.1]) if nchan == 1 else None
    with plan_mod.FxPlan(n_ant, nchan, ntaps, num_samp, window=window) as p:
        assert p.path == path
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 1e-6)
        rows = p.fx_rows(x, "SPECTRUM").cpu().numpy().astype(np.complex128)
        for mode in ("SPECTRUM", "CONTINUUM"):
            p.fx_accumulate(x)
            blocking = p.finalize(mode, gi.BANDWIDTH, reset=False)
            p.finalize_async(mode, gi.BANDWIDTH, reset=False)
            assert p.finalize_pending == 1
            np.testing.assert_array_equal(p.finalize_wait(), blocking)
            p.fx_accumulate(x[: n_chunks // 2 + 1])             # integration goes on: 1.5 x the chunks now
            p.finalize_async(mode, gi.BANDWIDTH, reset=True)
            longer = p.finalize_wait()
            both = np.concatenate([rows, rows[: n_chunks // 2 + 1]]).mean(axis=0)
            want_b = rows.mean(axis=0) if mode == "SPECTRUM" else rows.mean(axis=0).mean(axis=-1) / gi.BANDWIDTH
            want_l = both if mode == "SPECTRUM" else both.mean(axis=-1) / gi.BANDWIDTH
            assert rel_err(blocking, want_b) < 2e-6, mode
            assert rel_err(longer, want_l) < 2e-6, mode
            with pytest.raises(_lib.FxcError):                   # reset: nothing accumulated any more
                p.finalize(mode, gi.BANDWIDTH)


def test_two_integrations_in_flight(plan_mod, torch):
    """The next integration is queued before the host collects the previous one; results come back in order; a third
    outstanding result, and a blocking finalize while any is outstanding, are refused."""
    num_samp = 4096 * 5
    xa = torch.from_numpy(synth.synth_iq(21, 260, 2, num_samp)).cuda()
    xb = torch.from_numpy(synth.synth_iq(22, 9, 2, num_samp)).cuda()
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        p.fx_accumulate(xa)
        ref_a = p.finalize("SPECTRUM")
        p.fx_accumulate(xb)
        ref_b = p.finalize("SPECTRUM")
        for _ in range(3):
            p.fx_accumulate(xa)
            p.finalize_async("SPECTRUM")
            p.fx_accumulate(xb)
            p.finalize_async("SPECTRUM")
            assert p.finalize_pending == 2
            p.fx_accumulate(xa)
            with pytest.raises(_lib.FxcError):
                p.finalize_async("SPECTRUM")
            with pytest.raises(_lib.FxcError):
                p.finalize("SPECTRUM")
            np.testing.assert_array_equal(p.finalize_wait(), ref_a)
            np.testing.assert_array_equal(p.finalize_wait(), ref_b)
            np.testing.assert_array_equal(p.finalize("SPECTRUM"), ref_a)     # the third integration, queued above
        with pytest.raises(_lib.FxcError):
            p.finalize_wait()


@pytest.mark.parametrize("n_ant,nchan,mode", [(2, 4096, "SPECTRUM"), (2, 4096, "CONTINUUM"), (8, 4096, "SPECTRUM"), (12, 1024, "SPECTRUM")])
def test_finalize_delivered_into_the_callers_buffer(plan_mod, torch, n_ant, nchan, mode):
    """fxc_finalize_async_to: the destination is named when the finalize is queued and the device delivers into it -- the
    finishing kernel itself for small results in fxc_host_alloc memory, the side-stream copy for large ones (28 baselines and
    more) -- so the wait copies nothing.  Same bits as the blocking finalize, for pinned and for ordinary arrays, with two
    results in flight; the wait refuses another buffer."""
    import ctypes
    num_samp = nchan * 5
    xa = torch.from_numpy(synth.synth_iq(21, 6, n_ant, num_samp, delays=np.arange(n_ant) % 5)).cuda()
    xb = torch.from_numpy(synth.synth_iq(22, 3, n_ant, num_samp, delays=np.arange(n_ant) % 5)).cuda()
    with plan_mod.FxPlan(n_ant, nchan, 4, num_samp) as p:
        p.fx_accumulate(xa)
        ref_a = p.finalize(mode, gi.BANDWIDTH)
        p.fx_accumulate(xb)
        ref_b = p.finalize(mode, gi.BANDWIDTH)
        pinned = plan_mod.pinned_empty(ref_a.shape, np.complex128)
        plain = np.empty(ref_a.shape, dtype=np.complex128)
        pinned[...] = 0
        p.fx_accumulate(xa)
        p.finalize_async(mode, gi.BANDWIDTH, out=pinned)
        p.fx_accumulate(xb)
        p.finalize_async(mode, gi.BANDWIDTH, out=plain)
        assert p.finalize_pending == 2
        other = np.empty(ref_a.shape, dtype=np.complex128)
        assert p._lib.fxc_finalize_wait(p._h, ctypes.c_void_p(other.ctypes.data)) == _lib.FXC_ERR_ARG   # not the buffer it was queued with
        got_a = p.finalize_wait()
        assert got_a is pinned
        np.testing.assert_array_equal(got_a, ref_a)
        got_b = p.finalize_wait()
        assert got_b is plain
        np.testing.assert_array_equal(got_b, ref_b)
        with pytest.raises(ValueError):
            p.finalize_async(mode, gi.BANDWIDTH, out=np.empty(3, dtype=np.complex128))


def test_fold_is_bit_reproducible(plan_mod, torch):
    """The fold sums rows, phases and partials in a fixed order:
    repeated integrations of the same frames are identical bit for bit (700 chunk pairs: every split in use)."""
    num_samp = 4096 * 2
    x = torch.from_numpy(synth.synth_iq(31, 700, 2, num_samp)).cuda()
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        outs = []
        for _ in range(6):
            p.fx_accumulate(x)
            outs.append(p.finalize("SPECTRUM"))
        for o in outs[1:]:
            np.testing.assert_array_equal(o, outs[0])
        # a pass folded right away (export in between) and one folded together with the finalize: the same sums
        p.fx_accumulate(x)
        p.acc_export(p.new_sums())
        np.testing.assert_array_equal(p.finalize("SPECTRUM"), outs[0])


def test_finalize_sums_needs_reduced_sums(plan_mod, torch):
    x = torch.from_numpy(synth.synth_iq(41, 3, 2, 4096 * 4)).cuda()
    with plan_mod.FxPlan(2, 4096, 4, 4096 * 4) as p:
        p.fx_accumulate(x)
        with pytest.raises(_lib.FxcError):
            p.finalize_sums(None, "SPECTRUM")                    # nothing reduced into the plan yet
        p.reduce(None, 0)
        ref = p.finalize_sums(None, "SPECTRUM")
        np.testing.assert_array_equal(p.finalize("SPECTRUM"), ref)


def test_one_communicator_serves_several_plans(plan_mod, torch):
    """fxc_reduce twice back to back, ncclAllReduce (root < 0) on one plan and ncclReduce to rank 0 on another, both on
    the same communicator (a world of one here: what must hold is that the collectives run on each plan's stream and
    leave the sums unchanged); a root no rank has is refused before anything is queued."""
    num_samp = 4096 * 6
    xa = torch.from_numpy(synth.synth_iq(51, 5, 2, num_samp)).cuda()
    xb = torch.from_numpy(synth.synth_iq(52, 7, 2, num_samp)).cuda()
    uid = plan_mod.RcclComm.unique_id()
    with plan_mod.RcclComm(0, 0, 1, uid) as comm, plan_mod.FxPlan(2, 4096, 4, num_samp) as pa, \
            plan_mod.FxPlan(2, 4096, 4, num_samp, stream="owned") as pb:
        pa.fx_accumulate(xa)
        pb.fx_accumulate(xb)
        ref_a = pa.finalize("SPECTRUM", reset=False)
        ref_b = pb.finalize("SPECTRUM", reset=False)
        pa.reduce(comm, None)
        pb.reduce(comm, 0)
        pa.reduce(comm, 0)
        pb.reduce(comm, None)
        np.testing.assert_array_equal(pb.finalize_sums(None, "SPECTRUM"), ref_b)
        np.testing.assert_array_equal(pa.finalize_sums(None, "SPECTRUM"), ref_a)
        with pytest.raises(ValueError):
            pa.reduce(comm, 1)
        np.testing.assert_array_equal(pa.finalize("SPECTRUM"), ref_a)     # the refused call queued nothing
        # the communicator's own account of itself (fxc_comm_info / fxc_comm_probe): asked of the live ncclComm_t
        info = comm.info()
        assert info["ranks_seen"] == 1 and info["rank_seen"] == 0 and info["device_seen"] == 0, info
        assert (info["world_given"], info["rank_given"], info["device_given"]) == (1, 0, 0)
        assert info["reduces"] == 4 and info["async_error"] == 0 and info["rccl_version"] >= 20000, info
        assert comm.probe() == 1
        version, path = plan_mod.RcclComm.library()
        assert version == info["rccl_version"] and "rccl" in path


def test_headline_config_at_full_size(plan_mod, torch):
    """BASELINE.json configs[1] at its own size: 10 000 distinct integration frames of 2 x 262 144 samples resident in
    HBM (41.9 GB), one fxc_fx_accumulate + finalize -- what bench.py times.  The integration must equal the float64 mean
    of the 10 000 per-frame rows, and sampled frames the oracle (effex.py:490-527), to 1e-5 of max|vis|."""
    frames, num_samp, seed = 10000, 262144, 1234
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < frames * 2 * num_samp * 8 * 1.05:
        pytest.skip("needs 44 GB of free device memory")
    x = torch.empty((frames, 2, num_samp), dtype=torch.complex64, device="cuda")
    plan_mod.synth_fill(x, seed)
    window = design_window(4, 4096)
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        assert p.path == "fused"
        p.set_delay(gi.BANDWIDTH, gi.FREQUENCY, 0.0)
        p.fx_accumulate(x)
        integ = p.finalize("SPECTRUM")[0]
        rows_sum = torch.zeros(4096, dtype=torch.complex128, device="cuda")
        for lo in range(0, frames, 2048):
            rows_sum += p.fx_rows(x[lo:lo + 2048], "SPECTRUM")[:, 0].to(torch.complex128).sum(dim=0)
        mean_rows = (rows_sum / frames).cpu().numpy()
        assert rel_err(integ, mean_rows) < TOL_VIS
        for f in (0, 7777):
            xf = synth.synth_iq(seed, 1, 2, num_samp, first_chunk=f)[0]
            np.testing.assert_array_equal(xf, x[f].cpu().numpy())
            ref = fx_oracle.pfb_xcorr(xf[0], xf[1], 4, 4096, window, gi.BANDWIDTH, gi.FREQUENCY, 0.0, "SPECTRUM")
            got = p.fx_rows(x[f:f + 1], "SPECTRUM")[0, 0].cpu().numpy()
            assert rel_err(got, ref) < TOL_VIS, f
    del x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("frames,extra,n_chunks", [(2, 0, 700), (3, 8 * 37, 600), (1, 4096 - 8, 1030)])
def test_dc_removal_inside_the_fused_kernel(plan_mod, torch, frames, extra, n_chunks):
    """fxc_fx_*_u8 with remove_dc on launches of at least two rounds of chunks per workgroup: the fused kernel sums the
    bytes of a workgroup's next chunk itself (k_fused4096.h::U8State) and the pre-pass covers only the first round and
    the tail.  Exact integer sums and the pre-pass's float64 formula, so the rows of the whole chunks must equal -- bit
    for bit -- the rows of calls of one round each, whose chunks all get their offsets from the pre-pass (the tail chunks,
    cut into frame ranges differently by the two launches, agree to rounding); sampled rows against the oracle chain (pyrtlsdr
    conversion behind effex.py:652, DC removal effex.py:394-395, then effex.py:490-527).  Sizes that are not whole frames
    keep the pre-pass for every chunk."""
    num_samp = 4096 * frames + extra
    rng = np.random.default_rng(77 + frames)
    u8 = rng.integers(0, 256, size=(n_chunks, 2, num_samp, 2), dtype=np.uint8)
    u8[:, 0, :, 0] = np.clip(u8[:, 0, :, 0].astype(int) // 2 + (np.arange(n_chunks) % 97)[:, None], 0, 255)   # DC differs chunk to chunk
    u8[:, 1, 3:, :] = (u8[:, 0, :-3, :] // 2 + u8[:, 1, 3:, :] // 2)
    ud = torch.from_numpy(u8).cuda()
    window = design_window(4, 4096)
    with plan_mod.FxPlan(2, 4096, 4, num_samp) as p:
        assert p.path == "fused"
        rows = p.fx_rows_u8(ud, "SPECTRUM", remove_dc=True).cpu().numpy()
        g = p.info["cu_count"]                   # calls of exactly one round of whole chunks: every offset from the pre-pass
        small = np.concatenate([p.fx_rows_u8(ud[lo:lo + g], "SPECTRUM", remove_dc=True).cpu().numpy()
                                for lo in range(0, n_chunks, g)])
        n_full = n_chunks // g * g               # whole chunks, dealt round-robin; the rest is the tail
        np.testing.assert_array_equal(rows[:n_full], small[:n_full])
        assert rel_err(rows[n_full:], small[n_full:]) < 1e-6
        for c in (0, 255, 256, 300, n_chunks - 1):
            z = fx_oracle.u8_to_complex(u8[c:c + 1])[0]
            ref = fx_oracle.pfb_xcorr(fx_oracle.remove_dc(z[0]), fx_oracle.remove_dc(z[1]), 4, 4096, window, gi.BANDWIDTH,
                                      gi.FREQUENCY, 0.0, "SPECTRUM")
            assert rel_err(rows[c, 0], ref) < TOL_VIS, c
        p.fx_accumulate_u8(ud, remove_dc=True)
        integ = p.finalize("SPECTRUM")
        assert rel_err(integ[0], rows[:, 0].astype(np.complex128).mean(axis=0)) < 2e-6


def test_integration_in_many_workspace_passes():
    """A call over more chunks than the workspace holds runs in passes: all but the last fold their raw rows right away, the
    last one's fold waits for whatever comes next (h_run.h::fold_or_defer).  With the workspace bound forced down to 8 MiB
    (FXC_WS_MB, read once per process: a child process) the integrations of the 2-antenna, 8-antenna, tiled and generic
    routes must equal the float64 mean of their per-chunk rows, as they do in one pass."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from effex_amd import synth
from effex_amd.plan import FxPlan
for n_ant, nchan, num_samp, n_chunks, path in ((2, 4096, 4096 * 8, 700, "fused"), (8, 4096, 4096 * 4, 40, "fused"),
                                               (2, 1024, 1024 * 16, 300, "tiled"), (3, 8, 8 * 40, 200, "generic"),
                                               (3, 64, 64 * 40, 200, "tiled"), (2, 128, 128 * 40, 200, "tiled"),
                                               (2, 8192, 8192 * 8 + 5, 100, "tiled"),      # (two passes per batch: antenna 0's spectra are the bound)
                                               (2, 6000, 6000 * 6 + 11, 120, "generic")):   # (two passes of the kernels built for 6000 channels: likewise)
    x = torch.from_numpy(synth.synth_iq(7, n_chunks, n_ant, num_samp)).cuda()
    with FxPlan(n_ant, nchan, 4, num_samp) as p:
        assert p.path == path, (p.path, path)
        rows = p.fx_rows(x, "SPECTRUM").cpu().numpy().astype(np.complex128)
        p.fx_accumulate(x[: n_chunks // 2])
        p.fx_accumulate(x[n_chunks // 2:])
        p.finalize_async("SPECTRUM")
        integ = p.finalize_wait()
        ws = p.info["workspace_bytes"]
    err = np.abs(integ - rows.mean(axis=0)).max() / np.abs(rows.mean(axis=0)).max()
    print(path, n_ant, "workspace", ws, "err", err)
    assert ws <= (80 << 20), ws            # the bound held (plus the fold's partials, up to 58 MB for 28 baselines)
    assert err < 2e-6, err
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FXC_WS_MB="8")
    proc = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and proc.stdout.strip().endswith("ok"), proc.stdout[-2000:] + proc.stderr[-2000:]


def test_reduce_between_two_gpus():
    """fxc_comm_create / fxc_reduce with a world of two, one process per GPU (what bench.py's RCCL preflight runs): needs
    two GPUs, skipped on the one-GPU boxes this repository is built on."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from effex_amd.plan import RcclComm
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uid = RcclComm.unique_id().hex()
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--rccl-child", uid, str(r), "2", str(r)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    for r, out in enumerate(outs):          # each rank's communicator saw two ranks, and RCCL added up two ones
        seen = json.loads(out[out.index("ok {") + 3:].splitlines()[0])
        assert seen["ranks_seen"] == 2 and seen["rank"] == r and seen["ranks_summed"] == 2 and seen["reduces"] == 3, seen
