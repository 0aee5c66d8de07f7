"""gloo tests of the sharding plumbing on CPU, world sizes 2 and 3.

The device arithmetic is covered by tests/test_gpu_parity.py::test_sharded_integration_equals_single_rank;
here several processes own disjoint chunk ranges and either (a) build the tensor their GPU would export (raw
cross-spectra sums + spectra count — produced by the oracle, which is what the HIP path is checked
against) and reduce it, or (b) drive ``ShardedIntegrator.finalize()`` itself through a plan whose
export / finalize_sums are implemented with the oracle; the result must equal the single-rank integration.
``bench.py --dry-run-dist`` runs bench.py's own multi-rank control flow under ``torch.distributed.run``."""
import subprocess
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NCHAN, NTAPS, NUM_SAMP, N_CHUNKS = 256, 4, 256 * 8, 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _local_sums(x, lo, hi, window):
    import fx_oracle
    sums = np.zeros(NCHAN + 1, dtype=np.complex128)
    for c in range(lo, hi):
        f0 = fx_oracle.spectrometer_poly(x[c, 0], NTAPS, NCHAN, window)
        f1 = fx_oracle.spectrometer_poly(x[c, 1], NTAPS, NCHAN, window)
        sums[:NCHAN] += (f0 * np.conj(f1)).sum(axis=0)
        sums[NCHAN] += f0.shape[0]
    return sums


def _worker(rank, world, port, to_all, queue):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from effex_amd import sharding, synth
    from effex_amd.window import design_window
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        window = design_window(NTAPS, NCHAN)
        lo, hi = sharding.chunk_range(rank, world, N_CHUNKS)
        # each rank generates only its own shard of the synthetic stream
        x_local = synth.synth_iq(1234, hi - lo, 2, NUM_SAMP, first_chunk=lo)
        x_view = {lo + k: x_local[k] for k in range(hi - lo)}
        sums = np.zeros(NCHAN + 1, dtype=np.complex128)
        import fx_oracle
        for c in range(lo, hi):
            f0 = fx_oracle.spectrometer_poly(x_view[c][0], NTAPS, NCHAN, window)
            f1 = fx_oracle.spectrometer_poly(x_view[c][1], NTAPS, NCHAN, window)
            sums[:NCHAN] += (f0 * np.conj(f1)).sum(axis=0)
            sums[NCHAN] += f0.shape[0]
        t = torch.from_numpy(sums)
        sharding.reduce_sums(t, root=0, to_all=to_all)
        queue.put((rank, lo, hi, t.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_chunk_range_partitions():
    from effex_amd import sharding
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 10000):
            ranges = [sharding.chunk_range(r, world, n) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.chunk_range(2, 2, 10)


@pytest.mark.parametrize("to_all", [False, True])
def test_two_rank_reduce_equals_single_rank(to_all):
    import torch.multiprocessing as mp
    from effex_amd import synth
    from effex_amd.window import design_window
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, to_all, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        rank, lo, hi, sums = queue.get(timeout=120)
        results[rank] = (lo, hi, sums)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][:2] == (0, 3) and results[1][:2] == (3, 7)
    x = synth.synth_iq(1234, N_CHUNKS, 2, NUM_SAMP)
    ref = _local_sums(x, 0, N_CHUNKS, design_window(NTAPS, NCHAN))
    np.testing.assert_allclose(results[0][2], ref, rtol=1e-12, atol=1e-18)
    assert results[0][2][NCHAN].real == N_CHUNKS * (NUM_SAMP // NCHAN)
    if to_all:
        np.testing.assert_array_equal(results[1][2], results[0][2])


class OraclePlan(object):
    """Stands in for FxPlan on a machine without a GPU: the same surface ShardedIntegrator uses (new_sums,
    fx_accumulate, acc_export, finalize_sums, acc_reset, sync), with the oracle doing the arithmetic."""
    n_baselines, nchan, _follow = 1, NCHAN, True

    def __init__(self, window, rot):
        self.window, self.rot = window, rot
        self.acc = np.zeros(NCHAN + 1, dtype=np.complex128)
        self.resets = 0

    def new_sums(self):
        import torch
        return torch.zeros(NCHAN + 1, dtype=torch.complex128)

    def fx_accumulate(self, x):
        self.acc += _local_sums(x, 0, len(x), self.window)
        return len(x)

    def acc_export(self, sums):
        import torch
        sums.copy_(torch.from_numpy(self.acc))
        return sums

    def finalize_sums(self, sums, mode="SPECTRUM", bandwidth=1.0):
        s = sums.numpy()
        vis = np.fft.fftshift(s[:NCHAN] / s[NCHAN].real * np.conj(self.rot))       # effex.py:520-521
        return vis[None, :] if mode == "SPECTRUM" else np.array([vis.mean() / bandwidth])

    def acc_reset(self):
        self.acc[:] = 0
        self.resets += 1

    def sync(self):
        pass


def _integrator_worker(rank, world, port, to_all, queue):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import fx_oracle
    from effex_amd import sharding, synth
    from effex_amd.window import design_window
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rot = fx_oracle.rot_table(NCHAN, 2.4e6, 1.4204e9, 1e-6)
        plan = OraclePlan(design_window(NTAPS, NCHAN), rot)
        integ = sharding.ShardedIntegrator(plan, rank, world)
        assert integ.transport == "torch.distributed"
        lo, hi = integ.my_range(N_CHUNKS)
        outs = []
        for _ in range(2):                     # two integrations back to back: the reset in between matters
            integ.accumulate(synth.synth_iq(1234, hi - lo, 2, NUM_SAMP, first_chunk=lo))
            outs.append(integ.finalize("SPECTRUM", 2.4e6, root=world - 1, to_all=to_all))
        queue.put((rank, (lo, hi), outs, plan.resets))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,to_all", [(2, False), (2, True), (3, False), (3, True)])
def test_sharded_integrator_finalize_over_gloo(world, to_all):
    """ShardedIntegrator.finalize(): export -> reduce to a non-zero root (or all-reduce) -> finalize on the root ->
    reset, twice in a row, for world sizes 2 and 3."""
    import torch.multiprocessing as mp
    import fx_oracle
    from effex_amd import synth
    from effex_amd.window import design_window
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_integrator_worker, args=(r, world, port, to_all, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        rank, rng, outs, resets = queue.get(timeout=180)
        results[rank] = (rng, outs, resets)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [results[r][0] for r in range(world)] == [((r * N_CHUNKS) // world, ((r + 1) * N_CHUNKS) // world)
                                                     for r in range(world)]
    x = synth.synth_iq(1234, N_CHUNKS, 2, NUM_SAMP)
    rot = fx_oracle.rot_table(NCHAN, 2.4e6, 1.4204e9, 1e-6)
    ref = fx_oracle.fx_integrate(x, NCHAN, design_window(NTAPS, NCHAN), rot=rot)
    root = world - 1
    for rank in range(world):
        _, outs, resets = results[rank]
        assert resets == 2
        for out in outs:
            if to_all or rank == root:
                np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-20)
            else:
                assert out is None


class OracleRowsPlan(object):
    """Stands in for FxPlan.fx_rows on a machine without a GPU (the oracle does the arithmetic; rows rounded to the
    device's complex64)."""
    n_baselines, nchan = 1, NCHAN

    def __init__(self, window, rot_args):
        self.window, self.rot_args = window, rot_args
        self.calls = []

    def fx_rows(self, x, mode="SPECTRUM", bandwidth=1.0, remove_dc=False, out=None):
        import fx_oracle
        bw, fc, tau = self.rot_args
        self.calls.append(len(x))
        rows = []
        for pair in x:
            a, b = (fx_oracle.remove_dc(pair[0]), fx_oracle.remove_dc(pair[1])) if remove_dc else (pair[0], pair[1])
            rows.append(fx_oracle.pfb_xcorr(a, b, NTAPS, NCHAN, self.window, bw, fc, tau, mode))
        if mode == "SPECTRUM":
            res = np.stack(rows).astype(np.complex64)[:, None, :]
        else:
            res = np.asarray(rows, dtype=np.complex128)[:, None]
        if out is not None:
            out[...] = res
            return out
        return res


def _rows_worker(rank, world, port, path, mode, n_chunks, batch, queue):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from effex_amd import rowsink, sharding, synth
    from effex_amd.window import design_window
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = OracleRowsPlan(design_window(NTAPS, NCHAN), (2.4e6, 1.4204e9, 1e-6))
        rows = sharding.ShardedRows(plan, rank, world, batch=batch)
        header = rowsink.header_line(1, 2.4e6, 1.4204e9, NUM_SAMP, NCHAN, 49.6, mode)
        freqs = rowsink.spectrum_freqs(NCHAN, 2.4e6, 1.4204e9)

        def read_chunks(lo, hi):       # each rank generates only its own chunks of the synthetic stream
            return synth.synth_iq(1234, hi - lo, 2, NUM_SAMP, first_chunk=lo) + np.complex64(0.05 - 0.02j)

        lo, hi = rows.run(path, header, freqs, read_chunks, n_chunks, mode, 2.4e6, remove_dc=True)
        # after run() every rank sees the whole, committed file
        queue.put((rank, lo, hi, plan.calls, len(rowsink.RowFile(path).rows)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,n_chunks,batch", [(2, "SPECTRUM", 7, 2), (3, "SPECTRUM", 11, 3), (3, "CONTINUUM", 5, 1),
                                                       (8, "SPECTRUM", 9, 1)])
def test_sharded_rows_write_one_file_over_gloo(tmp_path, world, mode, n_chunks, batch):
    """SURVEY.md 8e, time-series mode: ranks own disjoint rows of one shared row file, no collective.  The file of
    `world` ranks equals the single-rank file byte for byte, its rows are the oracle's (DC removal included), nothing is
    visible to a reader before the count is published, and tools/rows_to_csv.py's conversion gives the csv the reference's
    writer produces for those rows (effex.py:667-696)."""
    import torch.multiprocessing as mp
    import fx_oracle
    from effex_amd import rowsink, sharding, synth
    from effex_amd.window import design_window
    path = str(tmp_path / "shared.fxb")
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, path, mode, n_chunks, batch, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(queue.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(g[1], g[2]) for g in got] == [sharding.batch_range(r, world, n_chunks, batch) for r in range(world)]
    assert got[0][1] == 0 and got[-1][2] == n_chunks and all(got[i][2] == got[i + 1][1] for i in range(world - 1))
    assert all(g[4] == n_chunks for g in got)
    assert all(c <= batch for g in got for c in g[3])
    # the single-rank file, written by the same class in this process
    window = design_window(NTAPS, NCHAN)
    single = str(tmp_path / "single.fxb")
    header = rowsink.header_line(1, 2.4e6, 1.4204e9, NUM_SAMP, NCHAN, 49.6, mode)
    freqs = rowsink.spectrum_freqs(NCHAN, 2.4e6, 1.4204e9)
    x = synth.synth_iq(1234, n_chunks, 2, NUM_SAMP) + np.complex64(0.05 - 0.02j)
    reads = []                                        # a sequential reader (a file, a socket) must see every batch asked for once, in order

    def read_once(lo, hi):
        reads.append((lo, hi))
        return x[lo:hi]

    sharding.ShardedRows(OracleRowsPlan(window, (2.4e6, 1.4204e9, 1e-6)), batch=batch).run(
        single, header, freqs, read_once, n_chunks, mode, 2.4e6, remove_dc=True)
    assert reads == [(lo, min(n_chunks, lo + batch)) for lo in range(0, n_chunks, batch)]
    assert open(path, "rb").read() == open(single, "rb").read()
    back = rowsink.RowFile(path)
    assert back.rows.shape == (n_chunks, NCHAN if mode == "SPECTRUM" else 1)
    ref = np.stack([np.atleast_1d(fx_oracle.pfb_xcorr(fx_oracle.remove_dc(x[c, 0]), fx_oracle.remove_dc(x[c, 1]), NTAPS, NCHAN,
                                                      window, 2.4e6, 1.4204e9, 1e-6, mode)) for c in range(n_chunks)])
    np.testing.assert_array_equal(np.asarray(back.rows), ref.astype(back.row_dtype))
    out_csv, ref_csv = str(tmp_path / "back.csv"), str(tmp_path / "ref.csv")
    assert rowsink.to_csv(path, out_csv) == n_chunks
    with rowsink.CsvSink(ref_csv, header, freqs if mode == "SPECTRUM" else None) as sink:
        sink.write_rows(ref.astype(back.row_dtype))
    assert open(out_csv, "rb").read() == open(ref_csv, "rb").read()
    loaded = np.loadtxt(out_csv, dtype=np.complex128, delimiter=',', skiprows=2 if mode == "SPECTRUM" else 1)   # post_process.py:201-219
    np.testing.assert_allclose(loaded.reshape(ref.shape), ref, rtol=1e-6)


def test_shared_row_file_shows_nothing_before_the_count_is_published(tmp_path):
    from effex_amd import rowsink
    path = str(tmp_path / "s.fxb")
    rowsink.create_shared(path, "h:1", None, 4, np.complex64, 6)
    with rowsink.RowWindow(path, 2, 5) as win:
        win.rows[...] = 3
        assert len(rowsink.RowFile(path).rows) == 0          # a live reader, a killed writer's leftover
    with pytest.raises(ValueError):
        rowsink.RowWindow(path, 4, 9)                        # outside the file
    rowsink.commit_shared(path, 6)
    rows = np.asarray(rowsink.RowFile(path).rows)
    assert rows.shape == (6, 4) and (rows[2:5] == 3).all() and not rows[:2].any() and not rows[5:].any()


def _dry_run(n_ranks, *extra, env=None, expect_failure=False):
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--steps", "3",
           "--warmup", "1", "--dry-run-dist"] + list(extra)
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    if expect_failure:
        assert proc.returncode != 0 and not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")], proc.stdout[-2000:]
        return proc.stderr
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # rank 0 alone prints the line
    return json.loads(lines[0])


def test_bench_supervisor_falls_back_when_the_communicator_hangs():
    """The watchdog around the real RCCL communicator (bench.py::supervise): every rank is a supervisor that runs the
    measurement in a child; a child stuck between "comm:start" and "comm:done" (here: the test hook, on every rank) is
    killed after --comm-timeout and a fresh child takes the torch.distributed transport on a rendezvous of its own.  The
    run completes, exits 0, and says what happened."""
    env = dict(os.environ, FXC_BENCH_TEST_COMM_HANG="1")
    line = _dry_run(3, "--supervise", "--comm-timeout", "3", "--allow-torch-fallback", env=env)
    assert line["n_gpus"] == 3 and line["frames_total"] == 300 and line["mean_chunk_index"] == 149.5
    assert "no RCCL communicator on rank" in line["fallback"] and "within 3 s" in line["fallback"]
    assert line["rccl"]["ranks_seen"] is None and line["rccl"]["fallback_reason"] == line["fallback"]
    # without the hang: the same supervised launch finishes on the first attempt
    line = _dry_run(2, "--supervise", "--comm-timeout", "30")
    assert line["fallback"] is None and line["frames_total"] == 200


def test_bench_supervisor_decides_for_all_ranks():
    """One rank of three never gets its communicator; the other two are past theirs and wait for it in a collective.  The
    verdict of the supervisor that times out is everybody's (a file all supervisors poll): with --allow-torch-fallback all
    three start the torch.distributed attempt together and the run completes; without it (the default: fxc_reduce or
    nothing) all three kill their children and the launch exits non-zero within seconds, printing no line."""
    import time
    env = dict(os.environ, FXC_BENCH_TEST_COMM_HANG="rank:1")
    line = _dry_run(3, "--supervise", "--comm-timeout", "3", "--allow-torch-fallback", env=env)
    assert line["frames_total"] == 300 and line["mean_chunk_index"] == 149.5
    assert "no RCCL communicator on rank 1 within 3 s" in line["fallback"]
    t0 = time.time()
    err = _dry_run(3, "--supervise", "--comm-timeout", "3", env=env, expect_failure=True)
    assert "no RCCL communicator on rank 1 within 3 s" in err and "fxc_reduce is required" in err
    assert time.time() - t0 < 120


def test_bench_dry_run_dist_two_ranks():
    """bench.py's world > 1 control flow (rank env, first frame per rank, ShardedIntegrator's queued finalize with two
    integrations in flight, barrier, max-over-ranks timing, the per-rank table) under torch.distributed.run
    --nproc-per-node 2 on gloo / CPU tensors: weak scaling, the default."""
    line = _dry_run(2)
    assert line["dry_run"] and line["n_gpus"] == 2 and line["steps"] == 3 and line["frames_per_rank"] == [100, 100]
    assert line["scaling"] == "weak" and line["frames_total"] == 200
    assert line["first_chunk_last_rank"] == 100 and line["transport"] == "torch.distributed"
    assert line["rccl"]["ranks_seen"] is None and "gloo" in line["rccl"]["fallback_reason"]      # says why RCCL saw no ranks
    assert line["mean_chunk_index"] == 99.5      # mean over both ranks' chunk ranges: the reduce reached the root
    assert [r["rank"] for r in line["ranks"]["per_rank"]] == [0, 1]
    assert line["ranks"]["ms_per_step_this_rank"]["max"] <= line["ms_per_step"] * 1.5 + 1.0


def test_bench_dry_run_dist_eight_ranks():
    """The size the driver launches: 8 ranks, weak (the default) and strong with an uneven split."""
    line = _dry_run(8)
    assert line["n_gpus"] == 8 and line["frames_per_rank"] == [100] * 8 and line["frames_total"] == 800
    assert line["first_chunk_last_rank"] == 700 and line["mean_chunk_index"] == 399.5
    assert [r["rank"] for r in line["ranks"]["per_rank"]] == list(range(8))
    line = _dry_run(8, "--scaling", "strong", "--frames", "1001")
    assert line["frames_total"] == 1001 and sum(line["frames_per_rank"]) == 1001
    assert max(line["frames_per_rank"]) - min(line["frames_per_rank"]) == 1 and line["mean_chunk_index"] == 500.0


def test_bench_dry_run_dist_strong_scaling_three_ranks():
    """--scaling strong ('10 000 total'): --frames in all, contiguous uneven ranges."""
    line = _dry_run(3, "--scaling", "strong", "--frames", "100")
    assert line["scaling"] == "strong" and line["frames_total"] == 100 and line["frames_per_rank"] == [33, 33, 34]
    assert line["first_chunk_last_rank"] == 66 and line["mean_chunk_index"] == 49.5


def _comm_worker(rank, world, port, queue):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from effex_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        try:
            comm = sharding.make_comm(0, rank, world)
            queue.put((rank, "comm", None))
            comm.close()
        except Exception as exc:          # no GPU here: every rank must come back with an error, none may hang
            queue.put((rank, "error", str(exc)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_make_comm_fails_on_every_rank_without_a_gpu():
    """sharding.make_comm without GPUs (this container): the unique id is drawn and broadcast, fxc_comm_create refuses on
    every rank (no HIP device) — nobody is left waiting in a collective, which is what bench.py's fall-back to the
    torch.distributed transport relies on."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: communicator creation would succeed or need one GPU per rank")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(queue.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == ["error", "error"], got
    assert all("device" in g[2].lower() or "rccl" in g[2].lower() for g in got), got
