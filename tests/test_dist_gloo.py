"""world_size-2 gloo test of the sharding plumbing on CPU.

The device arithmetic is covered by tests/test_gpu_parity.py::test_sharded_integration_equals_single_rank;
here two processes own disjoint chunk ranges, each builds the tensor its GPU would export (raw
cross-spectra sums + spectra count — produced by the oracle, which is what the HIP path is checked
against), and the reduced result must equal the single-rank integration."""
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
