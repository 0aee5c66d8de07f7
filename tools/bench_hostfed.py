#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-fed path (never the bench.py `value`): blocking fxc_fx_rows on host
buffers vs the double-buffered fxc_pipe_* front end.  python tools/bench_hostfed.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    from effex_amd import synth
    from effex_amd.plan import FxPlan, FxPipeline
    num_samp = 2 ** 18
    chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n_batches = max(24, 3072 // chunks)        # >= 1 s of copies: the link needs ~0.1 s of traffic to reach full rate
    x = synth.synth_iq(3, chunks, 2, num_samp)
    batches = [x] * n_batches
    plan = FxPlan(2, 4096, 4, num_samp)
    for _ in range(max(4, 512 // chunks)):
        plan.fx_rows(x)                   # warm up (also ramps the link clocks)
    t0 = time.perf_counter()
    for b in batches:
        plan.fx_rows(b)
    t_block = time.perf_counter() - t0
    with FxPipeline(plan, chunks, depth=depth) as pipe:
        pipe.push(batches[0]); pipe.pop()
        t0 = time.perf_counter()
        pipe.push(batches[0])
        for b in batches[1:]:
            pipe.push(b)
            pipe.pop()
        pipe.pop()
        t_pipe = time.perf_counter() - t0
        # zero-copy producer: the source writes straight into the pinned slot (fill time excluded: a real
        # source - file read, socket, SDR DMA - lands there anyway)
        for _ in range(depth):
            pipe.acquire()[...] = x
            pipe.submit()
        for _ in range(depth):
            pipe.pop()
        t0 = time.perf_counter()
        for _ in range(depth - 1):
            pipe.acquire(); pipe.submit()
        for _ in range(n_batches - (depth - 1)):
            pipe.acquire(); pipe.submit()
            pipe.pop()
        for _ in range(depth - 1):
            pipe.pop()
        t_zero = time.perf_counter() - t0
    # the same front end fed with the receivers' bytes (a quarter of the PCIe traffic)
    u8 = np.random.default_rng(1).integers(0, 256, size=(chunks, 2, num_samp, 2), dtype=np.uint8)
    with FxPipeline(plan, chunks, depth=depth, u8=True) as pipe:
        for _ in range(depth):
            pipe.acquire()[...] = u8
            pipe.submit()
        for _ in range(depth):
            pipe.pop()
        nb8 = 4 * n_batches
        t0 = time.perf_counter()
        for _ in range(depth - 1):
            pipe.acquire(); pipe.submit()
        for _ in range(nb8 - (depth - 1)):
            pipe.acquire(); pipe.submit()
            pipe.pop()
        for _ in range(depth - 1):
            pipe.pop()
        t_u8 = time.perf_counter() - t0
    samples = n_batches * chunks * num_samp
    gb = samples * 16 / 1e9
    print(json.dumps({"depth": depth, "workload": "2 antennas, num_samp 2^18, nchan 4096, %d chunk pairs per batch, host numpy in / rows out" % chunks,
                      "blocking_Msamples_per_s": round(samples / t_block / 1e6, 1), "blocking_GBps_in": round(gb / t_block, 2),
                      "pipelined_Msamples_per_s": round(samples / t_pipe / 1e6, 1), "pipelined_GBps_in": round(gb / t_pipe, 2),
                      "pipelined_zero_copy_Msamples_per_s": round(samples / t_zero / 1e6, 1),
                      "pipelined_zero_copy_GBps_in": round(gb / t_zero, 2),
                      "pipelined_bytes_zero_copy_Msamples_per_s": round(4 * samples / t_u8 / 1e6, 1),
                      "pipelined_bytes_GBps_in": round(4 * samples * 4 / 1e9 / t_u8, 2)}))


if __name__ == "__main__":
    main()
