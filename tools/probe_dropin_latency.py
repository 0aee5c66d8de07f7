#!/usr/bin/env python3
"""Developer probe: what one reference-sized host-buffer call (4 MiB of complex64 in, one 32 KiB row out) is made of on
this box -- the pinned copy alone (one DMA, or the two streams on two copy streams), the kernels alone on resident samples,
and the host's wait for an idle stream."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def wall(fn, n=300, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return round((time.perf_counter() - t0) / n * 1e6, 1)


def main():
    from effex_amd import synth
    from effex_amd.plan import FxPlan
    n = 2 ** 18
    x = synth.synth_iq(5, 1, 2, n)
    h = torch.from_numpy(x).pin_memory()
    d = torch.empty_like(h, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    for mib in (1, 2, 4, 8, 16):
        hh = torch.empty(mib << 20, dtype=torch.uint8).pin_memory()
        dd = torch.empty_like(hh, device="cuda")
        us = wall(lambda: (dd.copy_(hh, non_blocking=True), torch.cuda.current_stream().synchronize()))
        res["h2d_pinned_%dMiB_us" % mib] = us
        res["h2d_pinned_%dMiB_GBps" % mib] = round((mib << 20) / us / 1e3, 1)

    def two_streams():
        with torch.cuda.stream(s1):
            d[0, 0].copy_(h[0, 0], non_blocking=True)
        with torch.cuda.stream(s2):
            d[0, 1].copy_(h[0, 1], non_blocking=True)
        s1.synchronize()
        s2.synchronize()
    res["h2d_4MiB_as_2x2MiB_on_two_streams_us"] = wall(two_streams)
    res["sync_of_an_idle_stream_us"] = wall(lambda: torch.cuda.current_stream().synchronize())
    ho = torch.empty((1, 1, 4096), dtype=torch.complex64).pin_memory()
    with FxPlan(2, 4096, 4, n) as plan:
        res["fx_rows_resident_plus_sync_us"] = wall(lambda: (plan.fx_rows(d), torch.cuda.current_stream().synchronize()))
        res["fx_rows_resident_dc_plus_sync_us"] = wall(lambda: (plan.fx_rows(d, remove_dc=True), torch.cuda.current_stream().synchronize()))
        res["fx_rows_resident_rows_to_host_us"] = wall(lambda: (ho.copy_(plan.fx_rows(d), non_blocking=True), torch.cuda.current_stream().synchronize()))
        hx = np.ascontiguousarray(x)
        res["fx_rows_pageable_host_call_us"] = wall(lambda: plan.fx_rows(hx))
    with FxPlan(2, 4096, 4, n, stream="owned") as plan:
        from effex_amd.plan import pinned_empty
        pin = pinned_empty(x.shape, np.complex64)
        pin[...] = x
        out = pinned_empty((1, 1, 4096), np.complex64)
        res["fx_rows_pinned_host_call_owned_stream_us"] = wall(lambda: plan.fx_rows(pin, out=out))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
