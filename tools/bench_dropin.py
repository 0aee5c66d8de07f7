#!/usr/bin/env python3
"""Developer aid: wall time of the drop-in Correlator._run_task() per chunk pair (host complex128 buffers in, host
row out — exactly the reference's call, effex.py:490-527), SPECTRUM and CONTINUUM, and where that time goes: the
host's complex128 -> complex64 pass over the two streams, then one fxc_fx_rows call on host buffers (4 MiB over PCIe,
the kernels, the row back)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    from effex_amd import synth
    from effex_amd.correlator import Correlator, SyntheticSource
    x = synth.synth_iq(5, 1, 2, 2 ** 18)[0].astype(np.complex128)
    for mode in ("SPECTRUM", "CONTINUUM"):
        cor = Correlator(source=SyntheticSource(), mode=mode)
        try:
            cor.gpu_iq_0[:] = x[0]
            cor.gpu_iq_1[:] = x[1]
            for _ in range(5):
                cor._run_task()
            t0 = time.perf_counter()
            n = 100
            for _ in range(n):
                cor._run_task()
            dt = (time.perf_counter() - t0) / n
            pair = np.empty((1, 2, x.shape[1]), dtype=np.complex64)
            t0 = time.perf_counter()
            for _ in range(n):
                pair[0, 0] = cor.gpu_iq_0
                pair[0, 1] = cor.gpu_iq_1
            fill = (time.perf_counter() - t0) / n
            plan = cor._plan()
            lib_mode = 'SPECTRUM' if mode == 'SPECTRUM' else 'CONTINUUM'
            t0 = time.perf_counter()
            for _ in range(n):
                plan.fx_rows(pair, lib_mode, cor.bandwidth)
            call = (time.perf_counter() - t0) / n
            print(json.dumps({"mode": mode, "ms_per_chunk_pair": round(dt * 1e3, 3),
                              "of_which_host_c128_to_c64_ms": round(fill * 1e3, 3),
                              "of_which_fx_rows_on_host_buffers_ms": round(call * 1e3, 3),
                              "Msamples_per_s": round(2 ** 18 / dt / 1e6, 1),
                              "x_realtime_at_2.4Msps": round(2 ** 18 / dt / 2.4e6, 1)}))
            if mode == "SPECTRUM":
                # the same call fed the receivers' bytes (FileSource / SocketSource hand these over, effex.py:652 on the device)
                rng = np.random.default_rng(3)
                cor._u8_pair = rng.integers(0, 256, size=(1, 2, x.shape[1], 2), dtype=np.uint8)
                for _ in range(5):
                    cor._run_task()
                t0 = time.perf_counter()
                for _ in range(n):
                    cor._run_task()
                dt = (time.perf_counter() - t0) / n
                cor._u8_pair = None
                print(json.dumps({"mode": mode, "input": "uint8 I,Q bytes", "ms_per_chunk_pair": round(dt * 1e3, 3),
                                  "x_realtime_at_2.4Msps": round(2 ** 18 / dt / 2.4e6, 1)}))
        finally:
            cor.close()


if __name__ == "__main__":
    main()
