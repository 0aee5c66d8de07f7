#!/usr/bin/env python3
"""Developer aid: wall time of the drop-in Correlator._run_task() per chunk pair (host complex128 buffers in, host
row out — exactly the reference's call, effex.py:490-527), SPECTRUM and CONTINUUM."""
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
            print(json.dumps({"mode": mode, "ms_per_chunk_pair": round(dt * 1e3, 3),
                              "Msamples_per_s": round(2 ** 18 / dt / 1e6, 1),
                              "x_realtime_at_2.4Msps": round(2 ** 18 / dt / 2.4e6, 1)}))
        finally:
            cor.close()


if __name__ == "__main__":
    main()
