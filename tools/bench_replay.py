#!/usr/bin/env python3
"""Developer aid: replaying two recordings (rtl_sdr-style bytes, or complex64 samples with --fmt c64: DC removal on the
device, fxc_pipe_create_iq) through the drop-in state machine -- one device call per
chunk pair (the reference's loop, effex.py:391-410) against Correlator(batch=...) -- rows and samples per second, binary
sidecar output.  The recordings are synthetic bytes written to a temporary directory (page cache)."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=257, help="chunk pairs per recording (the first one calibrates)")
    ap.add_argument("--num-samp", type=int, default=2 ** 18)
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4, 16, 64])
    ap.add_argument("--mode", default="SPECTRUM")
    ap.add_argument("--fmt", default="u8", choices=["u8", "c64"])
    ap.add_argument("--nbins", type=int, default=2 ** 12, help="--resolution of the reference (any integer)")
    args = ap.parse_args()
    from effex_amd.correlator import Correlator, FileSource
    rng = np.random.default_rng(7)
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for a in range(2):
            path = os.path.join(tmp, "rx%d.%s" % (a, args.fmt))
            raw = rng.integers(0, 256, size=args.chunks * args.num_samp * 2, dtype=np.uint8)
            if args.fmt == "c64":
                raw = ((raw.astype(np.float32) - 127.5) / 127.5).view(np.complex64)
            raw.tofile(path)
            paths.append(path)
        for batch in args.batches:
            out = os.path.join(tmp, "rows_%d.fxb" % batch)
            cor = Correlator(num_samp=args.num_samp, nbins=args.nbins, source=FileSource(paths[0], paths[1], fmt=args.fmt), output_file=out,
                             output_format='bin', mode=args.mode, batch=batch)
            cor._plan()                       # plan creation is not part of the replay
            t0 = time.perf_counter()
            rows = cor.run_state_machine()
            dt = time.perf_counter() - t0
            print(json.dumps({"fmt": args.fmt, "nbins": args.nbins, "batch": batch, "rows": rows, "seconds": round(dt, 4), "rows_per_s": round(rows / dt, 1),
                              "Msamples_per_s": round(rows * args.num_samp / dt / 1e6, 1),
                              "x_realtime_at_2.4Msps": round(rows * args.num_samp / dt / 2.4e6, 1)}))
            os.remove(out)


if __name__ == "__main__":
    main()
