#!/usr/bin/env python3
"""Developer aid: rows per second of the visibility sinks (effex_amd.rowsink) -- the reference's csv writer
(np.savetxt of one 4096-bin complex row per chunk pair, effex.py:687-696) against the binary sidecar -- on the host
alone, and fed by the device (fx_rows on resident chunk pairs -> host -> sink; host-fed FxPipeline popping straight
into a mapped window of the sidecar).

    python tools/bench_sink.py [--rows 4096] [--gpu]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--csv-rows", type=int, default=64)
    ap.add_argument("--gpu", action="store_true")
    args = ap.parse_args()
    import numpy as np
    from effex_amd import rowsink
    nbins = 4096
    rng = np.random.default_rng(1)
    rows = (rng.standard_normal((args.rows, nbins)) + 1j * rng.standard_normal((args.rows, nbins))).astype(np.complex64) * 1e-5
    header = rowsink.header_line(1, 2.4e6, 1.4204e9, 2 ** 18, nbins, 49.6, 'SPECTRUM')
    freqs = rowsink.spectrum_freqs(nbins, 2.4e6, 1.4204e9)
    out = {}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        t0 = time.perf_counter()
        with rowsink.CsvSink(os.path.join(tmp, "a.csv"), header, freqs) as s:
            s.write_rows(rows[:args.csv_rows])
        dt = time.perf_counter() - t0
        out["csv_rows_per_s"] = round(args.csv_rows / dt, 1)
        out["csv_bytes_per_row"] = os.path.getsize(os.path.join(tmp, "a.csv")) // (args.csv_rows + 2)
        t0 = time.perf_counter()
        with rowsink.BinSink(os.path.join(tmp, "a.fxb"), header, freqs, nbins) as s:
            for lo in range(0, args.rows, 256):
                s.write_rows(rows[lo:lo + 256])
        dt = time.perf_counter() - t0
        out["bin_rows_per_s_write"] = round(args.rows / dt, 1)
        t0 = time.perf_counter()
        with rowsink.BinSink(os.path.join(tmp, "b.fxb"), header, freqs, nbins) as s:
            view = s.reserve(args.rows)
            view[:] = rows
            s.commit(args.rows)
        dt = time.perf_counter() - t0
        out["bin_rows_per_s_mapped"] = round(args.rows / dt, 1)
        if args.gpu:
            import torch
            from effex_amd.plan import FxPlan, FxPipeline, synth_fill
            num_samp, n = 2 ** 18, 2048
            x = torch.empty((n, 2, num_samp), dtype=torch.complex64, device="cuda")
            synth_fill(x, 1234)
            with FxPlan(2, nbins, 4, num_samp) as plan:
                plan.fx_rows(x)
                plan.sync()
                for name, make in (("bin", lambda p: rowsink.BinSink(p, header, freqs, nbins)),):
                    t0 = time.perf_counter()
                    with make(os.path.join(tmp, "g.fxb")) as s:
                        for _ in range(4):
                            view = s.reserve(n)
                            r = plan.fx_rows(x)[:, 0]                 # [n, nbins] complex64 on the device
                            torch.from_numpy(view).copy_(r)           # D2H straight into the mapped window
                            s.commit(n)
                    dt = time.perf_counter() - t0
                    out["device_resident_to_%s_rows_per_s" % name] = round(4 * n / dt, 1)
                # host-fed: chunk pairs cross PCIe both ways; rows popped straight into the sidecar
                batch, nb = 16, 32
                xh = x[:batch].cpu().numpy()
                with rowsink.BinSink(os.path.join(tmp, "p.fxb"), header, freqs, nbins) as s, \
                        FxPipeline(plan, batch, depth=2, mode="SPECTRUM") as pipe:
                    view = s.reserve(batch * nb)
                    t0 = time.perf_counter()
                    pipe.push(xh)
                    for b in range(nb):
                        if b + 1 < nb:
                            pipe.push(xh)
                        pipe.pop(out=view[b * batch:(b + 1) * batch])
                        s.commit(batch)
                    dt = time.perf_counter() - t0
                out["host_fed_pipeline_to_bin_rows_per_s"] = round(batch * nb / dt, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
