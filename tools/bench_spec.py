#!/usr/bin/env python3
"""Developer aid: the two-antenna F+X kernel compiled per channel count (fx_spec.h through hiprtc) against the any-shape
mixed-radix kernel (FXC_RTC=0, a child process) on the same box: integration time over 1 024 chunk pairs of 2^18 samples,
and -- with --check -- both against the oracle on a small case.

    python tools/bench_spec.py [--cases 1000,96,1536,...] [--reps 5] [--check] [--u8]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def run_cases(args):
    import numpy as np
    import torch
    from effex_amd import synth
    from effex_amd.plan import FxPlan, synth_fill
    from effex_amd.window import design_window
    num_samp, n_chunks = 2 ** 18, args.chunks
    x = torch.empty((n_chunks, 2, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 1234)
    xb = torch.randint(0, 256, (n_chunks, 2, num_samp, 2), dtype=torch.uint8, device="cuda") if args.u8 else None
    for nchan in (int(v) for v in args.cases.split(",")):
        out = {"tag": "rtc" if os.environ.get("FXC_RTC", "1") != "0" else "any-shape", "nchan": nchan, "ntaps": args.taps}
        with FxPlan(2, nchan, args.taps, num_samp, dev=args.dev) as plan:
            info = plan.info
            out.update(path=plan.path, specialised=info["specialised"], vgprs=info["spec_vgprs"], block=info["block"], lds=info["lds_bytes"])
            if args.check:
                import fx_oracle
                ns, nc = nchan * 37 + 5, 3
                xs = synth.synth_iq(99 + nchan, nc, 2, ns)
                window = design_window(args.taps, nchan)
                with FxPlan(2, nchan, args.taps, ns, window=window, dev=args.dev) as small:
                    rows = small.fx_rows(torch.from_numpy(xs).cuda(), "SPECTRUM").cpu().numpy()
                    small.fx_accumulate(torch.from_numpy(xs).cuda())
                    integ = small.finalize("SPECTRUM")
                    out["check_specialised"] = small.info["specialised"]
                ref = np.stack([fx_oracle.pfb_xcorr(xs[c, 0], xs[c, 1], args.taps, nchan, window, 2.4e6, 1.4204e9, 0.0, "SPECTRUM")
                                for c in range(nc)])
                out["rows_err"] = float(np.abs(rows[:, 0] - ref).max() / np.abs(ref).max())
                out["integ_err"] = float(np.abs(integ[0] - ref.mean(axis=0)).max() / np.abs(ref.mean(axis=0)).max())
            call = (lambda: plan.fx_accumulate_u8(xb, remove_dc=True)) if args.u8 else (lambda: plan.fx_accumulate(x))
            call()
            plan.finalize()
            ms = []
            for _ in range(args.reps):
                plan.timer_start()
                call()
                ms.append(plan.timer_stop())
                plan.finalize()
            ms.sort()
            med = ms[len(ms) // 2]
            gb = n_chunks * 2 * num_samp * 8 / 1e9
            out.update(median_ms=round(med, 3), algorithmic_GBps=round(gb / med * 1e3, 1), frac_of_8TBs=round(gb / med * 1e3 / 8000, 4))
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--taps", type=int, default=4)
    ap.add_argument("--chunks", type=int, default=1024)
    ap.add_argument("--cases", default="1000,96,1536,3000,250,720,1001,2000,12,4000,5000,600,1200")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--u8", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--dev", action="store_true", help="the developer library (libfxcorr_dev.so): FXC_RTC_ABL timing ablations live there")
    args = ap.parse_args()
    if args.child:
        return run_cases(args)
    for rtc in ("1", "0"):       # each arm in a process of its own (the knob is read when a plan is made; separate processes keep it honest)
        env = dict(os.environ, FXC_RTC=rtc)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:], env=env, check=False)


if __name__ == "__main__":
    main()
