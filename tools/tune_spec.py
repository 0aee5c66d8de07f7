#!/usr/bin/env python3
"""Developer aid (run ON THE GPU BOX): which stage list of fx_spec.h is fastest for a channel count -- the prime-factor order of round 5
and the first K candidates of h_rtc.h::spec_stage_lists (developer library: FXC_RTC_COMPOSITE / FXC_RTC_PICK), each checked against the
oracle and timed over 1 024 chunk pairs of 2^18 samples in ONE process.  One JSON line per (channel count, candidate);
tools/make_spec_table.py turns the log into effex_amd/csrc/spec_tuned.h.

    python tools/tune_spec.py --cases 1000,1200,... [--picks 4] [--taps 4] >> gpurun_out/r06/tune.jsonl
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
KNOBS = ("FXC_RTC_COMPOSITE", "FXC_RTC_PICK", "FXC_RTC_RADICES", "FXC_RTC_U", "FXC_RTC_TW_EARLY", "FXC_RTC_OOB_ZERO", "FXC_RTC_TUNED", "FXC_RTC_LAYOUT", "FXC_RTC_WAVES", "FXC_RTC_GROUPS")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", required=True)
    ap.add_argument("--picks", type=int, default=4)
    ap.add_argument("--taps", type=int, default=4)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--chunks", type=int, default=1024)
    ap.add_argument("--lists", default="", help="extra stage lists to time for every case that they multiply to, e.g. 4,25,10;5,20,10")
    ap.add_argument("--mode", default="fx", choices=("fx", "f"), help="fx: two antennas, F + X (above 4096 channels: the F-only build + the second pass); "
                                                                     "f: the F stage alone (fxc_channelize over 1 024 streams)")
    ap.add_argument("--env-arms", default="", help="instead of legacy / pick arms: one arm per ';'-separated group of NAME=VALUE assignments (',' between them), "
                                                   "e.g. 'FXC_RTC_LAYOUT=0;FXC_RTC_LAYOUT=1' -- the library's own choice under each")
    args = ap.parse_args()
    import numpy as np
    import torch
    import fx_oracle
    from effex_amd import _lib, synth
    from effex_amd.plan import FxPlan, synth_fill
    from effex_amd.window import design_window
    lib = _lib.load(dev=True)
    num_samp = 2 ** 18
    x = torch.empty((args.chunks, 2, num_samp), dtype=torch.complex64, device="cuda")
    synth_fill(x, 1234)
    extra = [tuple(int(v) for v in l.split(",")) for l in args.lists.split(";") if l]
    for nchan in (int(v) for v in args.cases.split(",")):
        # (legacy: round 5's prime-factor order; tuned: what the library chooses today -- spec_tuned.h or its ranking; pickK: the K-th ranked list)
        arms = [("legacy", {"FXC_RTC_COMPOSITE": "0"}), ("tuned", {})] + [("pick%d" % k, {"FXC_RTC_PICK": str(k)}) for k in range(args.picks)]
        arms += [("list", {"FXC_RTC_RADICES": ",".join(map(str, l))}) for l in extra if int(np.prod(l)) == nchan]
        if args.env_arms:
            arms = [(grp, dict(kv.split("=") for kv in grp.split(",") if kv)) for grp in args.env_arms.split(";")]
        ns, nc = nchan * 37 + 5, 3
        xs = synth.synth_iq(99 + nchan, nc, 2, ns)
        window = design_window(args.taps, nchan)
        ref = np.stack([fx_oracle.pfb_xcorr(xs[c, 0], xs[c, 1], args.taps, nchan, window, 2.4e6, 1.4204e9, 0.0, "SPECTRUM") for c in range(nc)])
        seen = set()
        for tag, env in arms:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            # the build(s) this arm gets: F + X in one pass (variant 0), or -- the F stage alone, and above 4096 channels for two antennas the
            # F-only build + the second pass -- variants 2 and 3
            variants = (2,) if args.mode == "f" else ((0,) if nchan <= 4096 else (2, 3))
            reps_ = {}
            failed = None
            for v in variants:
                buf = ctypes.create_string_buffer(1024)
                if lib.fxc_spec_probe(nchan, args.taps, v, None, buf, len(buf)) != 0:
                    failed = (lib.fxc_last_error(None) or b"").decode()[:200]
                    break
                reps_[v] = dict(kv.split("=") for kv in buf.value.decode().split())
            if failed is not None:
                print(json.dumps({"nchan": nchan, "tag": tag, "mode": args.mode, "error": failed}), flush=True)
                continue
            rep = reps_[variants[0]]
            key = tuple((reps_[v]["stages"], reps_[v]["frames_per_step"], reps_[v]["groups"]) for v in variants) + ((tag,) if args.env_arms else ())
            if key in seen:
                continue
            seen.add(key)
            out = {"nchan": nchan, "ntaps": args.taps, "tag": tag, "mode": args.mode, "stages": rep["stages"], "u": int(rep["frames_per_step"]), "groups": rep["groups"],
                   "vgprs": int(rep["vgprs"]), "threads": int(rep["tpr"]) * int(rep["slots"]), "resident": int(rep["resident"])}
            if 3 in reps_:
                out.update(stages_xm=reps_[3]["stages"], u_xm=int(reps_[3]["frames_per_step"]), vgprs_xm=int(reps_[3]["vgprs"]))
            if args.mode == "f":
                xs1 = xs.reshape(-1, ns)
                with FxPlan(1, nchan, args.taps, ns, window=window, dev=True) as small:
                    spec = small.channelize(torch.from_numpy(xs1).cuda()).cpu().numpy()
                    out["specialised"] = small.info["specialised"]
                ref_f = fx_oracle.spectrometer_poly(xs1[0], args.taps, nchan, window)
                out["rows_err"] = float(np.abs(spec[0] - ref_f).max() / np.abs(ref_f).max())
                xf = x.view(-1, num_samp)
                with FxPlan(1, nchan, args.taps, num_samp, dev=True) as plan:
                    o = plan.channelize(xf)
                    plan.sync()
                    ms = []
                    for _ in range(args.reps):
                        del o
                        plan.timer_start()
                        o = plan.channelize(xf)
                        ms.append(plan.timer_stop())
                    del o
            else:
                with FxPlan(2, nchan, args.taps, ns, window=window, dev=True) as small:
                    rows = small.fx_rows(torch.from_numpy(xs).cuda(), "SPECTRUM").cpu().numpy()
                    out["specialised"] = small.info["specialised"]
                out["rows_err"] = float(np.abs(rows[:, 0] - ref).max() / np.abs(ref).max())
                with FxPlan(2, nchan, args.taps, num_samp, dev=True) as plan:
                    plan.fx_accumulate(x)
                    plan.finalize()
                    ms = []
                    for _ in range(args.reps):
                        plan.timer_start()
                        plan.fx_accumulate(x)
                        ms.append(plan.timer_stop())
                        plan.finalize()
            ms.sort()
            out["ms"] = round(ms[len(ms) // 2], 4)
            out["frac_of_8TBs"] = round(args.chunks * 2 * num_samp * 8 / 1e9 / out["ms"] * 1e3 / 8000, 4)
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
