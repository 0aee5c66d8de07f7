#!/usr/bin/env python3
"""Developer aid: the kernels compiled per channel count (fx_spec.h through hiprtc) against the any-shape mixed-radix kernel over
many channel counts -- every smooth count the search accepts in a range, or a random sample of them -- with random tap counts,
chunk lengths and chunk counts: rows, integration, byte ingest and the F stage alone.

    python tools/soak_spec.py [--seconds 600] [--seed 1] [--max-nchan 2048]
"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel_err(a, b):
    import numpy as np
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-nchan", type=int, default=2100)
    ap.add_argument("--min-nchan", type=int, default=2, help="2049 with --max-nchan 4097: the lean builds only")
    args = ap.parse_args()
    import numpy as np
    import torch
    from effex_amd import _lib, synth
    from effex_amd.plan import FxPlan
    lib = _lib.load()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    cases, worst, shapes = 0, {}, set()
    while time.time() < t_end:
        nchan = int(rng.integers(args.min_nchan, args.max_nchan))
        ntaps = int(rng.choice([1, 2, 3, 4, 4, 4]))
        if nchan & (nchan - 1) == 0:      # the powers of two have kernels of their own (tuned from 16 on, the generic radix-2 below)
            continue
        if lib.fxc_spec_probe(nchan, ntaps, 0, None, None, 0) != 0:
            continue
        frames = int(rng.integers(1, 400 if nchan <= 256 else (60 if nchan <= 1024 else 24)))
        num_samp = nchan * frames + int(rng.integers(0, nchan))
        n_chunks = max(1, min(int(rng.choice([1, 2, 3, 7, 40, 300, 700])), int(2.0e7 // (2 * num_samp))))
        x = torch.from_numpy(synth.synth_iq(int(rng.integers(1, 1 << 30)), n_chunks, 2, num_samp)).cuda()
        tag = dict(nchan=nchan, ntaps=ntaps, frames=frames, n_chunks=n_chunks, num_samp=num_samp)
        try:
            os.environ["FXC_RTC"] = "0"
            g = FxPlan(2, nchan, ntaps, num_samp)
            os.environ["FXC_RTC"] = "1"
            with FxPlan(2, nchan, ntaps, num_samp) as f, g:
                assert f.info["specialised"] & 1 and g.info["specialised"] == 0, (tag, f.path, f.info, g.info)
                shapes.add((nchan, f.info["block"], f.info["lds_bytes"]))
                rf, rg = f.fx_rows(x).cpu().numpy(), g.fx_rows(x).cpu().numpy()
                e = {"rows": rel_err(rf, rg)}
                f.fx_accumulate(x)
                e["integ"] = rel_err(f.finalize("SPECTRUM"), rg.astype(np.complex128).mean(axis=0))
                xs = x.reshape(-1, num_samp)[: max(1, min(2 * n_chunks, 5))]
                e["spectra"] = rel_err(f.channelize(xs).cpu().numpy(), g.channelize(xs).cpu().numpy())
                # (bit 1 of fxc_info.specialised: the F stage alone ran the build for this channel count, where there is one)
                assert bool(f.info["specialised"] & 2) == (lib.fxc_spec_probe(nchan, ntaps, 2, None, None, 0) == 0) and g.info["specialised"] == 0, tag
                if rng.random() < 0.3:
                    nb = min(n_chunks, 3)
                    u8 = torch.randint(0, 256, (nb, 2, num_samp, 2), dtype=torch.uint8, device="cuda")
                    e["bytes"] = rel_err(f.fx_rows_u8(u8, remove_dc=True).cpu().numpy(), g.fx_rows_u8(u8, remove_dc=True).cpu().numpy())
        except Exception as exc:
            print(json.dumps({"FAILED": str(exc), **tag}), flush=True)
            raise
        finally:
            os.environ.pop("FXC_RTC", None)
        for k, v in e.items():
            worst[k] = max(worst.get(k, 0.0), v)
        if not all(v < 6e-6 for v in e.values()):
            print(json.dumps({"MISMATCH": e, **tag}), flush=True)
            raise SystemExit(1)
        cases += 1
    print(json.dumps({"cases": cases, "channel_counts": len({s[0] for s in shapes}), "seconds": args.seconds, "seed": args.seed,
                      "worst_rel_err_specialised_vs_any_shape": worst}))


if __name__ == "__main__":
    main()
