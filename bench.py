#!/usr/bin/env python3
"""bench.py — the headline measurement of BASELINE.json on MI355X.

One *step* = one full integration: reset the accumulator, push this rank's integration frames
(chunk pairs, device-resident complex64 IQ) through the fused F+X HIP kernel, reduce the exported
cross-spectra across ranks (RCCL, N > 1) and finalise to host.  Workload at every N: BASELINE.json
configs[1] — 2 antennas, num_samp = 262144, ntaps = 4, nchan = 4096, 10 000 frames *per GPU* (weak
scaling).  ``value`` = samples per antenna stream processed by all ranks / wall time (one sample = one
complex time sample per antenna stream, so a chunk pair counts 262144 samples — SURVEY.md §8d).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_SAMP = 262144
NCHAN = 4096
NTAPS = 4
N_ANT = 2
FRAMES = 10000
SEED = 1234
BANDWIDTH = 2.4e6
FREQUENCY = 1.4204e9
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md chip table (spec); measured copy ceiling 6290
BYTES_PER_FRAME = N_ANT * NUM_SAMP * 8          # complex64 IQ read once
ACC_BYTES = NCHAN * 16 * 2                      # per workgroup: float64 partial row read + written per chunk


# ----------------------------------------------------------------------------------------------
# CPU baseline: the oracle (numpy/scipy restatement, "port") on the host cores, bounded sample.
# Runs BEFORE this process touches the GPU; workers are spawned, never forked from a HIP process.
# ----------------------------------------------------------------------------------------------
def _cpu_worker(args):
    seconds, seed_offset = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import numpy as np
    import fx_oracle
    from effex_amd import synth
    from effex_amd.window import design_window
    window = design_window(NTAPS, NCHAN).astype(np.float32)
    x = synth.synth_iq(SEED + seed_offset, 1, 2, NUM_SAMP)[0]
    fx_oracle.pfb_xcorr(x[0], x[1], NTAPS, NCHAN, window, BANDWIDTH, FREQUENCY, 0.0, "SPECTRUM", dtype=np.complex64)
    frames = 0
    t0 = time.perf_counter()
    while True:
        fx_oracle.pfb_xcorr(x[0], x[1], NTAPS, NCHAN, window, BANDWIDTH, FREQUENCY, 0.0, "SPECTRUM",
                            dtype=np.complex64)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return frames, dt


def cpu_baseline(seconds=8.0):
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    workers = max(1, min(cores, 64))
    frames1, dt1 = _cpu_worker((min(seconds, 4.0), 0))
    single = frames1 * NUM_SAMP / dt1 / 1e6
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(seconds, k) for k in range(workers)])
    multi = sum(f * NUM_SAMP / dt for f, dt in res) / 1e6
    total_frames = sum(f for f, _ in res)
    return {"value": round(multi, 2), "unit": "Msamples/s", "cores": workers, "kind": "port",
            "single_core_value": round(single, 2),
            "sample": "%d frames of the same workload (S=%d, N=%d, T=%d, 2 ant, complex64 numpy/scipy oracle), "
                      "%d worker processes x %.0f s on independent frames" % (total_frames, NUM_SAMP, NCHAN, NTAPS,
                                                                             workers, seconds)}


def pmc_traffic_per_frame():
    """HBM bytes per frame from the newest committed PMC summary (profiles/*/pmc_hbm_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 correction)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_hbm_traffic.json"))):
        try:
            with open(path) as fh:
                d = json.load(fh)
            best = (d["hbm_traffic_bytes_per_launch"] / d["frames_per_launch"], os.path.relpath(path, ROOT))
        except Exception:
            pass
    return best


def power_sample(step, seconds=4.0):
    """Package power and shader clock read by rocm-smi while `step` keeps the GPU busy (untimed, after the timed
    region).  Evidence for DESIGN.md §6 (the kernel runs at the package power cap); None if rocm-smi is missing."""
    import re
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    try:
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end - 2.0:      # let the clocks settle under load first
            step()
        proc = subprocess.Popen([exe, "--showpower", "--showclocks", "--showmaxpower"], stdout=subprocess.PIPE,
                                stderr=subprocess.DEVNULL, text=True)
        while proc.poll() is None:
            step()
        text = proc.stdout.read()
        watts = re.search(r"GPU\[0\]\s*:\s*(?:Current Socket|Average) Graphics Package Power \(W\):\s*([0-9.]+)", text)
        cap = re.search(r"GPU\[0\].*?Max Graphics Package Power \(W\):\s*([0-9.]+)", text)
        sclk = re.search(r"GPU\[0\].*?sclk clock level:.*?\((\d+)Mhz\)", text)
        if not watts:
            return None
        return {"package_w": float(watts.group(1)), "cap_w": float(cap.group(1)) if cap else None,
                "sclk_mhz": int(sclk.group(1)) if sclk else None, "source": "rocm-smi, sampled under load after the timed region"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=FRAMES, help="integration frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power sample after the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)       # before any HIP initialisation in this process

    import numpy as np
    import torch
    import torch.distributed as dist
    from effex_amd import sharding
    from effex_amd.plan import FxPlan, synth_fill

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # --- synthetic input, device resident: `frames` distinct chunk pairs per rank if they fit --------
    frames = args.frames
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    need = frames * BYTES_PER_FRAME
    pool_frames = frames if need < 0.8 * free_b else max(512, int(0.5 * free_b // BYTES_PER_FRAME))
    x = torch.empty((pool_frames, N_ANT, NUM_SAMP), dtype=torch.complex64, device=dev)
    synth_fill(x, SEED, first_chunk=rank * frames)
    torch.cuda.synchronize(dev)

    plan = FxPlan(N_ANT, NCHAN, NTAPS, NUM_SAMP, device=local_rank)
    assert plan.path == "fused", "headline workload must run on the fused HIP kernel"
    plan.set_delay(BANDWIDTH, FREQUENCY, 0.0)
    integ = sharding.ShardedIntegrator(plan, rank, world)

    def step():
        done = 0
        while done < frames:                      # one launch when the whole run is resident
            n = min(pool_frames, frames - done)
            plan.fx_accumulate(x[:n])
            done += n
        return integ.finalize("SPECTRUM", BANDWIDTH, root=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    out = None
    for _ in range(args.warmup):
        out = step()
    plan.kernel_profiling(True)
    plan.kernel_time(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = plan.kernel_time(reset=True)
    plan.kernel_profiling(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    power = power_sample(step) if (world == 1 and not args.no_power) else None

    if rank == 0:
        assert out is not None and np.isfinite(out).all() and np.abs(out).max() > 0
        samples = float(frames) * NUM_SAMP * world * args.steps
        value = samples / elapsed / 1e6
        frames_per_launch = frames * args.steps / max(launches, 1)
        algo_bytes = frames_per_launch * BYTES_PER_FRAME + NCHAN * 16        # + one cross-spectrum per integration
        avg_kernel_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = algo_bytes / avg_kernel_s / 1e9
        pmc = pmc_traffic_per_frame()
        line = {
            "metric": "2-ant FX correlator throughput (PFB+FFT+X, integrated)",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "value_per_gpu": round(value / world, 1),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 2-antenna FX, num_samp=262144, ntaps=4, nchan=4096, "
                                   "%d integration frames per GPU per step" % frames,
                       "frames_per_gpu": frames, "resident_frames": pool_frames, "num_samp": NUM_SAMP,
                       "nchan": NCHAN, "ntaps": NTAPS, "n_ant": N_ANT, "path": plan.path,
                       "sample_definition": "one complex sample per antenna stream",
                       "parallelism": "frames sharded over %d GPU(s), one RCCL reduce of the cross-spectra "
                                      "per integration" % world},
            "roofline": {"bound": "hbm", "kernel": "fx_fused4096_kernel", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_of_measured_copy_ceiling": round(achieved / 6290.0, 4),
                         "bytes_per_launch": int(algo_bytes), "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                         "launches": int(launches),
                         "traffic": None if pmc is None else int(pmc[0] * frames_per_launch),
                         "traffic_source": None if pmc is None else pmc[1]},
            "cpu_baseline": cpu,
            "power": power,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
