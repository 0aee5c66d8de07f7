#!/usr/bin/env python3
"""bench.py — the headline measurement of BASELINE.json on MI355X.

One *step* = one full integration: reset the accumulator, push this rank's integration frames
(chunk pairs, device-resident complex64 IQ) through the fused F+X HIP kernel, reduce the exported
cross-spectra across ranks (RCCL, N > 1) and finalise to host.  Workload at every N: BASELINE.json
configs[1] — 2 antennas, num_samp = 262144, ntaps = 4, nchan = 4096, 10 000 frames *per GPU* (weak
scaling).  ``value`` = samples per antenna stream processed by all ranks / wall time (one sample = one
complex time sample per antenna stream, so a chunk pair counts 262144 samples — SURVEY.md §8d).

After the timed region (untimed) the run checks what it timed: the integration against the float64 mean of
the per-frame rows over the same frames, and sampled frames against oracle rows computed in the
``cpu_baseline`` leg (1e-5 of max|vis|); it fails otherwise.  ``other_configs`` carries
short runs of BASELINE configs[2] (nchan = 1, num_samp = 2^20) and configs[4] (8 antennas, nchan 4096).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --dry-run-dist   # control flow on gloo/CPU
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
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md chip table (spec); its float4-copy ceiling is 6290 (read-only stream measured here: 6100-6500)
BYTES_PER_FRAME = N_ANT * NUM_SAMP * 8          # complex64 IQ read once
TOL_VIS = 1e-5                 # vs the float64 oracle, of max|vis|
CHECK_FRAMES = (0, 7777)       # frames of rank 0 whose rows are checked against the oracle


# ----------------------------------------------------------------------------------------------
# CPU baseline: the oracle (numpy/scipy restatement, "port") on the host cores, bounded sample.
# Runs BEFORE this process touches the GPU; workers are spawned, never forked from a HIP process.
# The same leg produces the oracle rows the GPU result is checked against after the timed region.
# ----------------------------------------------------------------------------------------------
def _cpu_worker(args):
    seconds, seed_offset = args[:2]
    c128 = len(args) > 2 and args[2]          # the reference's own precision (effex.py:109-110, 551), single core once
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import numpy as np
    import fx_oracle
    from effex_amd import synth
    from effex_amd.window import design_window
    window = design_window(NTAPS, NCHAN) if c128 else design_window(NTAPS, NCHAN).astype(np.float32)
    dtype = np.complex128 if c128 else np.complex64
    x = synth.synth_iq(SEED + seed_offset, 1, 2, NUM_SAMP)[0]
    if c128:
        x = x.astype(np.complex128)
    fx_oracle.pfb_xcorr(x[0], x[1], NTAPS, NCHAN, window, BANDWIDTH, FREQUENCY, 0.0, "SPECTRUM", dtype=dtype)
    frames = 0
    t0 = time.perf_counter()
    while True:
        fx_oracle.pfb_xcorr(x[0], x[1], NTAPS, NCHAN, window, BANDWIDTH, FREQUENCY, 0.0, "SPECTRUM", dtype=dtype)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return frames, dt


def _oracle_rows(frame_ids):
    """float64 oracle rows of the given frames of rank 0's synthetic stream (checker, not measured)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fx_oracle
    from effex_amd import synth
    from effex_amd.window import design_window
    window = design_window(NTAPS, NCHAN)
    rows = {}
    for f in frame_ids:
        x = synth.synth_iq(SEED, 1, 2, NUM_SAMP, first_chunk=f)[0]
        rows[f] = fx_oracle.pfb_xcorr(x[0], x[1], NTAPS, NCHAN, window, BANDWIDTH, FREQUENCY, 0.0, "SPECTRUM")
    return rows


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(seconds=8.0, frames=FRAMES):
    import multiprocessing as mp
    logical = os.cpu_count() or 1
    usable = logical
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        pass
    workers = max(1, usable)                 # one single-threaded worker per usable core, no cap
    frames1, dt1 = _cpu_worker((min(seconds, 4.0), 0))
    single = frames1 * NUM_SAMP / dt1 / 1e6
    frames2, dt2 = _cpu_worker((2.0, 0, True))
    single_c128 = frames2 * NUM_SAMP / dt2 / 1e6
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(seconds, k) for k in range(workers)])
    multi = sum(f * NUM_SAMP / dt for f, dt in res) / 1e6
    total_frames = sum(f for f, _ in res)
    check = _oracle_rows([f for f in CHECK_FRAMES if f < frames])
    return {"value": round(multi, 2), "unit": "Msamples/s", "cores": workers, "kind": "port",
            "single_core_value": round(single, 2), "single_core_complex128_value": round(single_c128, 2),
            "cpu_count_logical": logical, "cpu_count_usable": usable,
            "cpu_model": cpu_model(),
            "sample": "%d frames of the same workload (S=%d, N=%d, T=%d, 2 ant, complex64 numpy/scipy oracle), "
                      "%d single-threaded worker processes (one per usable core) x %.0f s on independent frames"
                      % (total_frames, NUM_SAMP, NCHAN, NTAPS, workers, seconds)}, check


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


def rank_env(args):
    """RANK / LOCAL_RANK / WORLD_SIZE as torch.distributed.run exports them; --gpus must agree with the launch."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            # started bare with --gpus N: launch the ranks ourselves (a child process, before anything here touches HIP)
            import socket
            import subprocess
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            raise SystemExit(subprocess.call(cmd))
        args.gpus = world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    return rank, local_rank, world


def max_over_ranks(seconds, world, device):
    import torch
    import torch.distributed as dist
    if world > 1 and dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------------------------
# --dry-run-dist: the world > 1 control flow of main() on gloo / CPU tensors — rank env, per-rank first_chunk,
# ShardedIntegrator.finalize (reduce to root + finalize on root + reset), barrier, max-over-ranks timing — with a
# stand-in for the plan, so the first 8-GPU launch is not the first time this code runs.
# ----------------------------------------------------------------------------------------------
class _DryPlan(object):
    """Stands in for FxPlan: 'integrates' by adding the chunk indices it is given."""
    n_baselines, nchan, _follow = 1, 8, True

    def __init__(self):
        import torch
        self.acc = torch.zeros(self.nchan + 1, dtype=torch.complex128)

    def new_sums(self):
        import torch
        return torch.zeros(self.nchan + 1, dtype=torch.complex128)

    def fx_accumulate(self, chunk_ids):
        for c in chunk_ids:
            self.acc[: self.nchan] += complex(c, -c)
            self.acc[self.nchan] += 1
        return len(chunk_ids)

    def acc_export(self, sums):
        sums.copy_(self.acc)
        return sums

    def finalize_sums(self, sums, mode="SPECTRUM", bandwidth=1.0):
        return (sums[: self.nchan] / sums[self.nchan].real).numpy()

    def acc_reset(self):
        self.acc.zero_()

    def sync(self):
        pass


def dry_run_dist(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from effex_amd import sharding
    rank, local_rank, world = rank_env(args)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = args.frames
    first_chunk = rank * frames                       # weak scaling: every rank owns `frames` frames
    plan = _DryPlan()
    integ = sharding.ShardedIntegrator(plan, rank, world)

    def step():
        plan.fx_accumulate(range(first_chunk, first_chunk + frames))
        return integ.finalize("SPECTRUM", BANDWIDTH, root=0)

    def fence():
        if world > 1:
            dist.barrier()

    out = None
    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, torch.device("cpu"))
    if rank == 0:
        total = world * frames
        want = complex((total - 1) / 2.0, -(total - 1) / 2.0)      # mean of 0 .. total-1
        assert out is not None and np.allclose(out, want), (out, want)
        print(json.dumps({"dry_run": True, "backend": "gloo", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "frames_per_rank": frames, "first_chunk_last_rank": (world - 1) * frames,
                          "transport": integ.transport, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
                          "mean_chunk_index": want.real}))
    else:
        assert out is None
    if world > 1:
        dist.destroy_process_group()


def other_configs(x, dev, reps=5):
    """Short, untimed-region runs of the other single-GPU BASELINE configs on the resident synthetic bytes (viewed with
    their own shapes): configs[2] continuum streaming limit (nchan = 1, num_samp = 2^20) and configs[4] (8 antennas,
    28 baselines, nchan 4096), plus two three-pass shapes (32 taps, 8192 channels).  HIP-event median of `reps` calls each."""
    import numpy as np
    from effex_amd.plan import FxPlan
    out = []
    flat = x.view(-1)

    def run(name, n_ant, nchan, num_samp, n_chunks, window, mode, rows, ntaps=NTAPS):
        need = n_chunks * n_ant * num_samp
        if flat.numel() < need:
            return
        xv = flat[:need].view(n_chunks, n_ant, num_samp)
        with FxPlan(n_ant, nchan, ntaps, num_samp, window=window, device=dev.index) as plan:
            def call():
                if rows:
                    plan.fx_rows(xv, mode, BANDWIDTH)
                else:
                    plan.acc_reset()
                    plan.fx_accumulate(xv)
                    plan.finalize(mode, BANDWIDTH)
            call()
            plan.sync()
            ms = []
            for _ in range(reps):
                plan.timer_start()
                call()
                ms.append(plan.timer_stop())
            ms.sort()
            med = ms[len(ms) // 2]
            algo = need * 8
            out.append({"config": name, "path": plan.path, "n_ant": n_ant, "nchan": nchan, "ntaps": ntaps, "num_samp": num_samp,
                        "n_chunks": n_chunks, "mode": mode, "median_ms": round(med, 4),
                        "value": round(n_chunks * num_samp / med / 1e3, 1), "unit": "Msamples/s",
                        "algorithmic_GBps": round(algo / med / 1e6, 1), "frac_of_8TBs": round(algo / med / 1e6 / HBM_PEAK_GBS, 4)})

    run("configs[2]: continuum streaming limit, nchan=1, num_samp=2^20, one scalar per chunk pair", 2, 1, 2 ** 20, 2048,
        np.array([0.4, 0.3, 0.2, 0.1]), "CONTINUUM", True)
    run("configs[4]: 8 antennas, 28 baselines, nchan=4096, num_samp=262144, integrated", 8, NCHAN, NUM_SAMP, 512, None,
        "SPECTRUM", False)
    # not BASELINE configs: the reference test's own shape (tests/test_effex.py:62-66) and the largest --nfft of its CLI
    # examples; both are three-pass routes (3 x the algorithmic traffic by construction)
    run("reference test shape: 2 antennas, nchan=2048, ntaps=32, num_samp=262144, integrated", 2, 2048, NUM_SAMP, 1024, None,
        "SPECTRUM", False, ntaps=32)
    run("--nfft 8192: 2 antennas, nchan=8192, ntaps=4, num_samp=262144, integrated", 2, 8192, NUM_SAMP, 1024, None,
        "SPECTRUM", False)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps first; after idle the clock needs about five launches to settle")
    ap.add_argument("--frames", type=int, default=FRAMES, help="integration frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power sample after the timed region")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of configs[2], configs[4] and the two three-pass shapes")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the check after the timed region (profiling runs: its fx_rows launches would mix into "
                         "the per-kernel statistics)")
    ap.add_argument("--reduce", choices=("rccl", "torch"), default="rccl",
                    help="N > 1: fxc_reduce (libfxcorr calls RCCL on the plan's stream) or torch.distributed")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend for N > 1.  gloo + fewer GPUs than ranks (ranks share GPUs round-robin) "
                         "runs the real multi-rank flow on a one-GPU box for testing; its timing means nothing")
    ap.add_argument("--dry-run-dist", action="store_true",
                    help="run the multi-rank control flow on gloo / CPU tensors with a stand-in plan (no GPU)")
    args = ap.parse_args()

    if args.dry_run_dist:
        if args.frames == FRAMES:
            args.frames = 100
        return dry_run_dist(args)

    rank, local_rank, world = rank_env(args)

    from effex_amd import _lib
    if not _lib.is_in_tree():
        raise SystemExit("bench.py measures the in-tree library only (FXCORR_LIB points at %s)" % _lib.LIB_PATH)

    cpu, check_rows = None, {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, check_rows = cpu_baseline(args.cpu_seconds, args.frames)   # before any HIP initialisation in this process

    import numpy as np
    import torch
    import torch.distributed as dist
    from effex_amd import sharding
    from effex_amd.plan import FxPlan, synth_fill

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    gpu = local_rank % n_dev          # (a launcher may also show every rank just its own GPU: then this is device 0)
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    # --- synthetic input, device resident: `frames` distinct chunk pairs per rank if they fit --------
    frames = args.frames
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    need = frames * BYTES_PER_FRAME
    pool_frames = frames if need < 0.8 * free_b else max(512, int(0.5 * free_b // BYTES_PER_FRAME))
    x = torch.empty((pool_frames, N_ANT, NUM_SAMP), dtype=torch.complex64, device=dev)
    synth_fill(x, SEED, first_chunk=rank * frames)
    torch.cuda.synchronize(dev)

    plan = FxPlan(N_ANT, NCHAN, NTAPS, NUM_SAMP, device=gpu)
    assert plan.path == "fused", "headline workload must run on the fused HIP kernel"
    plan.set_delay(BANDWIDTH, FREQUENCY, 0.0)
    comm, comm_note = None, None
    if world > 1 and args.reduce == "rccl":
        try:
            if args.dist_backend != "nccl":
                raise RuntimeError("ranks share GPUs in the gloo test set-up: RCCL wants one GPU per rank")
            comm = sharding.make_comm(gpu, rank, world)
        except Exception as exc:                  # RCCL not usable: torch.distributed carries the reduce
            comm, comm_note = None, "no RCCL communicator (%s)" % exc
        ok = torch.tensor([1 if comm is not None else 0], device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)           # all ranks or none
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
    integ = sharding.ShardedIntegrator(plan, rank, world, comm=comm)

    def issue():
        """Queue one integration: the F+X launch(es) over this rank's frames, then reduce + finalize + reset (one
        kernel on one GPU).  Nothing here waits for the device."""
        done = 0
        while done < frames:                      # one launch when the whole run is resident
            n = min(pool_frames, frames - done)
            plan.fx_accumulate(x[:n])
            done += n
        integ.finalize_async("SPECTRUM", BANDWIDTH, root=0)

    def run_steps(k):
        """k integrations; every one is finalised to host memory and collected.  Integration j + 1 is queued before the
        host waits for the result of j (two may be in flight), so the device never idles on the host's round trip."""
        res = None
        for j in range(k):
            issue()
            if j > 0:
                res = integ.finalize_wait()
        if k > 0:
            res = integ.finalize_wait()
        return res

    def step():
        return run_steps(1)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    out = run_steps(args.warmup)
    plan.kernel_profiling(True)
    plan.kernel_time(reset=True)
    fence()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = plan.kernel_time(reset=True)
    plan.kernel_profiling(False)
    elapsed = max_over_ranks(elapsed, world, dev)

    # --- untimed: check what was timed -----------------------------------------------------------------
    # (i) the integration against the float64 mean of the per-frame rows over the same frames, summed over ranks;
    # (ii) rank 0's sampled frames against the oracle rows of the cpu_baseline leg (N = 1)
    rows_sum = torch.zeros(NCHAN, dtype=torch.complex128, device=dev)
    done = frames if args.no_verify else 0
    while done < frames:                          # the same launches' frames as step(), 2048 rows at a time
        n = min(pool_frames, frames - done)
        for lo in range(0, n, 2048):
            hi = min(n, lo + 2048)
            rows_sum += plan.fx_rows(x[lo:hi], "SPECTRUM")[:, 0].to(torch.complex128).sum(dim=0)
        done += n
    if world > 1:
        sharding.reduce_sums(rows_sum, to_all=True)
    verify = None
    if rank == 0 and not args.no_verify:
        rows_mean = (rows_sum / (frames * world)).cpu().numpy()
        err_rows = float(np.abs(out[0] - rows_mean).max() / np.abs(rows_mean).max())
        err_oracle = {}
        for f, ref in sorted(check_rows.items()):
            if f < pool_frames:
                got = plan.fx_rows(x[f:f + 1], "SPECTRUM")[0, 0].cpu().numpy()
                err_oracle[str(f)] = float(np.abs(got - ref).max() / np.abs(ref).max())
        verify = {"integration_vs_float64_mean_of_rows": err_rows, "rows_vs_oracle": err_oracle, "tolerance": TOL_VIS,
                  "frames": frames * world, "checked_after_timed_region": True}
        assert err_rows < TOL_VIS, verify
        assert all(e < TOL_VIS for e in err_oracle.values()), verify

    others = None
    if world == 1 and not args.no_other_configs:
        others = other_configs(x, dev)
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
                                      "per integration" % world, "dist_backend": args.dist_backend if world > 1 else None,
                       "reduce_transport": integ.transport if comm_note is None else integ.transport + "; " + comm_note,
                       "library": os.path.relpath(_lib.LIB_PATH, ROOT)},
            "roofline": {"bound": "hbm", "kernel": "fx_fused4096_kernel", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_of_copy_ceiling_6290": round(achieved / 6290.0, 4),
                         "bytes_per_launch": int(algo_bytes), "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                         "launches": int(launches),
                         "traffic": None if pmc is None else int(pmc[0] * frames_per_launch),
                         "traffic_source": None if pmc is None else pmc[1]},
            "cpu_baseline": cpu,
            "verify": verify,
            "other_configs": others,
            "power": power,
        }
        print(json.dumps(line))
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
