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
os.environ.setdefault("FXC_RTC_CACHE", os.path.join(ROOT, "build", "rtc_cache"))      # run-time compiled kernels (other_configs), kept in the tree

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
    pin = args[3] if len(args) > 3 else None  # the logical CPU this worker is bound to (one per physical core), or None
    if pin is not None:
        try:
            os.sched_setaffinity(0, {pin})
        except OSError:
            pass
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
    t0, c0 = time.perf_counter(), time.process_time()
    while True:
        fx_oracle.pfb_xcorr(x[0], x[1], NTAPS, NCHAN, window, BANDWIDTH, FREQUENCY, 0.0, "SPECTRUM", dtype=dtype)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return frames, dt, time.process_time() - c0      # wall seconds and the CPU seconds this process was given in them


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


def cgroup_quota_cores():
    """CPU quota of this process's cgroup in cores (cgroup v2 cpu.max, v1 cfs_quota/period), None = unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        return None if quota == "max" else round(float(quota) / float(period), 2)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            quota, period = float(fq.read()), float(fp.read())
        return None if quota <= 0 else round(quota / period, 2)
    except (OSError, ValueError):
        return None


def physical_core_map(usable):
    """{(package, core): [logical CPUs of it that this process may run on]} from sysfs; one entry per CPU if unreadable."""
    cores = {}
    for cpu in sorted(usable):
        base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
        try:
            with open(base + "physical_package_id") as fh:
                pkg = int(fh.read())
            with open(base + "core_id") as fh:
                core = int(fh.read())
        except (OSError, ValueError):
            pkg, core = -1, cpu
        cores.setdefault((pkg, core), []).append(cpu)
    return cores


def _cpu_pool_arm(ctx, seconds, pins, single):
    """One pool of single-threaded workers on independent frames for `seconds`; pins[k] = the CPU worker k is bound to or None."""
    t_pool = time.perf_counter()
    with ctx.Pool(len(pins)) as pool:
        res = pool.map(_cpu_worker, [(seconds, k, False, pin) for k, pin in enumerate(pins)])
    wall = time.perf_counter() - t_pool
    multi = sum(f * NUM_SAMP / dt for f, dt, _ in res) / 1e6
    worker_seconds = sum(dt for _, dt, _ in res)
    cpu_seconds = sum(c for _, _, c in res)
    eff = cpu_seconds / worker_seconds * len(pins) if worker_seconds > 0 else None     # cores' worth of CPU time obtained
    speedup = multi / single if single > 0 else None
    return {"value": round(multi, 2), "workers": len(pins), "pinned": pins[0] is not None,
            "frames_done": int(sum(f for f, _, _ in res)),
            "worker_seconds_obtained": round(worker_seconds, 1), "cpu_seconds_obtained": round(cpu_seconds, 1),
            "pool_wall_seconds": round(wall, 1),
            # effective_cores: CPU seconds the workers were given per second of their timed loops (scheduler / cgroup quota /
            # oversubscription show up here); speedup_over_single_core: what those cores delivered in units of the single-core
            # rate; their quotient is the rate of one obtained core against a lone worker's (shared memory bandwidth and
            # last-level cache, SMT siblings, all-core clock)
            "effective_cores": round(eff, 1) if eff is not None else None,
            "speedup_over_single_core": round(speedup, 1) if speedup is not None else None,
            "rate_per_obtained_core_vs_single": round(speedup / eff, 3) if speedup and eff else None}


def cpu_baseline(seconds=8.0, frames=FRAMES):
    import multiprocessing as mp
    logical = os.cpu_count() or 1
    try:
        usable_set = set(os.sched_getaffinity(0))
    except Exception:
        usable_set = set(range(logical))
    usable = len(usable_set)
    cores = physical_core_map(usable_set)
    physical = len(cores)
    quota = cgroup_quota_cores()
    frames1, dt1, cpu1 = _cpu_worker((min(seconds, 4.0), 0))
    single = frames1 * NUM_SAMP / dt1 / 1e6
    frames2, dt2, _ = _cpu_worker((2.0, 0, True))
    single_c128 = frames2 * NUM_SAMP / dt2 / 1e6
    ctx = mp.get_context("spawn")
    # two arms over independent frames: one worker per usable logical CPU, and one worker per
    # physical core, each bound to one CPU of its core (skipped when the two are the same set)
    arms = {"per_logical_cpu": _cpu_pool_arm(ctx, seconds, [None] * max(1, usable), single)}
    if physical < usable:
        arms["per_physical_core"] = _cpu_pool_arm(ctx, seconds, [cpus[0] for _, cpus in sorted(cores.items())], single)
    best_name = max(arms, key=lambda k: arms[k]["value"])
    best = arms[best_name]
    check = _oracle_rows([f for f in CHECK_FRAMES if f < frames])
    limits = []
    if quota is not None and quota < best["workers"]:
        limits.append("cgroup quota of %.1f cores" % quota)
    if best["effective_cores"] is not None and best["effective_cores"] < 0.9 * best["workers"]:
        limits.append("the workers were given %.1f cores' worth of CPU time" % best["effective_cores"])
    if best["rate_per_obtained_core_vs_single"] is not None and best["rate_per_obtained_core_vs_single"] < 0.8:
        limits.append("each obtained core ran at %.2f of a lone worker's rate (8 MB of streams and spectra per frame per worker: "
                      "shared memory bandwidth / last-level cache, SMT, all-core clock)" % best["rate_per_obtained_core_vs_single"])
    out = {"value": best["value"], "unit": "Msamples/s", "cores": best["workers"], "kind": "port", "arm": best_name,
           "single_core_value": round(single, 2), "single_core_complex128_value": round(single_c128, 2),
           "single_core_cpu_fraction": round(cpu1 / dt1, 3),
           "cpu_count_logical": logical, "cpu_count_usable": usable, "physical_cores": physical,
           "cgroup_quota_cores": quota,
           "worker_seconds_obtained": best["worker_seconds_obtained"], "cpu_seconds_obtained": best["cpu_seconds_obtained"],
           "pool_wall_seconds": best["pool_wall_seconds"], "frames_done": best["frames_done"],
           "effective_cores": best["effective_cores"], "speedup_over_single_core": best["speedup_over_single_core"],
           "rate_per_obtained_core_vs_single": best["rate_per_obtained_core_vs_single"],
           "limited_by": "; ".join(limits) if limits else None,
           "arms": arms, "cpu_model": cpu_model(),
           "sample": "%d frames of the same workload (S=%d, N=%d, T=%d, 2 ant, complex64 numpy/scipy oracle), "
                     "%d single-threaded worker processes (%s) x %.0f s on independent frames"
                     % (best["frames_done"], NUM_SAMP, NCHAN, NTAPS, best["workers"],
                        "one per physical core, pinned" if best["pinned"] else "one per usable logical CPU", seconds)}
    return out, check


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


def power_sample(step, seconds=4.0, gpu_index=0):
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
        g = r"GPU\[%d\]" % int(gpu_index)
        watts = re.search(g + r"\s*:\s*(?:Current Socket|Average) Graphics Package Power \(W\):\s*([0-9.]+)", text)
        cap = re.search(g + r"[^\n]*?Max Graphics Package Power \(W\):\s*([0-9.]+)", text)
        sclk = re.search(g + r"[^\n]*?sclk clock level:[^\n]*?\((\d+)Mhz\)", text)
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
    if args.supervise and world > 1 and not os.environ.get(CHILD_ENV):
        raise SystemExit(supervise(args, rank, world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.supervise and args.reduce != "torch":     # stands in for the real run's communicator window
        milestone("comm:start")
        _fake_comm_hang()
        milestone("comm:done")
    if args.scaling == "weak":                        # every rank owns `frames` frames
        first_chunk, frames = rank * args.frames, args.frames
    else:                                             # `frames` in all, contiguous ranges
        lo, hi = sharding.chunk_range(rank, world, args.frames)
        first_chunk, frames = lo, hi - lo
    total = args.frames * world if args.scaling == "weak" else args.frames
    plan = _DryPlan()
    integ = sharding.ShardedIntegrator(plan, rank, world)

    def issue():
        plan.fx_accumulate(range(first_chunk, first_chunk + frames))
        integ.finalize_async("SPECTRUM", BANDWIDTH, root=0)

    def run_steps(k):                                 # the pipelined loop of main()
        res = None
        for j in range(k):
            issue()
            if j > 0:
                res = integ.finalize_wait()
        if k > 0:
            res = integ.finalize_wait()
        return res

    def fence():
        if world > 1:
            dist.barrier()

    out = run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    fence()
    elapsed_rank = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_rank, world, torch.device("cpu"))
    ranks = gather_rank_stats({"rank": rank, "frames": frames, "first_frame": first_chunk,
                               "ms_per_step_this_rank": round(elapsed_rank / max(args.steps, 1) * 1e3, 4)}, world)
    if rank == 0:
        want = complex((total - 1) / 2.0, -(total - 1) / 2.0)      # mean of 0 .. total-1
        assert out is not None and np.allclose(out, want), (out, want)
        assert sum(r["frames"] for r in ranks) == total and [r["rank"] for r in ranks] == list(range(world))
        print(json.dumps({"dry_run": True, "backend": "gloo", "n_gpus": world, "steps": args.steps, "scaling": args.scaling,
                          "warmup": args.warmup, "frames_per_rank": [r["frames"] for r in ranks], "frames_total": total,
                          "first_chunk_last_rank": ranks[-1]["first_frame"],
                          "transport": integ.transport, "fallback": os.environ.get(NOTE_ENV),
                          "rccl": {"transport": "torch.distributed", "ranks_seen": None, "rank": None, "ranks_summed": None,
                                   "version": None, "strict": False, "torch_backend": "gloo",
                                   "fallback_reason": os.environ.get(NOTE_ENV) or "dry run on gloo / CPU tensors: no GPU, no RCCL"},
                          "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
                          "ranks": {"per_rank": ranks, "ms_per_step_this_rank": spread(ranks, "ms_per_step_this_rank")},
                          "mean_chunk_index": want.real}))
    else:
        assert out is None
    if world > 1:
        dist.destroy_process_group()


def other_configs(x, dev, reps=5):
    """Short, untimed-region runs of the other single-GPU BASELINE configs on the resident synthetic bytes (viewed with
    their own shapes): configs[2] continuum streaming limit (nchan = 1, num_samp = 2^20) and configs[4] (8 antennas,
    28 baselines, nchan 4096), plus a three-pass and a two-pass shape (32 taps; 8192 channels) and a channel count that is not a power
    of two (1000, 3000, 6000).  HIP events around four calls back to
    back (results collected one call behind, as the headline loop does), per call, median of `reps`."""
    import numpy as np
    from effex_amd.plan import FxPlan, pinned_empty
    out = []
    flat = x.view(-1)

    def run(name, n_ant, nchan, num_samp, n_chunks, window, mode, rows, ntaps=NTAPS):
        need = n_chunks * n_ant * num_samp
        if flat.numel() < need:
            return
        xv = flat[:need].view(n_chunks, n_ant, num_samp)
        with FxPlan(n_ant, nchan, ntaps, num_samp, window=window, device=dev.index) as plan:
            # integrations are delivered into pinned buffers named when they are queued (fxc_finalize_async_to)
            outs = [pinned_empty((plan.n_baselines, nchan) if mode == "SPECTRUM" else (plan.n_baselines,), np.complex128)
                    for _ in range(2)]

            def many(k):
                """k calls back to back; an integration's result is collected while the next one runs, as in main()"""
                for j in range(k):
                    if rows:
                        plan.fx_rows(xv, mode, BANDWIDTH)
                    else:
                        plan.fx_accumulate(xv)
                        plan.finalize_async(mode, BANDWIDTH, reset=True, out=outs[j & 1])
                        if j > 0:
                            plan.finalize_wait()
                if not rows and k > 0:
                    plan.finalize_wait()
            many(2)
            plan.sync()
            ms = []
            for _ in range(reps):
                plan.timer_start()
                many(4)
                ms.append(plan.timer_stop() / 4)
            ms.sort()
            med = ms[len(ms) // 2]
            algo = need * 8
            out.append({"config": name, "path": plan.path, "specialised_per_channel_count": bool(plan.info["specialised"]),
                        "code_object": {0: None, 1: "built by hiprtc", 2: "run-time cache", 3: "pre-built (rtc_prebuilt/)"}.get(plan.info["spec_source"]), "n_ant": n_ant, "nchan": nchan, "ntaps": ntaps, "num_samp": num_samp,
                        "n_chunks": n_chunks, "mode": mode, "median_ms": round(med, 4),
                        "value": round(n_chunks * num_samp / med / 1e3, 1), "unit": "Msamples/s",
                        "algorithmic_GBps": round(algo / med / 1e6, 1), "frac_of_8TBs": round(algo / med / 1e6 / HBM_PEAK_GBS, 4)})

    run("configs[2]: continuum streaming limit, nchan=1, num_samp=2^20, one scalar per chunk pair", 2, 1, 2 ** 20, 2048,
        np.array([0.4, 0.3, 0.2, 0.1]), "CONTINUUM", True)
    run("configs[4]: 8 antennas, 28 baselines, nchan=4096, num_samp=262144, integrated", 8, NCHAN, NUM_SAMP, 512, None,
        "SPECTRUM", False)
    # not BASELINE configs: the reference test's own shape (tests/test_effex.py:62-66) and the largest --nfft of its CLI
    # examples; the first is a three-pass route (3 x the algorithmic traffic by construction), the second two passes
    # (antenna 0's spectra through HBM once: DESIGN.md 6)
    run("reference test shape: 2 antennas, nchan=2048, ntaps=32, num_samp=262144, integrated", 2, 2048, NUM_SAMP, 1024, None,
        "SPECTRUM", False, ntaps=32)
    run("--nfft 8192: 2 antennas, nchan=8192, ntaps=4, num_samp=262144, integrated", 2, 8192, NUM_SAMP, 1024, None,
        "SPECTRUM", False)
    # --resolution is a free integer in the reference (effex.py:733-739): a channel count that is not a power of two
    # (mixed-radix Stockham kernel, F and X in one pass)
    run("--resolution 1000: 2 antennas, nchan=1000, ntaps=4, num_samp=262144, integrated", 2, 1000, NUM_SAMP, 1024, None,
        "SPECTRUM", False)
    # ... and one above 2048 channels (the lean build of the same kernel: taps and first twiddles from L2 tables)
    run("--resolution 3000: 2 antennas, nchan=3000, ntaps=4, num_samp=262144, integrated", 2, 3000, NUM_SAMP, 1024, None,
        "SPECTRUM", False)
    # ... and one above 4096 channels: two passes of the kernels built for the channel count (antenna 0's spectra through HBM once:
    # 2 x the algorithmic traffic by construction)
    run("--resolution 6000: 2 antennas, nchan=6000, ntaps=4, num_samp=262144, integrated", 2, 6000, NUM_SAMP, 1024, None,
        "SPECTRUM", False)
    return out


# ----------------------------------------------------------------------------------------------
# RCCL preflight (N > 1, --reduce auto | rccl): fxc_comm_create / fxc_reduce in a fresh CHILD process per rank, under a
# timeout.  A collective that hangs (one rank failing in ncclCommInitRank leaves the others waiting in theirs) then
# costs the child, not the benchmark: the parent kills it and the run falls back to the torch.distributed transport
# (or exits non-zero with --strict-rccl).  The child is started with subprocess (never an exec of this process, which
# may already hold the GPU) and checks both forms of the reduce against the values every rank can predict.
# ----------------------------------------------------------------------------------------------
def _rccl_child(argv):
    uid_hex, rank, world, gpu = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    import numpy as np
    import torch
    from effex_amd import synth
    from effex_amd.plan import FxPlan, RcclComm
    num_samp, n_chunks = NCHAN * 4, 3
    x = torch.from_numpy(synth.synth_iq(SEED, n_chunks, 2, num_samp)).cuda(gpu)     # the same frames on every rank
    with FxPlan(N_ANT, NCHAN, NTAPS, num_samp, device=gpu) as plan, RcclComm(gpu, rank, world, bytes.fromhex(uid_hex)) as comm:
        plan.fx_accumulate(x)
        one = plan.finalize_sums(plan.acc_export(plan.new_sums()), "SPECTRUM")      # this rank's own integration
        for root in (None, 0, world - 1):                # ncclAllReduce, ncclReduce to the first and to the last rank
            plan.reduce(comm, root)
            if root is None or root == rank:
                got = plan.finalize_sums(None, "SPECTRUM")      # sums and spectra counts both scale by `world`
                err = float(np.abs(got - one).max() / np.abs(one).max())
                if not err < 1e-12:
                    print("rccl preflight: rank %d root %s: reduced result off by %.3g" % (rank, root, err))
                    return 4
        plan.sync()
        info, summed = comm.info(), comm.probe()          # what the communicator says of itself; an all-reduce of ones
        if not (info["ranks_seen"] in (world, None) and info["rank_seen"] in (rank, None) and summed == world):
            print("rccl preflight: rank %d: the communicator reports %s, %d ranks summed" % (rank, info, summed))
            return 5
    print("rccl preflight: rank %d of %d on GPU %d ok %s" % (rank, world, gpu, json.dumps(
        {"ranks_seen": info["ranks_seen"], "rank": info["rank_seen"], "device_seen": info["device_seen"], "ranks_summed": summed,
         "version": info["rccl_version"], "reduces": info["reduces"]})))
    return 0


def rccl_preflight(gpu, rank, world, timeout_s):
    """-> (ok on every rank, note).  Collective over torch.distributed (already initialised)."""
    import subprocess
    import torch
    import torch.distributed as dist
    from effex_amd.plan import RcclComm
    box = [None]
    if rank == 0:
        try:
            box[0] = RcclComm.unique_id().hex()
        except Exception as exc:
            box[0] = "error: %s" % exc
    dist.broadcast_object_list(box, src=0)
    note, ok = None, True
    if box[0].startswith("error"):
        ok, note = False, box[0]
    else:
        cmd = [sys.executable, os.path.abspath(__file__), "--rccl-child", box[0], str(rank), str(world), str(gpu)]
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        try:
            out, _ = proc.communicate(timeout=timeout_s)
            if proc.returncode != 0:
                ok, note = False, "rank %d: child exited %s: %s" % (rank, proc.returncode, (out or "").strip()[-300:])
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.communicate()
            ok, note = False, "rank %d: no answer within %.0f s (killed)" % (rank, timeout_s)
    notes = [None] * world
    dist.all_gather_object(notes, note)
    bad = [n for n in notes if n]
    return (not bad), ("; ".join(bad) if bad else None)


# ----------------------------------------------------------------------------------------------
# Watchdog around the REAL communicator (N > 1, fxc_reduce): the preflight above proves in a throw-away child that
# fxc_comm_create / fxc_reduce work; the measuring process then has to call ncclCommInitRank itself, and a collective
# that hangs there cannot be abandoned from inside the process that is stuck in it.  So with N > 1 every rank started by
# the launcher is a *supervisor* that never touches the GPU: it runs the measurement in a child process (a fresh child,
# never an exec of a process that holds the GPU), watches the child's milestones ("comm:start" / "comm:done" in a status
# file), and if the communicator is not up within --comm-timeout seconds -- or the child dies inside that window -- kills
# exactly that child and starts another one on the torch.distributed transport (--reduce torch) with a rendezvous of its
# own (a fresh port published by rank 0's supervisor through a file; the launcher's store still holds the first
# attempt's keys).  The fall-back is recorded in the line (config.reduce_transport).
# ----------------------------------------------------------------------------------------------
STATUS_ENV, CHILD_ENV, NOTE_ENV = "FXC_BENCH_STATUS", "FXC_BENCH_CHILD", "FXC_BENCH_FALLBACK_NOTE"


def milestone(text):
    path = os.environ.get(STATUS_ENV)
    if path:
        with open(path, "a") as fh:
            fh.write("%s %.3f\n" % (text, time.time()))


def _fake_comm_hang():
    """Test hook (tests/test_dist_gloo.py): the first attempt never gets its communicator -- on every rank ("1") or on
    one of them ("rank:K"), the others then wait for it in their first collective."""
    hook = os.environ.get("FXC_BENCH_TEST_COMM_HANG")
    if hook and not os.environ.get(NOTE_ENV) and hook in ("1", "rank:%s" % os.environ.get("RANK", "0")):
        time.sleep(3600)


def one_node_launch(world):
    """True when every rank of the launch runs on this node (torch.distributed.run exports LOCAL_WORLD_SIZE)."""
    return int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) == world


def strict_rccl(args):
    """fxc_reduce or nothing: the default for a launch over the nccl backend (one GPU per rank); the torch.distributed
    transport takes over only when --allow-torch-fallback says so (or the transport was asked for with --reduce torch)."""
    if args.reduce == "torch":
        return False
    if args.strict_rccl:
        return True
    return args.dist_backend == "nccl" and not args.allow_torch_fallback


def supervise(args, rank, world):
    """One per rank, never touches the GPU.  Decisions are all-or-nothing over the ranks of the node: the supervisor that
    sees its child fail (or hang) inside the communicator window publishes a verdict file -- "abort" (strict, the default)
    or "fallback" -- that every supervisor polls; on "abort" all kill their children and exit non-zero, on "fallback" all
    kill their children and start the torch.distributed attempt on a rendezvous rank 0 publishes.  The files carry the
    launcher's pid and port in their names, and rank 0 clears stale ones before the first attempt."""
    import signal
    import socket
    import subprocess
    import tempfile
    strict = strict_rccl(args)
    if not one_node_launch(world):
        strict = True                      # the file rendezvous below is one node's: no fall-back across nodes
    t_start = time.time()
    base = os.path.join(tempfile.gettempdir(), "fxbench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
    port_file, verdict_file = base + ".port", base + ".verdict"
    if rank == 0:
        for stale in (port_file, verdict_file):
            try:
                os.unlink(stale)
            except OSError:
                pass
    status = tempfile.NamedTemporaryFile(prefix="fxbench_status_r%d_" % rank, suffix=".txt", delete=False)
    status.close()
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]      # (the child knows it is one: CHILD_ENV)
    live = [None]

    def start(extra_args, extra_env):
        env = dict(os.environ, **{CHILD_ENV: "1", STATUS_ENV: status.name})
        env.update(extra_env)
        open(status.name, "w").close()
        live[0] = subprocess.Popen(argv + extra_args, env=env)
        return live[0]

    def on_term(signum, _frame):           # the launcher ends its ranks with SIGTERM: the measuring child goes with us
        if live[0] is not None and live[0].poll() is None:
            live[0].kill()
        raise SystemExit(128 + signum)
    signal.signal(signal.SIGTERM, on_term)

    def state():
        with open(status.name) as fh:
            text = fh.read()
        begun = [ln for ln in text.splitlines() if ln.startswith("comm:start")]
        return (float(begun[-1].split()[1]) if begun else None), "comm:done" in text

    def verdict():
        """(kind, why) another supervisor (or this one) published during THIS launch, else None."""
        try:
            if os.path.getmtime(verdict_file) < t_start - 5.0:
                return None
            with open(verdict_file) as fh:
                kind, _, why = fh.read().partition(": ")
            return (kind, why) if kind in ("abort", "fallback") else None
        except OSError:
            return None

    def publish(kind, why):
        try:
            fd = os.open(verdict_file, os.O_WRONLY | os.O_CREAT | os.O_EXCL)     # the first verdict stands
            with os.fdopen(fd, "w") as fh:
                fh.write("%s: %s" % (kind, why))
        except FileExistsError:
            pass
        return verdict() or (kind, why)

    proc = start([], {})
    decided = None
    while decided is None:
        rc = proc.poll()
        begun, done = state()
        seen = verdict()
        if seen is not None:                                   # another rank's supervisor decided for everyone
            decided = seen
        elif rc is not None:
            if rc != 0 and begun is not None and not done:
                decided = publish("abort" if strict else "fallback", "the measuring process of rank %d exited %s while the "
                                  "RCCL communicator was being made" % (rank, rc))
            else:
                os.unlink(status.name)
                return rc
        elif begun is not None and not done and time.time() - begun > args.comm_timeout:
            decided = publish("abort" if strict else "fallback", "no RCCL communicator on rank %d within %.0f s"
                              % (rank, args.comm_timeout))
        else:
            time.sleep(0.2)
    kind, why = decided
    if proc.poll() is None:
        proc.kill()                          # exactly the child this supervisor started
    proc.wait()
    if kind == "abort":
        os.unlink(status.name)
        sys.stderr.write("bench.py rank %d: %s -- fxc_reduce is required (pass --allow-torch-fallback to take the "
                         "torch.distributed transport instead)\n" % (rank, why))
        return 3
    # fall back: a fresh child on the torch.distributed transport, with a rendezvous the first attempt never touched
    if rank == 0:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        with open(port_file + ".tmp", "w") as fh:
            fh.write(str(port))
        os.replace(port_file + ".tmp", port_file)
    else:
        t_end = time.time() + args.comm_timeout + 120
        while not os.path.exists(port_file):
            if time.time() > t_end:
                raise SystemExit("bench.py: rank %d fell back (%s) but rank 0 never published a rendezvous port" % (rank, why))
            time.sleep(0.2)
        port = int(open(port_file).read())
    sys.stderr.write("bench.py rank %d: %s -- falling back to --reduce torch on port %d\n" % (rank, why, port))
    proc = start(["--reduce", "torch"], {NOTE_ENV: why, "MASTER_PORT": str(port), "TORCHELASTIC_USE_AGENT_STORE": "False"})
    rc = proc.wait()
    os.unlink(status.name)
    if rank == 0:
        for used in (port_file, verdict_file):
            if os.path.exists(used):
                os.unlink(used)
    return rc


def rccl_evidence(comm, comm_note, strict, world, backend):
    """This rank's part of the line's "rccl" object: what the communicator fxc_comm_create made says about itself
    (fxc_comm_info: ncclCommCount / ncclCommUserRank / ncclCommCuDevice asked of the live ncclComm_t) and the count RCCL
    itself added up in one all-reduce of ones (fxc_comm_probe, collective).  Without a communicator: nulls and the reason."""
    from effex_amd.plan import RcclComm
    ev = {"transport": "fxc_reduce" if comm is not None else ("torch.distributed" if world > 1 else "none"),
          "ranks_seen": None, "rank": None, "device_seen": None, "ranks_summed": None, "version": None, "library": None,
          "async_error": None, "fallback_reason": comm_note, "strict": bool(strict), "torch_backend": backend if world > 1 else None}
    try:
        ev["version"], ev["library"] = RcclComm.library()
    except Exception as exc:
        ev["library"] = "not bound: %s" % exc
    if comm is not None:
        info = comm.info()
        ev.update(ranks_seen=info["ranks_seen"], rank=info["rank_seen"], device_seen=info["device_seen"],
                  version=info["rccl_version"] or ev["version"], async_error=info["async_error"])
        ev["ranks_summed"] = comm.probe()
    return ev


def gather_rank_stats(stats, world):
    """Every rank's dict on rank 0 (list indexed by rank)."""
    if world == 1:
        return [stats]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, stats)
    return out


def spread(rows, key):
    vals = [r[key] for r in rows if r.get(key) is not None]
    if not vals:
        return None
    return {"min": min(vals), "max": max(vals), "mean": round(sum(vals) / len(vals), 4), "argmax_rank": max(range(len(vals)), key=lambda i: vals[i])}


def rows_mode(args, plan, x, dev, rank, world, first_frame, frames, total_frames, check_rows):
    """--rows: K timed passes; in each, every rank turns its resident frames into visibility rows (fxc_fx_rows, --rows-batch
    frames per call) and writes them into its own window of one shared row file -- the reference's product, one row per
    chunk pair (effex.py:402-410, 687-696), with no collective on the data path.  Barrier + synchronize on both sides of the
    timed region, max over ranks; afterwards (untimed) rank 0 reads the file back and checks it."""
    import tempfile
    import numpy as np
    import torch
    import torch.distributed as dist
    from effex_amd import rowsink, sharding
    box = [None]
    if rank == 0:
        box[0] = args.rows_file or os.path.join(tempfile.mkdtemp(prefix="fxrows_"), "visibilities.fxb")
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    path = box[0]
    header = rowsink.header_line(1, BANDWIDTH, FREQUENCY, NUM_SAMP, NCHAN, 49.6, "SPECTRUM")
    freqs = rowsink.spectrum_freqs(NCHAN, BANDWIDTH, FREQUENCY)
    writer = sharding.ShardedRows(plan, rank, world, batch=args.rows_batch)
    assert writer.my_range(total_frames) == (first_frame, first_frame + frames)
    to_file = args.rows_sink == "file"
    probe = sorted(set([0, frames // 2, frames - 1])) if frames > 0 else []
    kept = {}                                         # --rows-sink pinned: copies of the probe rows, taken as the batches go by

    def read_chunks(lo, hi):
        return x[lo - first_frame:hi - first_frame]

    def consumer(first, rows):                        # the batch sits in a pinned slot until this returns
        for f in probe:
            if first <= first_frame + f < first + rows.shape[0]:
                kept[f] = np.array(rows[first_frame + f - first, 0])

    def one_pass():
        writer.run(path if to_file else None, header, freqs, read_chunks, total_frames, "SPECTRUM", BANDWIDTH,
                   consumer=None if to_file else consumer)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_pass()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    fence()
    elapsed_rank = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_rank, world, dev)
    ranks = gather_rank_stats({"rank": rank, "frames": frames, "first_frame": first_frame,
                               "ms_per_step_this_rank": round(elapsed_rank / max(args.steps, 1) * 1e3, 4),
                               "rows_per_s_this_rank": round(frames * args.steps / elapsed_rank, 1)}, world)
    if rank == 0:
        direct = plan.fx_rows(x[probe], "SPECTRUM")[:, 0].cpu().numpy() if probe else None
        if to_file:
            back = rowsink.RowFile(path)
            assert back.rows.shape == (total_frames, NCHAN) and back.header == header, (back.rows.shape, back.header)
            got = {f: np.asarray(back.rows[first_frame + f]) for f in probe}
            for f in check_rows:
                if f < frames:
                    got[f] = np.asarray(back.rows[f])
            assert np.abs(np.asarray(back.rows[-1])).max() > 0          # the last rank's last row arrived
            rows_in_file = int(back.rows.shape[0])
            del back
        else:
            got, rows_in_file = dict(kept), None
            assert sorted(got) == probe, (sorted(got), probe)
        assert sum(r["frames"] for r in ranks) == total_frames
        # sampled rows against a direct call on the same frames, and rank 0's against the oracle (N = 1)
        err_direct = max([float(np.abs(got[f] - direct[i]).max() / np.abs(direct[i]).max()) for i, f in enumerate(probe)] or [0.0])
        err_oracle = {str(f): float(np.abs(got[f] - ref).max() / np.abs(ref).max())
                      for f, ref in sorted(check_rows.items()) if f in got}
        assert err_direct < TOL_VIS and all(e < TOL_VIS for e in err_oracle.values()), (err_direct, err_oracle)
        rows_s = total_frames * args.steps / elapsed
        print(json.dumps({
            "metric": "2-ant FX correlator time series (PFB+FFT+X, one visibility row per frame into %s)"
                      % ("one shared row file" if to_file else "pinned host memory"),
            "value": round(rows_s, 1), "unit": "rows/s", "Msamples_per_s": round(rows_s * NUM_SAMP / 1e6, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1] frames as a time series: 2-antenna FX, num_samp=262144, ntaps=4, nchan=4096, "
                                   "%d frames %s, one complex64 row of 4096 bins per frame" % (args.frames, "per GPU" if args.scaling == "weak" else "in all"),
                       "frames_total": total_frames, "rows_batch": args.rows_batch, "row_bytes": NCHAN * 8, "rows_sink": args.rows_sink,
                       "delivery": "rows written by the finishing kernel into two pinned host slots across PCIe (FXC_MEM_DEVICE_TO_PINNED), "
                                   "batch k + 1 queued before batch k is collected" + ("; four threads pwrite a finished slot into the file" if to_file else ""),
                       "parallelism": "frames sharded over %d GPU(s) in whole batches, disjoint windows of one row file, no collective" % world,
                       "dist_backend": args.dist_backend if world > 1 else None, "path": plan.path},
            "verify": {"rows_in_file": rows_in_file, "file_vs_direct_call": err_direct, "rows_vs_oracle": err_oracle,
                       "tolerance": TOL_VIS, "checked_after_timed_region": True},
            "ranks": {"per_rank": ranks, "rows_per_s_this_rank": spread(ranks, "rows_per_s_this_rank"),
                      "ms_per_step_this_rank": spread(ranks, "ms_per_step_this_rank")}}))
        if to_file and not args.rows_file:
            import shutil
            shutil.rmtree(os.path.dirname(path), ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--rccl-child":
        raise SystemExit(_rccl_child(sys.argv[2:]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps first; after idle the clock needs about five launches to settle")
    ap.add_argument("--frames", type=int, default=FRAMES,
                    help="integration frames per step: per GPU (--scaling weak) or in all (--scaling strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --frames per GPU (SURVEY.md 8d config 4, the default the driver runs); strong: --frames in "
                         "all, rank r integrates the contiguous range sharding.chunk_range(r, N, frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power sample after the timed region")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of configs[2], configs[4] and the two three-pass shapes")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the check after the timed region (profiling runs: its fx_rows launches would mix into "
                         "the per-kernel statistics)")
    ap.add_argument("--reduce", choices=("auto", "rccl", "torch"), default="auto",
                    help="N > 1: the cross-rank reduce.  rccl: fxc_reduce (libfxcorr calls RCCL on the plan's stream); torch: "
                         "torch.distributed (backend nccl = RCCL); auto: rccl if its preflight passes on every rank, else torch")
    ap.add_argument("--strict-rccl", action="store_true",
                    help="exit non-zero instead of falling back to torch.distributed when fxc_reduce cannot be used: the "
                         "default with --dist-backend nccl (one GPU per rank)")
    ap.add_argument("--allow-torch-fallback", action="store_true",
                    help="N > 1: if fxc_reduce cannot be used (preflight fails, no communicator within --comm-timeout) take the "
                         "torch.distributed transport instead of exiting non-zero; the line says so in rccl.fallback_reason")
    ap.add_argument("--rccl-timeout", type=float, default=120.0, help="seconds the RCCL preflight child of a rank may take")
    ap.add_argument("--comm-timeout", type=float, default=180.0,
                    help="N > 1: seconds the measuring process may spend making the real RCCL communicator before its supervisor "
                         "kills it and falls back to --reduce torch from a fresh child")
    ap.add_argument("--supervise", action="store_true", help="(tests) run the supervisor in --dry-run-dist mode too")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend for N > 1.  gloo + fewer GPUs than ranks (ranks share GPUs round-robin) "
                         "runs the real multi-rank flow on a one-GPU box for testing; its timing means nothing")
    ap.add_argument("--rows", action="store_true",
                    help="the reference-faithful time-series mode instead of the integration (SURVEY.md 8e, second paragraph): one "
                         "visibility row per frame, every rank writes the rows of its own frames into its window of ONE shared "
                         "row file (effex_amd.sharding.ShardedRows), no collective; prints rows/s")
    ap.add_argument("--rows-batch", type=int, default=512, help="--rows: frames per device call")
    ap.add_argument("--rows-sink", choices=("file", "pinned"), default="file",
                    help="--rows: where the rows end up -- the shared row file (the reference's product), or pinned host memory only "
                         "(a consumer's buffer: what the device side delivers with the file system out of the way)")
    ap.add_argument("--rows-file", default=None, help="--rows: the shared row file (default: a temporary file, removed)")
    ap.add_argument("--dry-run-dist", action="store_true",
                    help="run the multi-rank control flow on gloo / CPU tensors with a stand-in plan (no GPU)")
    args = ap.parse_args()

    if args.dry_run_dist:
        if args.frames == FRAMES:
            args.frames = 100
        return dry_run_dist(args)

    rank, local_rank, world = rank_env(args)
    if (world > 1 and args.reduce != "torch" and args.dist_backend == "nccl" and not args.rows
            and not os.environ.get(CHILD_ENV)):
        raise SystemExit(supervise(args, rank, world))      # this process never touches the GPU

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

    # --- this rank's frames of the synthetic stream, device resident if they fit ------------------------
    total_frames = args.frames * world if args.scaling == "weak" else args.frames
    if args.rows:      # whole batches of the global frame index per rank: the file is a single rank's, byte for byte
        lo, hi = sharding.batch_range(rank, world, total_frames, args.rows_batch)
        first_frame, frames = lo, hi - lo
    elif args.scaling == "weak":
        first_frame, frames = rank * args.frames, args.frames
    else:
        lo, hi = sharding.chunk_range(rank, world, args.frames)
        first_frame, frames = lo, hi - lo
    if frames < 1 and not args.rows:      # (--rows: a rank without a batch writes an empty window and joins the barriers)
        raise SystemExit("rank %d has no frames: --frames %d over %d ranks" % (rank, args.frames, world))
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    need = frames * BYTES_PER_FRAME
    pool_frames = frames if need < 0.8 * free_b else max(512, int(0.5 * free_b // BYTES_PER_FRAME))
    x = torch.empty((pool_frames, N_ANT, NUM_SAMP), dtype=torch.complex64, device=dev)
    synth_fill(x, SEED, first_chunk=first_frame)
    torch.cuda.synchronize(dev)

    plan = FxPlan(N_ANT, NCHAN, NTAPS, NUM_SAMP, device=gpu)
    assert plan.path == "fused", "headline workload must run on the fused HIP kernel"
    plan.set_delay(BANDWIDTH, FREQUENCY, 0.0)
    if args.rows:
        if pool_frames < frames:
            raise SystemExit("--rows needs this rank's %d frames resident (%d fit)" % (frames, pool_frames))
        return rows_mode(args, plan, x, dev, rank, world, first_frame, frames, total_frames, check_rows)
    comm, comm_note = None, None
    if world > 1 and args.reduce != "torch":
        if args.dist_backend != "nccl":
            comm_note = "ranks share GPUs in the gloo test set-up: RCCL wants one GPU per rank"
        else:
            ok, why = rccl_preflight(gpu, rank, world, args.rccl_timeout)
            if ok:
                milestone("comm:start")          # from here the supervisor's clock runs (--comm-timeout)
                try:
                    _fake_comm_hang()
                    comm = sharding.make_comm(gpu, rank, world)
                except Exception as exc:
                    comm, why = None, "rank %d: %s" % (rank, exc)
                flag = torch.tensor([1 if comm is not None else 0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)           # all ranks or none
                if int(flag.item()) == 0 and comm is not None:
                    comm.close()
                    comm = None
                milestone("comm:done")
            if comm is None:
                comm_note = "no RCCL communicator (%s)" % (why or "another rank failed")
        if comm is None and strict_rccl(args):
            raise SystemExit("bench.py: fxc_reduce is not usable: %s (--allow-torch-fallback takes the torch.distributed "
                             "transport instead)" % comm_note)
    if os.environ.get(NOTE_ENV):
        comm_note = "fell back from fxc_reduce: " + os.environ[NOTE_ENV]
    integ = sharding.ShardedIntegrator(plan, rank, world, comm=comm)
    rccl = None
    if world > 1:
        why_not = comm_note or ("the torch.distributed transport was asked for (--reduce torch)" if comm is None else None)
        rccl = rccl_evidence(comm, why_not, strict_rccl(args), world, args.dist_backend)

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
    elapsed_rank = time.perf_counter() - t0
    kernel_ms, launches = plan.kernel_time(reset=True)
    plan.kernel_profiling(False)
    elapsed = max_over_ranks(elapsed_rank, world, dev)

    # --- untimed: latency of the reduce alone (export + collective, HIP events on the plan's stream) -----
    reduce_us = None
    if world > 1:
        plan.fx_accumulate(x[:min(pool_frames, 64)])
        lat = []
        for _ in range(12):
            fence()
            plan.timer_start()
            integ.reduce(root=0)
            lat.append(plan.timer_stop() * 1e3)
        plan.acc_reset()
        lat.sort()
        reduce_us = round(lat[len(lat) // 2], 1)

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
        rows_mean = (rows_sum / total_frames).cpu().numpy()
        err_rows = float(np.abs(out[0] - rows_mean).max() / np.abs(rows_mean).max())
        err_oracle = {}
        for f, ref in sorted(check_rows.items()):
            if f < pool_frames:
                got = plan.fx_rows(x[f:f + 1], "SPECTRUM")[0, 0].cpu().numpy()
                err_oracle[str(f)] = float(np.abs(got - ref).max() / np.abs(ref).max())
        verify = {"integration_vs_float64_mean_of_rows": err_rows, "rows_vs_oracle": err_oracle, "tolerance": TOL_VIS,
                  "frames": total_frames, "checked_after_timed_region": True}
        assert err_rows < TOL_VIS, verify
        assert all(e < TOL_VIS for e in err_oracle.values()), verify

    others = None
    if world == 1 and not args.no_other_configs:
        others = other_configs(x, dev)

    def busy():                                    # keeps this rank's GPU on the F+X kernel, no collective
        plan.fx_accumulate(x[:min(pool_frames, frames)])
        plan.acc_reset()
        plan.sync()

    power = power_sample(busy, gpu_index=gpu) if not args.no_power else None

    # --- every rank's own numbers, for the reader of an efficiency < 1 ------------------------------------
    my = {"rank": rank, "gpu": gpu, "frames": frames, "first_frame": first_frame,
          "avg_kernel_ms": round(kernel_ms / max(launches, 1), 4), "launches": int(launches),
          "ms_per_step_this_rank": round(elapsed_rank / max(args.steps, 1) * 1e3, 4),
          "sclk_mhz": power["sclk_mhz"] if power else None, "package_w": power["package_w"] if power else None,
          "reduce_us": reduce_us}
    if rccl is not None:
        my["rccl"] = {k: rccl[k] for k in ("ranks_seen", "rank", "device_seen", "ranks_summed", "async_error")}
        my["rccl"]["reduces_queued"] = comm.info()["reduces"] if comm is not None else 0
    ranks = gather_rank_stats(my, world)

    if rank == 0:
        assert out is not None and np.isfinite(out).all() and np.abs(out).max() > 0
        samples = float(total_frames) * NUM_SAMP * args.steps
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 2-antenna FX, num_samp=262144, ntaps=4, nchan=4096, "
                                   "%d integration frames %s per step" % (args.frames, "per GPU" if args.scaling == "weak" else "in all"),
                       "frames_per_gpu": frames, "frames_total": total_frames, "resident_frames": pool_frames,
                       "num_samp": NUM_SAMP, "nchan": NCHAN, "ntaps": NTAPS, "n_ant": N_ANT, "path": plan.path,
                       "sample_definition": "one complex sample per antenna stream",
                       "parallelism": "frames sharded over %d GPU(s), one RCCL reduce of the cross-spectra "
                                      "per integration" % world, "dist_backend": args.dist_backend if world > 1 else None,
                       "reduce_transport": integ.transport if comm_note is None else integ.transport + "; " + comm_note,
                       "steps_pipelined": "integration j + 1 is queued before the host collects the result of j "
                                          "(fxc_finalize_async / fxc_finalize_wait); all %d results are collected inside the "
                                          "timed region" % args.steps,
                       "library": os.path.relpath(_lib.LIB_PATH, ROOT)},
            "roofline": {"bound": "hbm", "kernel": "fx_fused4096_kernel", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_of_copy_ceiling_6290": round(achieved / 6290.0, 4),
                         "bytes_per_launch": int(algo_bytes), "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                         "launches": int(launches),
                         "traffic": None if pmc is None else int(pmc[0] * frames_per_launch),
                         "traffic_source": None if pmc is None else pmc[1],
                         "rank": 0},
            "cpu_baseline": cpu,
            "verify": verify,
            "other_configs": others,
            "power": power,
        }
        if world > 1:
            # the proof of the ranks RCCL saw, from the communicator itself (rank 0's answers; every rank's are in
            # ranks.per_rank[*].rccl): ranks_seen = ncclCommCount, ranks_summed = an all-reduce of ones
            per = [r.get("rccl") or {} for r in ranks]
            rccl["reduces_queued"] = per[0].get("reduces_queued")
            rccl["all_ranks_agree"] = (comm is not None and all(q.get("ranks_seen") == world and q.get("ranks_summed") == world
                                                                and q.get("rank") == k for k, q in enumerate(per)))
            line["rccl"] = rccl
            # what an efficiency below 1 is made of: the slowest rank's kernel (clock under the power cap differs from
            # GPU to GPU), the reduce, and what is left of the step outside the kernel
            line["ranks"] = {"per_rank": ranks,
                             "avg_kernel_ms": spread(ranks, "avg_kernel_ms"), "sclk_mhz": spread(ranks, "sclk_mhz"),
                             "package_w": spread(ranks, "package_w"), "reduce_us": spread(ranks, "reduce_us"),
                             "ms_per_step_this_rank": spread(ranks, "ms_per_step_this_rank"),
                             "roofline_frac": spread([{"f": BYTES_PER_FRAME * r["frames"] / (r["avg_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                       if r["avg_kernel_ms"] else None} for r in ranks], "f"),
                             "step_minus_slowest_kernel_ms": round(elapsed / args.steps * 1e3 - max(r["avg_kernel_ms"] for r in ranks), 4)}
        print(json.dumps(line))
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
