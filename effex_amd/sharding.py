"""Multi-GPU sharding of integration frames — host-side plumbing.

Chunks are independent (``channelize_poly`` starts every chunk with zero PFB history and the reference
processes chunk pairs one at a time with no carried state, effex/effex.py:391-410), so each rank
integrates a contiguous range of the global chunk index on its own GPU with no data-path collective.
The only exchange is one sum-reduction of the exported accumulators — ``n_baselines*nchan + 1``
complex128 (raw cross-spectra sums + the spectra count; 64 KiB for 2 antennas) — once per integration,
then ``fxc_finalize_sums`` on the root.  Two transports:

* ``RcclComm`` (``fxc_comm_*`` / ``fxc_reduce``, include/fxcorr.h): libfxcorr calls RCCL itself and enqueues
  the ncclReduce on the plan's stream right behind the export — no host synchronisation anywhere between the
  last F+X kernel and the finalize.  ``make_comm`` builds it, handing the unique id round through
  ``torch.distributed``.
* ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests): the collective is issued on
  torch's current stream; a plan that follows that stream (the default) needs no host wait either — the
  collective is stream-ordered behind the export and the finalize behind the collective.  A plan on a stream
  of its own is fenced with ``plan.sync()`` first.

``ShardedRows`` is the other multi-GPU mode, the reference-faithful one: one visibility row per chunk pair, ranks own
disjoint rows of one shared row file, no collective at all.

No scaling curve has been measured yet (the builder has one GPU); the control flow is exercised by
tests/test_dist_gloo.py (world 2, 3 and 8, gloo) and by ``bench.py --dry-run-dist``.
"""


def chunk_range(rank, world_size, n_chunks):
    """Contiguous [lo, hi) of the global chunk index owned by ``rank``."""
    if not 0 <= rank < world_size:
        raise ValueError("rank {} outside world of {}".format(rank, world_size))
    return (rank * n_chunks) // world_size, ((rank + 1) * n_chunks) // world_size


def reduce_sums(sums, root=0, group=None, to_all=False):
    """Sum the exported accumulators across ranks (in place) through ``torch.distributed``.  Returns ``sums``.

    complex128 is reduced through its float64 view so every backend accepts it.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sums
    flat = torch.view_as_real(sums) if sums.is_complex() else sums
    if flat.is_cuda and dist.get_backend(group) == "gloo":      # test set-ups only: gloo reduces host tensors
        host = flat.cpu()
        if to_all:
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce(host, dst=root, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return sums
    if to_all:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(flat, dst=root, op=dist.ReduceOp.SUM, group=group)
    return sums


def make_comm(device, rank, world_size, group=None):
    """An ``RcclComm`` over the ranks of the initialised ``torch.distributed`` group: rank 0 draws the unique id,
    ``broadcast_object_list`` carries it, every rank joins (collective, blocking).  Returns None for one rank."""
    if world_size == 1:
        return None
    import torch.distributed as dist
    from .plan import RcclComm
    box = [None]
    if rank == 0:
        try:
            box[0] = RcclComm.unique_id()
        except Exception as exc:        # librccl not bindable: tell every rank instead of leaving them in the broadcast
            box[0] = exc
    dist.broadcast_object_list(box, src=0, group=group)
    if isinstance(box[0], Exception):
        raise RuntimeError("rank 0 could not draw an RCCL unique id: {}".format(box[0]))
    return RcclComm(device, rank, world_size, box[0])


class ShardedIntegrator(object):
    """One per rank: integrates this rank's chunks on its GPU, then reduces and finalises.

    ``comm``: an ``RcclComm`` -> ``fxc_reduce`` (RCCL called by libfxcorr on the plan's stream); None -> the
    exported sums go through ``torch.distributed`` (``group``)."""

    def __init__(self, plan, rank=0, world_size=1, group=None, comm=None):
        self.plan, self.rank, self.world_size, self.group, self.comm = plan, rank, world_size, group, comm
        self.sums = None if comm is not None else plan.new_sums()
        self._waits = []

    @property
    def transport(self):
        if self.world_size == 1:
            return "none (single rank)"
        return "rccl (fxc_reduce)" if self.comm is not None else "torch.distributed"

    def my_range(self, n_chunks):
        return chunk_range(self.rank, self.world_size, n_chunks)

    def accumulate(self, x_local):
        return self.plan.fx_accumulate(x_local)

    def reduce(self, root=0, to_all=False):
        """Export this rank's accumulator and sum it across ranks; asynchronous on the device."""
        if self.comm is not None:
            self.plan.reduce(self.comm, None if to_all else root)
            return None
        self.plan.acc_export(self.sums)
        if self.world_size > 1:
            if not getattr(self.plan, "_follow", False):
                self.plan.sync()       # a plan on its own stream: fence it before torch's stream takes over
            reduce_sums(self.sums, root=root, group=self.group, to_all=to_all)
        return self.sums

    def finalize(self, mode="SPECTRUM", bandwidth=1.0, root=0, to_all=False):
        """Returns the integrated visibilities on the root (every rank if ``to_all``), else None.  Like the C call it
        refuses while asynchronous results are outstanding: ``finalize_wait`` hands out the oldest one, which would not
        be this integration's."""
        if self._waits:
            raise _state_error("{} asynchronous finalize result(s) outstanding: collect them with finalize_wait() first"
                               .format(len(self._waits)))
        self.finalize_async(mode, bandwidth, root=root, to_all=to_all)
        return self.finalize_wait()

    def finalize_async(self, mode="SPECTRUM", bandwidth=1.0, root=0, to_all=False):
        """Queue reduce + finalize + reset on the device and return: the next integration can be queued before
        ``finalize_wait()`` collects this one (the host never waits between the last F+X kernel of one integration and
        the first of the next).  One rank: a single kernel (``fxc_finalize_async``)."""
        mine = to_all or self.rank == root
        if self.world_size == 1 and hasattr(self.plan, "finalize_async"):
            self.plan.finalize_async(mode, bandwidth, reset=True)
            self._waits.append(True)
            return
        self.reduce(root=root, to_all=to_all)
        if mine:
            if hasattr(self.plan, "finalize_sums_async"):
                self.plan.finalize_sums_async(self.sums, mode, bandwidth)     # sums None: the plan's reduced copy
                self._waits.append(True)
            else:                                        # stand-in plans of the CPU tests
                self._waits.append(self.plan.finalize_sums(self.sums, mode, bandwidth))
        else:
            self._waits.append(None)
        self.plan.acc_reset()

    def finalize_wait(self):
        """The oldest queued integration: the visibilities on the root (every rank if ``to_all``), else None."""
        if not self._waits:
            raise _state_error("no finalize result outstanding")
        item = self._waits.pop(0)
        if item is True:
            return self.plan.finalize_wait()
        return item


def _state_error(message):
    from . import _lib
    return _lib.FxcError(_lib.FXC_ERR_STATE, message)


def batch_range(rank, world_size, n_chunks, batch):
    """Contiguous [lo, hi) of the global chunk index owned by ``rank`` when chunks are dealt in whole batches of ``batch``
    (the last one may be short): every rank's device calls then cover exactly the chunk sets a single rank's would, so the
    rows are the single rank's bit for bit."""
    n_batches = (int(n_chunks) + int(batch) - 1) // int(batch)
    b_lo, b_hi = chunk_range(rank, world_size, n_batches)
    return min(b_lo * batch, n_chunks), min(b_hi * batch, n_chunks)


class ShardedRows(object):
    """The reference-faithful time-series mode over several GPUs (second paragraph): the reference's
    product is one visibility row per chunk pair (``_run_task`` -> the writer, effex/effex.py:402-410, 687-696; read back by
    ``post_process.py:201-219``), chunks are independent, so rank r computes the rows of its own contiguous range of the
    global chunk index and writes them into its own window of ONE shared binary sidecar (``effex_amd.rowsink``):

        rank 0      header line, frequency row, file sized for all rows (``rowsink.create_shared``)      -- barrier --
        every rank  ``fx_rows`` over its range, ``batch`` chunks per device call, rows land in the mapped window
                    (host buffers: the library writes them there; device buffers: one copy)              -- barrier --
        rank 0      publishes the row count (``rowsink.commit_shared``)

    No collective touches the data path; the two barriers go through ``torch.distributed`` (RCCL or gloo).  Ranges are
    whole batches of the global index (``batch_range``), so the file is byte-identical to a single rank's.
    ``tools/rows_to_csv.py`` turns it into the reference's csv."""

    def __init__(self, plan, rank=0, world_size=1, group=None, batch=64):
        if int(batch) < 1:
            raise ValueError("batch must be >= 1")
        self.plan, self.rank, self.world_size, self.group, self.batch = plan, int(rank), int(world_size), group, int(batch)

    def my_range(self, n_chunks):
        return batch_range(self.rank, self.world_size, n_chunks, self.batch)

    def _barrier(self):
        if self.world_size > 1:
            import torch.distributed as dist
            dist.barrier(group=self.group)

    def run(self, path, header, freqs, read_chunks, n_chunks, mode="SPECTRUM", bandwidth=1.0, remove_dc=False, consumer=None):
        """``read_chunks(lo, hi)`` -> the samples of global chunks [lo, hi): a host array or CUDA tensor
        [hi - lo, n_ant, num_samp] complex64 (or uint8 [..., 2]: the receivers' bytes).  Returns (lo, hi) of this rank.

        Device-resident samples: the rows of batch k + 1 are queued before those of batch k are taken off the device's hands --
        the finishing kernel writes them into one of two pinned host slots across PCIe (``FXC_MEM_DEVICE_TO_PINNED``: no copy,
        no wait), and a small pool of threads moves a finished slot into the file (``pwrite``) while the device is two batches on.
        ``consumer(first_chunk, rows)``: called instead of the file writer with each batch in its pinned slot (valid until the
        call returns); then ``path`` may be None and no file is made."""
        import numpy as np
        from . import rowsink
        spectrum = mode.upper() == "SPECTRUM"
        row_len = self.plan.n_baselines * (self.plan.nchan if spectrum else 1)
        dtype = np.complex64 if spectrum else np.complex128
        lo, hi = self.my_range(n_chunks)
        to_file = consumer is None
        if self.rank == 0 and to_file:
            rowsink.create_shared(path, header, freqs if spectrum else None, row_len, dtype, n_chunks)
        self._barrier()
        shape_tail = (self.plan.n_baselines, self.plan.nchan) if spectrum else (self.plan.n_baselines,)
        # every batch is read exactly once (a reader may be sequential -- a file, a socket): the first one says where the samples live
        first = read_chunks(lo, min(hi, lo + self.batch)) if hi > lo else None
        if first is not None and type(first).__module__.startswith("torch"):
            self._rows_from_device(path, lo, hi, read_chunks, mode, bandwidth, remove_dc, shape_tail, dtype, consumer, first)
        elif to_file:
            with rowsink.RowWindow(path, lo, hi) as win:
                for b_lo in range(lo, hi, self.batch):
                    b_hi = min(hi, b_lo + self.batch)
                    x = first if b_lo == lo else read_chunks(b_lo, b_hi)
                    dst = win.rows[b_lo - lo:b_hi - lo].reshape((b_hi - b_lo,) + shape_tail)
                    if str(getattr(x, "dtype", "")).endswith("uint8"):
                        self.plan.fx_rows_u8(x, mode, bandwidth, remove_dc=remove_dc, out=dst)
                    else:
                        self.plan.fx_rows(x, mode, bandwidth, remove_dc=remove_dc, out=dst)
        else:
            for b_lo in range(lo, hi, self.batch):
                b_hi = min(hi, b_lo + self.batch)
                x = first if b_lo == lo else read_chunks(b_lo, b_hi)
                u8 = str(getattr(x, "dtype", "")).endswith("uint8")
                consumer(b_lo, (self.plan.fx_rows_u8 if u8 else self.plan.fx_rows)(x, mode, bandwidth, remove_dc=remove_dc))
        self._barrier()
        if self.rank == 0 and to_file:
            rowsink.commit_shared(path, n_chunks)
        self._barrier()                      # nobody returns (and reads the file) before the count is there
        return lo, hi

    def _rows_marker(self, x):
        """A callable that returns once the rows queued so far are in their pinned slot.  The finishing kernel that writes the slot
        runs on the PLAN's stream: torch's current stream when the plan follows it, else the stream the plan was given
        (``set_stream``) -- or one of its own (``stream="owned"``), which only the plan itself can wait for."""
        import torch
        plan = self.plan
        done = torch.cuda.Event()
        if getattr(plan, "_follow", False):
            done.record(torch.cuda.current_stream(x.device))
            return done.synchronize
        raw = int(getattr(plan, "_stream", -1))
        if raw > 0:
            done.record(torch.cuda.ExternalStream(raw, device=x.device))
            return done.synchronize
        return plan.sync

    def _rows_from_device(self, path, lo, hi, read_chunks, mode, bandwidth, remove_dc, shape_tail, dtype, consumer, first=None):
        import concurrent.futures
        import torch
        from . import rowsink
        from .plan import pinned_empty
        slots = getattr(self, "_slots", None)
        if slots is None or slots[0].shape != (self.batch,) + shape_tail or slots[0].dtype != dtype:
            slots = self._slots = [pinned_empty((self.batch,) + shape_tail, dtype) for _ in range(2)]
        writers = 4
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=writers) if consumer is None else None
        win = rowsink.RowWindow(path, lo, hi, mapped=False) if consumer is None else None
        busy = [[], []]                     # file writes still reading slot s

        def deliver(first, rows, s):
            if consumer is not None:
                consumer(first, rows)
                return
            n = rows.shape[0]
            flat = rows.reshape(n, -1)
            step = (n + writers - 1) // writers
            busy[s] = [pool.submit(win.write, first - lo + r0, flat[r0:min(n, r0 + step)]) for r0 in range(0, n, step)]

        try:
            pending = None
            for k, b_lo in enumerate(range(lo, hi, self.batch)):
                b_hi, s = min(hi, b_lo + self.batch), k & 1
                for job in busy[s]:
                    job.result()            # the file has this slot's previous rows
                busy[s] = []
                x = first if (k == 0 and first is not None) else read_chunks(b_lo, b_hi)
                view = slots[s][:b_hi - b_lo]
                u8 = str(x.dtype).endswith("uint8")
                (self.plan.fx_rows_u8 if u8 else self.plan.fx_rows)(x, mode, bandwidth, remove_dc=remove_dc, out=view)
                done = self._rows_marker(x)
                if pending is not None:     # batch k - 1, while batch k is on the device
                    pending[2]()
                    deliver(pending[0], pending[1], pending[3])
                pending = (b_lo, view, done, s)
            if pending is not None:
                pending[2]()
                deliver(pending[0], pending[1], pending[3])
            for s in (0, 1):
                for job in busy[s]:
                    job.result()
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
            if win is not None:
                win.close()
