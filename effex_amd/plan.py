"""FxPlan — thin Python handle on an ``fxc_plan`` (include/fxcorr.h).

Buffers are either device-resident ``torch`` complex64 tensors (passed by ``data_ptr()``; PyTorch is
only the allocator / stream / collective plumbing) or host numpy complex64 arrays (the library stages
them itself).  Outputs come back in the same kind as the input.
"""
import ctypes

import numpy as np

from . import _lib
from .window import design_window

MODES = {"SPECTRUM": _lib.FXC_MODE_SPECTRUM, "CONTINUUM": _lib.FXC_MODE_CONTINUUM, "TEST": _lib.FXC_MODE_CONTINUUM}
PATHS = {None: -1, "auto": -1, "generic": _lib.FXC_PATH_GENERIC, "fused": _lib.FXC_PATH_FUSED,
         "stream": _lib.FXC_PATH_STREAM, "tiled": _lib.FXC_PATH_TILED}
PATH_NAMES = {_lib.FXC_PATH_GENERIC: "generic", _lib.FXC_PATH_FUSED: "fused", _lib.FXC_PATH_STREAM: "stream",
              _lib.FXC_PATH_TILED: "tiled"}


def _current_torch_stream(device):
    """torch's current HIP stream on ``device`` (0 = the default stream) or 0 without torch/GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.current_stream(device).cuda_stream)
    except ImportError:
        pass
    return 0


def _is_torch(x):
    return type(x).__module__.startswith("torch")


IQ_FORMATS = {"c64": _lib.FXC_IQ_C64, "u8": _lib.FXC_IQ_U8, "c128": _lib.FXC_IQ_C128}
_IQ_DTYPES = {"c64": np.complex64, "u8": np.uint8, "c128": np.complex128}


def pinned_empty(shape, dtype):
    """A numpy array in pinned host memory from ``fxc_host_alloc`` -- the counterpart of the reference's
    ``cusignal.get_shared_mem`` staging buffers (effex.py:109-110): host-buffer calls copy from it by direct DMA, and
    visibility rows asked for into such an array are written by the device itself.  Freed when the last view dies."""
    import weakref
    lib = _lib.load()
    shape = tuple(int(v) for v in (shape if np.ndim(shape) else (shape,)))
    dtype = np.dtype(dtype)
    n_bytes = max(1, int(np.prod(shape)) * dtype.itemsize)
    ptr = ctypes.c_void_p()
    _lib.check(lib.fxc_host_alloc(ctypes.byref(ptr), n_bytes), None)
    buf = (ctypes.c_char * n_bytes).from_address(ptr.value)
    fin = weakref.finalize(buf, lib.fxc_host_free, ctypes.c_void_p(ptr.value))
    fin.atexit = False          # at interpreter exit the HIP runtime may already be gone; the OS takes the pages back
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def rot_table(nbins, bandwidth, frequency, calibrated_delay):
    """rot[k] = exp(+2 pi i f_k tau), natural bin order, complex128 — effex/effex.py:516,519.

    Formed in float64 on the host: the phase is ~9e3 rad for tau = 1 us.
    """
    freqs = np.fft.fftfreq(nbins, d=1.0 / bandwidth) + frequency
    return np.exp(2j * np.pi * freqs * calibrated_delay)


class FxPlan(object):
    def __init__(self, n_ant, nchan, ntaps, num_samp, window=None, device=0, stream=None, path=None, dev=False):
        self._lib = _lib.load(dev=dev)          # dev: the developer build with the reference kernels (tests, tools/soak.py)
        self._h = ctypes.c_void_p()
        if window is None:
            window = design_window(ntaps, nchan)
        window = np.ascontiguousarray(window, dtype=np.float64)
        if window.shape != (int(ntaps) * int(nchan),):
            raise ValueError("window must have ntaps*nchan = {} taps, got {}".format(ntaps * nchan, window.shape))
        self.window = window
        self._pipes = []                                  # weak references to FxPipeline objects on this plan
        self._queued = []                                 # modes of the finalize results queued and not yet collected
        # Work is issued on the caller's stream so it is ordered with torch's own copies / kernels.  stream=None
        # follows torch's *current* stream call by call (``_follow``): under ``with torch.cuda.stream(s)`` the plan moves
        # to ``s`` (fxc_set_stream orders the two streams with an event), so inputs, kernels and outputs stay ordered.
        self._follow = stream is None
        if stream is None:
            stream = _current_torch_stream(int(device))
        elif stream == "owned":
            stream = -1                                   # FXC_STREAM_OWNED
        self._stream = int(stream)
        stream_ptr = ctypes.c_void_p(int(stream) & 0xFFFFFFFFFFFFFFFF) if stream else None
        rc = self._lib.fxc_plan_create(ctypes.byref(self._h), int(device), int(n_ant), int(nchan), int(ntaps),
                                       int(num_samp), window.ctypes.data, stream_ptr, PATHS[path])
        _lib.check(rc, None, self._lib)
        info = _lib.FxcInfo()
        self._check(self._lib.fxc_plan_get_info(self._h, ctypes.byref(info)))
        self.n_ant, self.n_baselines, self.nchan, self.ntaps = info.n_ant, info.n_baselines, info.nchan, info.ntaps
        self.num_samp, self.n_pts = info.num_samp, info.n_pts
        self.device = info.device
        self.path = PATH_NAMES[info.path]

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        _lib.check(rc, self._h, self._lib)

    def close(self):
        if getattr(self, "_abandoned", False):      # a pipe of this plan was abandoned with a thread still writing into its
            self._h = ctypes.c_void_p()             # slots (FxPipeline.abandon): the plan stays, for the process's life
            return
        if getattr(self, "_h", None) is not None and self._h:
            for ref in list(getattr(self, "_pipes", ())):     # pipes hold a pointer to the plan: they go first
                pipe = ref()
                if pipe is not None:
                    pipe.close()
            self._pipes = []
            # FXC_ERR_STATE: a pipe made behind this object's back still uses the plan -- keep the handle (the plan and
            # its device buffers stay valid for that pipe) and say so instead of leaking silently
            self._check(self._lib.fxc_plan_destroy(self._h))
            self._h = ctypes.c_void_p()

    def _sync_stream(self):
        """Plans created with stream=None issue every call on torch's current stream of their device."""
        if not self._follow:
            return
        cur = _current_torch_stream(self.device)
        if cur != self._stream:
            self._check(self._lib.fxc_set_stream(self._h, ctypes.c_void_p(cur) if cur else None))
            self._stream = cur

    def set_stream(self, stream):
        """Issue the plan's work on ``stream`` (a raw hipStream_t / ``torch.cuda.Stream.cuda_stream``) from now on and
        stop following torch's current stream."""
        self._follow = False
        stream = int(stream)
        self._check(self._lib.fxc_set_stream(self._h, ctypes.c_void_p(stream) if stream else None))
        self._stream = stream

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def info(self):
        info = _lib.FxcInfo()
        self._check(self._lib.fxc_plan_get_info(self._h, ctypes.byref(info)))
        return {name: getattr(info, name) for name, _ in _lib.FxcInfo._fields_}

    def _in(self, x, shape_tail, c128=False):
        """-> (pointer, mem_kind, leading count, keepalive).  ``c128``: complex128 input is handed over as it is
        (``self._fmt`` = FXC_IQ_C128) instead of being narrowed on the host."""
        self._sync_stream()
        self._fmt = _lib.FXC_IQ_C64
        if _is_torch(x):
            import torch
            if c128 and x.dtype == torch.complex128:
                self._fmt = _lib.FXC_IQ_C128
            elif x.dtype != torch.complex64:
                raise ValueError("device input must be a contiguous complex64 CUDA tensor")
            if not x.is_cuda or not x.is_contiguous():
                raise ValueError("device input must be a contiguous complex64 CUDA tensor")
            if x.device.index != self.device:
                raise ValueError("tensor is on device {} but the plan is on {}".format(x.device.index, self.device))
            shape = tuple(x.shape)
            ptr, kind, keep = x.data_ptr(), _lib.FXC_MEM_DEVICE, x
        else:
            if c128 and getattr(x, "dtype", None) == np.complex128:
                self._fmt = _lib.FXC_IQ_C128
                keep = np.ascontiguousarray(x)
            else:
                keep = np.ascontiguousarray(x, dtype=np.complex64)
            shape = keep.shape
            ptr, kind = keep.ctypes.data, _lib.FXC_MEM_HOST
        if len(shape) == len(shape_tail):
            shape = (1,) + shape
        if len(shape) != len(shape_tail) + 1 or tuple(shape[1:]) != tuple(shape_tail):
            raise ValueError("expected shape [n, {}], got {}".format(", ".join(map(str, shape_tail)), shape))
        return ptr, kind, shape[0], keep

    def _out(self, like, shape, dtype, out=None):
        self._to_pinned = False
        if out is not None:
            if _is_torch(like) and not _is_torch(out):
                # device-resident samples, rows into pinned host memory (``pinned_empty``): the finishing kernel writes them
                # there across PCIe, asynchronously -- the caller orders its reads with ``sync()`` or an event on the stream
                if out.dtype != dtype or out.shape != tuple(shape) or not out.flags.c_contiguous or not out.flags.writeable:
                    raise ValueError("out must be a writable C-contiguous {} array of shape {}".format(np.dtype(dtype).name, tuple(shape)))
                self._to_pinned = True
                return out, out.ctypes.data
            if _is_torch(like) != _is_torch(out):
                raise ValueError("out must be of the same kind (host array / CUDA tensor) as the input")
            if _is_torch(out):
                import torch
                tdt = torch.complex64 if dtype == np.complex64 else torch.complex128
                if out.dtype != tdt or tuple(out.shape) != tuple(shape) or not out.is_contiguous() or not out.is_cuda:
                    raise ValueError("out must be a contiguous CUDA tensor of shape {}".format(tuple(shape)))
                return out, out.data_ptr()
            if out.dtype != dtype or out.shape != tuple(shape) or not out.flags.c_contiguous or not out.flags.writeable:
                raise ValueError("out must be a writable C-contiguous {} array of shape {}".format(np.dtype(dtype).name, tuple(shape)))
            return out, out.ctypes.data
        if _is_torch(like):
            import torch
            tdt = torch.complex64 if dtype == np.complex64 else torch.complex128
            out = torch.empty(shape, dtype=tdt, device=like.device)
            return out, out.data_ptr()
        out = np.empty(shape, dtype=dtype)
        return out, out.ctypes.data

    # -- configuration ----------------------------------------------------------------------
    def set_rot(self, rot):
        rot = np.ascontiguousarray(rot, dtype=np.complex128)
        if rot.shape != (self.nchan,):
            raise ValueError("rot must have shape ({},)".format(self.nchan))
        self._check(self._lib.fxc_set_rot(self._h, rot.ctypes.data))

    def set_delay(self, bandwidth, frequency, calibrated_delay):
        self.set_rot(rot_table(self.nchan, bandwidth, frequency, calibrated_delay))

    # -- F stage ----------------------------------------------------------------------------
    def channelize(self, x):
        """x: [n_streams, num_samp] (or [num_samp]) complex64 -> [n_streams, n_pts, nchan] complex64."""
        ptr, kind, n, keep = self._in(x, (self.num_samp,))
        out, optr = self._out(x, (n, self.n_pts, self.nchan), np.complex64)
        self._check(self._lib.fxc_channelize(self._h, ptr, optr, n, kind))
        return out

    # -- F + X ------------------------------------------------------------------------------
    def fx_accumulate(self, x, remove_dc=False, c128=False):
        """x: [n_chunks, n_ant, num_samp] complex64; adds into the plan's accumulator (async).  ``remove_dc`` / ``c128``
        as for ``fx_rows``."""
        ptr, kind, n, keep = self._in(x, (self.n_ant, self.num_samp), c128)
        if remove_dc or self._fmt != _lib.FXC_IQ_C64:
            self._check(self._lib.fxc_fx_accumulate_iq(self._h, ptr, n, kind, self._fmt, int(bool(remove_dc))))
        else:
            self._check(self._lib.fxc_fx_accumulate(self._h, ptr, n, kind))
        return n

    def fx_rows(self, x, mode="SPECTRUM", bandwidth=1.0, remove_dc=False, out=None, c128=False):
        """One visibility row per chunk (the reference's ``_run_task`` result, effex.py:490-527).

        SPECTRUM -> [n_chunks, n_baselines, nchan] complex64; CONTINUUM/TEST -> [n_chunks, n_baselines]
        complex128.  ``remove_dc``: the per-chunk, per-antenna mean is removed on the device first (effex.py:394-395;
        ``fxc_fx_rows_iq``).  ``c128``: complex128 input crosses to the device as it is and is narrowed there (after the
        DC removal) instead of on the host.  ``out``: an array / tensor of the result's shape to receive the rows -- a
        ``pinned_empty`` array is written by the device itself.
        """
        m = MODES[mode.upper()]
        ptr, kind, n, keep = self._in(x, (self.n_ant, self.num_samp), c128)
        if m == _lib.FXC_MODE_SPECTRUM:
            out, optr = self._out(x, (n, self.n_baselines, self.nchan), np.complex64, out)
        else:
            out, optr = self._out(x, (n, self.n_baselines), np.complex128, out)
        if self._to_pinned:
            kind = _lib.FXC_MEM_DEVICE_TO_PINNED
        if remove_dc or self._fmt != _lib.FXC_IQ_C64:
            self._check(self._lib.fxc_fx_rows_iq(self._h, ptr, optr, n, kind, m, float(bandwidth), self._fmt,
                                                 int(bool(remove_dc))))
        else:
            self._check(self._lib.fxc_fx_rows(self._h, ptr, optr, n, kind, m, float(bandwidth)))
        return out

    def acc_reset(self):
        self._sync_stream()
        self._check(self._lib.fxc_acc_reset(self._h))

    def acc_export(self, sums):
        """sums: CUDA complex128 tensor [n_baselines*nchan + 1] (raw sums + {spectra count})."""
        import torch
        if sums.dtype != torch.complex128 or sums.numel() != self.n_baselines * self.nchan + 1 \
                or not sums.is_contiguous() or not sums.is_cuda:
            raise ValueError("sums must be a contiguous CUDA complex128 tensor of n_baselines*nchan + 1 elements")
        self._sync_stream()
        self._check(self._lib.fxc_acc_export(self._h, sums.data_ptr()))
        return sums

    def new_sums(self):
        import torch
        return torch.empty(self.n_baselines * self.nchan + 1, dtype=torch.complex128,
                           device=torch.device("cuda", self.device))

    def finalize_sums(self, sums=None, mode="SPECTRUM", bandwidth=1.0):
        """Visibilities from exported (and reduced) sums; ``sums=None``: the plan's own copy, as ``reduce`` leaves it."""
        m = MODES[mode.upper()]
        shape = (self.n_baselines, self.nchan) if m == _lib.FXC_MODE_SPECTRUM else (self.n_baselines,)
        out = np.empty(shape, dtype=np.complex128)
        self._sync_stream()
        ptr = sums.data_ptr() if sums is not None else None
        self._check(self._lib.fxc_finalize_sums(self._h, ptr, out.ctypes.data, m, float(bandwidth)))
        return out

    def reduce(self, comm=None, root=0):
        """``fxc_reduce``: export the accumulator into the plan and sum it over the ranks of ``comm`` (an ``RcclComm``;
        None = single rank) on the plan's stream — ncclReduce to ``root``, ncclAllReduce if ``root`` is None."""
        self._sync_stream()
        handle = comm.handle if comm is not None else None
        self._check(self._lib.fxc_reduce(self._h, handle, -1 if root is None else int(root)))

    def finalize(self, mode="SPECTRUM", bandwidth=1.0, reset=True):
        """Mean over everything accumulated, times conj(rot), fft-shifted -> numpy complex128."""
        self._sync_stream()
        m = MODES[mode.upper()]
        shape = (self.n_baselines, self.nchan) if m == _lib.FXC_MODE_SPECTRUM else (self.n_baselines,)
        out = np.empty(shape, dtype=np.complex128)
        self._check(self._lib.fxc_finalize(self._h, out.ctypes.data, m, float(bandwidth), int(bool(reset))))
        return out

    def _result(self, mode_code):
        shape = (self.n_baselines, self.nchan) if mode_code == _lib.FXC_MODE_SPECTRUM else (self.n_baselines,)
        return np.empty(shape, dtype=np.complex128)

    def finalize_async(self, mode="SPECTRUM", bandwidth=1.0, reset=True, out=None):
        """Queue ``finalize`` on the plan's stream and return at once (``fxc_finalize_async``): on the 2-antenna fast
        paths the fold of the last ``fx_accumulate``'s partial sums, the finalize, the reset and the write into pinned
        host memory are one kernel.  Collect with ``finalize_wait()``; up to two results may be outstanding, so the
        next integration can be queued before the host waits for this one.  ``out``: a complex128 array of the result's
        shape (best: ``pinned_empty``) the device delivers the result into (``fxc_finalize_async_to``);
        ``finalize_wait()`` then returns that array without copying anything."""
        self._sync_stream()
        m = MODES[mode.upper()]
        if out is not None:
            want = self._result(m).shape
            if out.dtype != np.complex128 or out.shape != want or not out.flags.c_contiguous or not out.flags.writeable:
                raise ValueError("out must be a writable C-contiguous complex128 array of shape {}".format(want))
            self._check(self._lib.fxc_finalize_async_to(self._h, out.ctypes.data, m, float(bandwidth), int(bool(reset))))
            self._queued.append((m, out))
            return
        self._check(self._lib.fxc_finalize_async(self._h, m, float(bandwidth), int(bool(reset))))
        self._queued.append(m)

    def finalize_sums_async(self, sums=None, mode="SPECTRUM", bandwidth=1.0):
        """``finalize_sums`` without the wait (``fxc_finalize_sums_async``); collect with ``finalize_wait()``."""
        self._sync_stream()
        m = MODES[mode.upper()]
        ptr = sums.data_ptr() if sums is not None else None
        self._check(self._lib.fxc_finalize_sums_async(self._h, ptr, m, float(bandwidth)))
        self._queued.append(m)

    def finalize_wait(self):
        """The oldest queued finalize result as numpy complex128 (blocks on that result's event only)."""
        if not self._queued:
            raise _lib.FxcError(_lib.FXC_ERR_STATE, "no finalize result outstanding")
        head = self._queued[0]
        out = head[1] if isinstance(head, tuple) else self._result(head)
        self._check(self._lib.fxc_finalize_wait(self._h, out.ctypes.data))
        self._queued.pop(0)
        return out

    def warm_bytes(self):
        """Have the byte-ingest build of the kernel for this channel count ready before the first byte call: a plan builds it
        lazily, inside that call -- up to seconds of hiprtc in the middle of a live stream when the shape is neither pre-built nor
        cached.  ``fxc_spec_probe`` runs the same search now and leaves the code object in the run-time cache (a file read later).
        No effect on shapes without such a kernel; returns True if one is there."""
        if not (self.info["specialised"] & 1):
            return False
        return self._lib.fxc_spec_probe(int(self.nchan), int(self.ntaps), 1, None, None, 0) == 0

    @property
    def finalize_pending(self):
        return int(self._lib.fxc_finalize_pending(self._h))

    def sync(self):
        self._check(self._lib.fxc_sync(self._h))

    # -- input conditioning (device resident) -----------------------------------------------
    def remove_dc(self, x, out=None):
        """Per-stream DC removal (effex.py:394-395) of a CUDA complex64 tensor [..., num_samp]; in place
        unless ``out`` is given."""
        import torch
        if x.dtype != torch.complex64 or not x.is_cuda or not x.is_contiguous() or x.shape[-1] != self.num_samp:
            raise ValueError("x must be a contiguous CUDA complex64 tensor [..., num_samp]")
        out = x if out is None else out
        self._sync_stream()
        self._check(self._lib.fxc_remove_dc(self._h, x.data_ptr(), out.data_ptr(), x.numel() // self.num_samp))
        return out

    def convert_u8(self, iq_u8, remove_dc=True):
        """RTL-SDR interleaved uint8 I,Q [..., num_samp, 2] (CUDA) -> complex64 [..., num_samp]."""
        import torch
        if iq_u8.dtype != torch.uint8 or not iq_u8.is_cuda or not iq_u8.is_contiguous() \
                or tuple(iq_u8.shape[-2:]) != (self.num_samp, 2):
            raise ValueError("iq_u8 must be a contiguous CUDA uint8 tensor [..., num_samp, 2]")
        self._sync_stream()
        out = torch.empty(iq_u8.shape[:-1], dtype=torch.complex64, device=iq_u8.device)
        self._check(self._lib.fxc_convert_u8(self._h, iq_u8.data_ptr(), out.data_ptr(), out.numel() // self.num_samp,
                                             int(bool(remove_dc))))
        return out

    def _in_u8(self, iq_u8):
        """uint8 I,Q [n_chunks, n_ant, num_samp, 2] (CUDA tensor or host array) -> (pointer, mem_kind, n, keepalive)."""
        tail = (self.n_ant, self.num_samp, 2)
        self._sync_stream()
        if _is_torch(iq_u8):
            import torch
            if iq_u8.dtype != torch.uint8 or not iq_u8.is_cuda or not iq_u8.is_contiguous():
                raise ValueError("device input must be a contiguous uint8 CUDA tensor")
            shape, ptr, kind, keep = tuple(iq_u8.shape), iq_u8.data_ptr(), _lib.FXC_MEM_DEVICE, iq_u8
        else:
            keep = np.ascontiguousarray(iq_u8, dtype=np.uint8)
            shape, ptr, kind = keep.shape, keep.ctypes.data, _lib.FXC_MEM_HOST
        if len(shape) == 3:
            shape = (1,) + shape
        if len(shape) != 4 or tuple(shape[1:]) != tail:
            raise ValueError("expected shape [n, {}, {}, 2], got {}".format(self.n_ant, self.num_samp, shape))
        return ptr, kind, shape[0], keep

    def fx_rows_u8(self, iq_u8, mode="SPECTRUM", bandwidth=1.0, remove_dc=True, out=None):
        """``fx_rows`` straight from RTL-SDR bytes: uint8 I,Q [n_chunks, n_ant, num_samp, 2], converted as pyrtlsdr does
        (effex.py:652) with the per-stream mean removed (effex.py:394-395) unless ``remove_dc`` is false.  On fused
        plans the F+X kernel reads the bytes itself."""
        m = MODES[mode.upper()]
        ptr, kind, n, keep = self._in_u8(iq_u8)
        if m == _lib.FXC_MODE_SPECTRUM:
            out, optr = self._out(iq_u8, (n, self.n_baselines, self.nchan), np.complex64, out)
        else:
            out, optr = self._out(iq_u8, (n, self.n_baselines), np.complex128, out)
        if self._to_pinned:
            kind = _lib.FXC_MEM_DEVICE_TO_PINNED
        self._check(self._lib.fxc_fx_rows_u8(self._h, ptr, optr, n, kind, m, float(bandwidth), int(bool(remove_dc))))
        return out

    def fx_accumulate_u8(self, iq_u8, remove_dc=True):
        """``fx_accumulate`` straight from RTL-SDR bytes (see ``fx_rows_u8``)."""
        ptr, kind, n, keep = self._in_u8(iq_u8)
        self._check(self._lib.fxc_fx_accumulate_u8(self._h, ptr, n, kind, int(bool(remove_dc))))
        return n

    # -- delay calibration ------------------------------------------------------------------
    def estimate_delay(self, iq_0, iq_1, rate):
        """Sub-sample delay between two equal-length streams in seconds (effex.py:583-627)."""
        if len(iq_0) != len(iq_1):
            raise AssertionError('Algorithm assumes input complex timeseries are of equal length.')
        n = int(len(iq_0))
        self._sync_stream()
        if _is_torch(iq_0) or _is_torch(iq_1):
            import torch
            for t in (iq_0, iq_1):
                if not _is_torch(t) or t.dtype != torch.complex64 or not t.is_cuda or t.dim() != 1 \
                        or t.device.index != self.device:
                    raise ValueError("device inputs must both be 1-D complex64 CUDA tensors on device {}".format(self.device))
            a, b, kind = iq_0.contiguous(), iq_1.contiguous(), _lib.FXC_MEM_DEVICE
            pa, pb = a.data_ptr(), b.data_ptr()
        else:
            a = np.ascontiguousarray(iq_0, dtype=np.complex64)
            b = np.ascontiguousarray(iq_1, dtype=np.complex64)
            kind, pa, pb = _lib.FXC_MEM_HOST, a.ctypes.data, b.ctypes.data
        out = ctypes.c_double()
        self._check(self._lib.fxc_estimate_delay(self._h, pa, pb, n, kind, float(rate), ctypes.byref(out)))
        return out.value

    # -- measurement ------------------------------------------------------------------------
    def timer_start(self):
        self._check(self._lib.fxc_timer_start(self._h))

    def timer_stop(self):
        ms = ctypes.c_double()
        self._check(self._lib.fxc_timer_stop(self._h, ctypes.byref(ms)))
        return ms.value

    def kernel_profiling(self, enable):
        self._check(self._lib.fxc_kernel_profiling(self._h, int(bool(enable))))

    def kernel_time(self, reset=True):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        self._check(self._lib.fxc_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(n), int(bool(reset))))
        return ms.value, n.value


class FxPipeline(object):
    """Host-fed, double-buffered front end on a plan (include/fxcorr.h ``fxc_pipe_*``): push batches of host
    chunks, pop their visibility rows; H2D, compute and D2H of successive batches overlap."""

    def __init__(self, plan, chunks_per_batch, depth=2, mode="SPECTRUM", bandwidth=1.0, u8=False, remove_dc=None, fmt=None):
        """``fmt``: sample format of the batches -- 'c64' (default), 'u8' (the receivers' bytes, uint8
        [chunks, n_ant, num_samp, 2]; ``u8=True`` is the same) or 'c128'.  ``remove_dc``: the per-chunk mean is removed
        on the device (effex.py:394-395); default: on for bytes, off otherwise (``fxc_pipe_create_iq``)."""
        self.plan = plan
        self.chunks = int(chunks_per_batch)
        self.mode = MODES[mode.upper()]
        fmt = fmt or ("u8" if u8 else "c64")
        if fmt not in IQ_FORMATS:
            raise ValueError("fmt must be one of {}".format(sorted(IQ_FORMATS)))
        self.fmt = fmt
        self.u8 = fmt == "u8"
        if remove_dc is None:
            remove_dc = self.u8
        self._shape = (self.chunks, plan.n_ant, plan.num_samp) + ((2,) if self.u8 else ())
        self._dtype = _IQ_DTYPES[fmt]
        self._h = ctypes.c_void_p()
        plan._check(plan._lib.fxc_pipe_create_iq(ctypes.byref(self._h), plan._h, self.chunks, int(depth), self.mode,
                                                 float(bandwidth), IQ_FORMATS[fmt], int(bool(remove_dc))))
        import weakref
        plan._pipes.append(weakref.ref(self))      # FxPlan.close() closes its pipes first

    def push(self, x):
        x = np.ascontiguousarray(x, dtype=self._dtype)
        if x.shape != self._shape:
            raise ValueError("expected a batch of shape {}".format(self._shape))
        self.plan._check(self.plan._lib.fxc_pipe_push(self._h, x.ctypes.data))

    def acquire(self):
        """Pinned input buffer of the next free slot as a numpy view (the batch shape) to fill in place."""
        ptr = ctypes.c_void_p()
        self.plan._check(self.plan._lib.fxc_pipe_acquire(self._h, ctypes.byref(ptr)))
        n_bytes = int(np.prod(self._shape)) * np.dtype(self._dtype).itemsize
        buf = (ctypes.c_uint8 * n_bytes).from_address(ptr.value)
        return np.frombuffer(buf, dtype=self._dtype).reshape(self._shape)

    def submit(self):
        self.plan._check(self.plan._lib.fxc_pipe_submit(self._h))

    def pop(self, out=None):
        """The oldest batch's rows.  ``out``: a C-contiguous array of the batch's shape and dtype to receive them in place --
        e.g. a window of a memory-mapped row file (``effex_amd.rowsink.BinSink.reserve``): the rows then go from the pinned
        result slot straight into the page cache."""
        if self.mode == _lib.FXC_MODE_SPECTRUM:
            shape, dtype = (self.chunks, self.plan.n_baselines, self.plan.nchan), np.complex64
        else:
            shape, dtype = (self.chunks, self.plan.n_baselines), np.complex128
        if out is None:
            out = np.empty(shape, dtype=dtype)
        elif out.dtype != dtype or out.size != int(np.prod(shape)) or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("out must be a writable C-contiguous {} array of {} elements".format(np.dtype(dtype).name, int(np.prod(shape))))
        self.plan._check(self.plan._lib.fxc_pipe_pop(self._h, out.ctypes.data))
        return out

    @property
    def in_flight(self):
        return self.plan._lib.fxc_pipe_in_flight(self._h)

    def close(self):
        if self._h:
            self.plan._lib.fxc_pipe_destroy(self._h)
            self._h = ctypes.c_void_p()

    def abandon(self):
        """Give the pipe up WITHOUT freeing it: for the error path of a caller whose producer thread may still be writing
        into a pinned slot (a read that did not notice its source being closed).  The slots, and the plan they belong to,
        stay allocated for the life of the process -- a leak instead of a write into freed memory."""
        self._h = ctypes.c_void_p()
        self.plan._abandoned = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class RcclComm(object):
    """One rank of an RCCL communicator made by libfxcorr (``fxc_comm_*``) for ``FxPlan.reduce``.

    ``unique_id()`` on rank 0 gives the 128 bytes every rank needs; how they travel is the caller's business
    (``effex_amd.sharding.make_comm`` broadcasts them through ``torch.distributed``).  Creation is collective."""

    def __init__(self, device, rank, world_size, unique_id):
        self._lib = _lib.load()
        self.handle = ctypes.c_void_p()
        if len(unique_id) != _lib.FXC_COMM_ID_BYTES:
            raise ValueError("unique_id must be {} bytes".format(_lib.FXC_COMM_ID_BYTES))
        buf = (ctypes.c_char * _lib.FXC_COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _lib.check(self._lib.fxc_comm_create(ctypes.byref(self.handle), int(device), int(rank), int(world_size),
                                             ctypes.cast(buf, ctypes.c_void_p)), None)
        self.rank, self.world_size, self.device = int(rank), int(world_size), int(device)

    @staticmethod
    def unique_id():
        lib = _lib.load()
        buf = (ctypes.c_char * _lib.FXC_COMM_ID_BYTES)()
        _lib.check(lib.fxc_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), None)
        return bytes(buf.raw)

    def info(self):
        """What the live communicator says about itself (``fxc_comm_info``): ``ranks_seen`` / ``rank_seen`` /
        ``device_seen`` are asked of the ncclComm_t (None where the bound RCCL lacks the query), ``*_given`` are what this
        object was made with; ``rccl_version``, ``async_error`` (0 = ncclSuccess), ``reduces`` queued through it."""
        d = _lib.FxcCommDesc()
        _lib.check(self._lib.fxc_comm_info(self.handle, ctypes.byref(d)), None)
        out = {name: int(getattr(d, name)) for name, _ in d._fields_}
        for key in ("ranks_seen", "rank_seen", "device_seen", "async_error"):
            if out[key] < 0:
                out[key] = None
        return out

    def probe(self):
        """Collective, blocking: every rank contributes 1.0 to one ncclAllReduce; returns the count RCCL added up."""
        n = ctypes.c_int64(0)
        _lib.check(self._lib.fxc_comm_probe(self.handle, ctypes.byref(n)), None)
        return int(n.value)

    @staticmethod
    def library():
        """(version, path) of the librccl bound at run time; raises FxcError(FXC_ERR_COMM) if none could be."""
        lib = _lib.load()
        v = ctypes.c_int(0)
        buf = ctypes.create_string_buffer(1024)
        _lib.check(lib.fxc_rccl_version(ctypes.byref(v), buf, len(buf)), None)
        return int(v.value), buf.value.decode("utf-8", "replace")

    def close(self):
        if getattr(self, "handle", None):
            self._lib.fxc_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def synth_fill(x, seed, first_chunk=0, delays=None, stream=None):
    """Fill a CUDA complex64 tensor [n_chunks, n_ant, num_samp] with the synthetic stream of
    ``effex_amd.synth`` (bit-identical), generated on the device."""
    import torch
    from . import synth
    lib = _lib.load()
    if x.dtype != torch.complex64 or not x.is_cuda or not x.is_contiguous() or x.dim() != 3:
        raise ValueError("x must be a contiguous CUDA complex64 tensor [n_chunks, n_ant, num_samp]")
    n_chunks, n_ant, num_samp = x.shape
    if delays is None:
        delays = synth.DEFAULT_DELAYS
    d = np.ascontiguousarray(delays[:n_ant], dtype=np.int32)
    if len(d) < n_ant:
        raise ValueError("not enough delays for n_ant")
    tone = synth.tone_table()
    if stream is None:
        stream = torch.cuda.current_stream(x.device).cuda_stream
    rc = lib.fxc_synth_fill(x.device.index, ctypes.c_void_p(int(stream)), x.data_ptr(), int(seed), int(first_chunk),
                            n_chunks, n_ant, num_samp, d.ctypes.data, tone.ctypes.data, len(tone))
    _lib.check(rc, None)
    return x
