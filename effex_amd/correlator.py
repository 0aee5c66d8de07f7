"""Drop-in ``Correlator`` for effex's F/X hot path on MI355X.

Keeps the call surface of ``/root/reference/effex/effex.py`` for the path (constructor keywords and
defaults ``:45-53``, validated properties ``:199-320``, state names and legal transitions ``:210-224``,
``_spectrometer_poly`` ``:530-555``, ``_pfb_xcorr`` / ``_run_task`` ``:490-527``, the ``.csv`` layout
``:667-696``) and routes the arithmetic through libfxcorr (HIP).  What it does *not* reproduce is the
reference's acquisition / threading / keyboard layer (SURVEY.md §8 scope): samples come from an
``IQSource`` object instead of two RTL-SDR dongles (``:81-82``), and rows are written by the caller's
thread instead of a writer thread fed through a ``multiprocessing.Queue`` (``:134,687-696``).

There is no CPU arithmetic path here: every F/X result comes from the HIP library.
"""
import logging
import time

import numpy as np

from . import rowsink, synth
from . import _lib
from .plan import FxPlan, pinned_empty, rot_table
from .window import design_window


class IQSource(object):
    """What the correlator needs from a receiver pair: tuning attributes and chunk pairs."""
    rs = None       # sample rate (Hz)      — RtlSdr.rs, effex.py:256-257
    fc = None       # centre frequency (Hz) — RtlSdr.fc, effex.py:268-269
    gain = None     # tuner gain            — RtlSdr.gain, effex.py:305-306

    def read(self, num_samp):
        """Return (iq_0, iq_1), each ``num_samp`` complex samples, or None when the stream ends.

        A source may instead hand over the receivers' raw bytes — two uint8 arrays [num_samp, 2] of interleaved
        I,Q (pyrtlsdr's ``format='bytes'``): conversion, DC removal and F+X then happen in one device call."""
        raise NotImplementedError

    def read_into(self, num_samp, out_0, out_1):
        """Fill ``out_0`` / ``out_1`` (uint8 [num_samp, 2] for a byte source, complex64 [num_samp] otherwise) with the next
        chunk pair; False when the stream ends.  The batched run (``Correlator(batch=...)``) hands over windows of a pinned
        staging slot, so a source that overrides this writes its data where the device copies it from."""
        pair = self.read(num_samp)
        if pair is None:
            return False
        out_0[...] = np.asarray(pair[0]).reshape(out_0.shape)
        out_1[...] = np.asarray(pair[1]).reshape(out_1.shape)
        return True

    def read_many_into(self, num_samp, out):
        """Fill ``out[k, a]`` (a batch window [K, 2, num_samp(, 2)]) with up to K consecutive chunk pairs; returns how many
        were filled (fewer than K: the stream has ended)."""
        k = 0
        while k < len(out) and self.read_into(num_samp, out[k, 0], out[k, 1]):
            k += 1
        return k

    def close(self):
        pass


class SyntheticSource(IQSource):
    """Counter-based synthetic sky (``effex_amd.synth``): a fixed number of chunk pairs."""

    def __init__(self, seed=1234, n_chunks=4, delays=synth.DEFAULT_DELAYS):
        self.seed, self.n_chunks, self.delays = seed, n_chunks, delays
        self._next = 0
        self.closed = False

    def read(self, num_samp):
        if self._next >= self.n_chunks:
            return None
        iq = synth.synth_iq(self.seed, 1, 2, int(num_samp), first_chunk=self._next, delays=self.delays)[0]
        self._next += 1
        return iq[0], iq[1]

    def close(self):
        self.closed = True


class ArraySource(IQSource):
    """Chunk pairs from an array [n_chunks, 2, num_samp]."""

    def __init__(self, chunks):
        self.chunks = np.asarray(chunks)
        self._next = 0
        self.closed = False

    def read(self, num_samp):
        if self._next >= len(self.chunks):
            return None
        pair = self.chunks[self._next]
        self._next += 1
        return pair[0][:num_samp], pair[1][:num_samp]

    def close(self):
        self.closed = True


class FileSource(IQSource):
    """Chunk pairs from two recordings, one per receiver (the reference only reads live RTL-SDR dongles,
    effex.py:81-82, 630-664; recorded streams are how its path is fed without them).

    ``fmt='u8'``: raw interleaved unsigned 8-bit I,Q as ``rtl_sdr`` writes them — handed over as bytes, so conversion,
    DC removal and F+X happen in one device call (``fxc_fx_rows_u8``).  ``fmt='c64'``: raw complex64 samples.  The files
    are memory-mapped and read chunk by chunk; a trailing partial chunk is dropped, as a short read ends the reference's
    run."""

    _READERS = 4        # read_many_into: threads per receiver

    def __init__(self, path_0, path_1, fmt='u8', rs=None, fc=None, gain=None):
        if fmt not in ('u8', 'c64'):
            raise ValueError("fmt must be 'u8' or 'c64'")
        self.fmt = fmt
        self.rs, self.fc, self.gain = rs, fc, gain
        dtype = np.uint8 if fmt == 'u8' else np.complex64
        self._paths = (path_0, path_1)
        self._fds = None            # file descriptors of read_many_into, opened on first use
        self._maps = [np.memmap(p, dtype=dtype, mode='r') for p in (path_0, path_1)]
        per = 2 if fmt == 'u8' else 1
        self.n_samples = min(len(m) for m in self._maps) // per
        self._next = 0
        self._pool = None           # copy threads of read_many_into, made on first use
        self.closed = False

    def read(self, num_samp):
        num_samp = int(num_samp)
        lo, hi = self._next, self._next + num_samp
        if hi > self.n_samples:
            return None
        self._next = hi
        if self.fmt == 'u8':
            return tuple(np.asarray(m[2 * lo:2 * hi]).reshape(num_samp, 2) for m in self._maps)
        return tuple(np.asarray(m[lo:hi]) for m in self._maps)

    def read_into(self, num_samp, out_0, out_1):
        num_samp = int(num_samp)
        lo, hi = self._next, self._next + num_samp
        if hi > self.n_samples:
            return False
        self._next = hi
        per = 2 if self.fmt == 'u8' else 1
        for m, out in zip(self._maps, (out_0, out_1)):       # page cache -> staging slot, one copy
            out.reshape(-1)[...] = m[per * lo:per * hi]
        return True

    def read_many_into(self, num_samp, out):
        """Scatter reads (``os.preadv``), ``_READERS`` per receiver in parallel: the kernel copies from the page cache
        straight into the batch window (through the memory map, with its page faults, eight threads moved 13 GB/s)."""
        import os
        num_samp = int(num_samp)
        k = min(len(out), (self.n_samples - self._next) // num_samp)
        if k <= 0:
            return 0
        bytes_per = 2 if self.fmt == 'u8' else 8
        lo = self._next
        self._next += k * num_samp
        if self._fds is None:
            self._fds = [os.open(path, os.O_RDONLY) for path in self._paths]

        def fill(job):
            a, c_lo, c_hi = job
            bufs = [out[c, a].reshape(-1).view(np.uint8) for c in range(c_lo, c_hi)]
            per_chunk = num_samp * bytes_per
            offset, want = (lo + c_lo * num_samp) * bytes_per, (c_hi - c_lo) * per_chunk
            # preadv takes at most IOV_MAX buffers and may return short: continue from where it stopped
            done = 0
            while done < want:
                c0, skip = divmod(done, per_chunk)
                iov = [bufs[c0][skip:]] + bufs[c0 + 1:c0 + 512]
                got = os.preadv(self._fds[a], iov, offset + done)
                if got <= 0:
                    raise EOFError("recording {} shrank while it was read".format(self._paths[a]))
                done += got

        if self._pool is None:
            import concurrent.futures
            self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=2 * self._READERS)
        parts = min(self._READERS, k)
        jobs = [(a, k * i // parts, k * (i + 1) // parts) for a in range(2) for i in range(parts)]
        list(self._pool.map(fill, jobs))
        return k

    def close(self):
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        if self._fds is not None:
            import os
            for fd in self._fds:
                os.close(fd)
            self._fds = None
        self._maps = []
        self.closed = True


class SocketSource(IQSource):
    """Chunk pairs from two byte streams over TCP, one per receiver (the reference's producers are live
    streams, ``sdr.stream()`` in two processes, effex.py:630-664; a network stream is how a remote or shared receiver --
    ``rtl_tcp``, a recorder replaying a capture -- is fed to the path).

    ``endpoints``: two ``(host, port)`` pairs to connect to, or two already connected sockets.  ``fmt='u8'``: interleaved
    unsigned 8-bit I,Q, ``rtl_tcp``'s sample format, handed over as bytes (conversion, DC removal and F+X in one device
    call); ``fmt='c64'``: raw little-endian complex64.  ``skip`` bytes are dropped from each stream first (``rtl_tcp``
    greets with a 12-byte header).  A stream that ends inside a chunk ends the run, as a short read ends the reference's
    (``read`` returns None).  Reads block; ``timeout`` seconds without data raise ``socket.timeout``."""

    def __init__(self, endpoints, fmt='u8', rs=None, fc=None, gain=None, skip=0, timeout=None):
        import socket
        if fmt not in ('u8', 'c64'):
            raise ValueError("fmt must be 'u8' or 'c64'")
        if len(endpoints) != 2:
            raise ValueError("two endpoints, one per receiver")
        self.fmt = fmt
        self.rs, self.fc, self.gain = rs, fc, gain
        self._socks = []
        for ep in endpoints:
            sock = ep if hasattr(ep, "recv_into") else socket.create_connection(tuple(ep), timeout=timeout)
            sock.settimeout(timeout)
            self._socks.append(sock)
        self.closed = False
        for sock in self._socks:
            if skip and self._recv_exact(sock, bytearray(int(skip))) is None:
                raise EOFError("stream ended inside its {}-byte greeting".format(skip))

    @staticmethod
    def _recv_exact(sock, buf):
        view, got = memoryview(buf), 0
        while got < len(buf):
            n = sock.recv_into(view[got:])
            if n == 0:
                return None
            got += n
        return buf

    def read(self, num_samp):
        num_samp = int(num_samp)
        per = 2 if self.fmt == 'u8' else 8
        out = []
        for sock in self._socks:          # the two receivers' chunks, one after the other (kernel socket buffers decouple them)
            buf = self._recv_exact(sock, bytearray(num_samp * per))
            if buf is None:
                return None
            out.append(np.frombuffer(buf, dtype=np.uint8).reshape(num_samp, 2) if self.fmt == 'u8'
                       else np.frombuffer(buf, dtype='<c8'))
        return tuple(out)

    def read_into(self, num_samp, out_0, out_1):
        for sock, out in zip(self._socks, (out_0, out_1)):   # socket -> staging slot, no copy in between
            if self._recv_exact(sock, out.reshape(-1).view(np.uint8)) is None:
                return False
        return True

    def close(self):
        for sock in self._socks:
            try:
                sock.close()
            except OSError:
                pass
        self._socks = []
        self.closed = True


def _without_mean(x):
    """x minus its complex mean (= the mean of the real parts and of the imaginary parts), in complex128 -- effex.py:394-395
    on the host.  Only the CALIBRATE state's one chunk pair per run still takes this route (the delay estimate reads the
    host arrays); RUN-state chunks are de-meaned on the device."""
    z = np.asarray(x).astype(np.complex128)
    return z - complex(z.real.mean(), z.imag.mean())


def _staging_empty(shape, dtype):
    """Pinned memory when a device is there to pin it for (``fxc_host_alloc``); on a box without one -- where no F/X call
    can succeed anyway -- an ordinary array, so that the class can still be constructed and configured."""
    try:
        return pinned_empty(shape, dtype), True
    except _lib.FxcError as exc:
        if exc.status != _lib.FXC_ERR_NODEVICE:
            raise
        return np.empty(shape, dtype=dtype), False


class Correlator(object):
    # class constants — effex.py:34-35
    _states = ('OFF', 'STARTUP', 'RUN', 'CALIBRATE', 'SHUTDOWN')
    _modes = ('SPECTRUM', 'CONTINUUM', 'TEST')
    # legal successors of each state — effex.py:210-224
    _transitions = {'OFF': ('STARTUP',),
                    'STARTUP': ('CALIBRATE', 'RUN', 'SHUTDOWN'),
                    'RUN': ('CALIBRATE', 'SHUTDOWN'),
                    'CALIBRATE': ('RUN', 'SHUTDOWN'),
                    'SHUTDOWN': ('OFF',)}
    _MAX_NUM_SAMP = 2 ** 18     # effex.py:282-283; lifted by max_num_samp= for the 2^20 continuum config

    class StateTransitionError(Exception):
        """effex.py:186-193."""

        def __init__(self, prev, next):
            self.prev = prev
            self.next = next
            self.message = 'Transition from {} to {} is not permitted.'.format(prev, next)

        def __str__(self):
            return repr(self.message)

    def __init__(self, run_time=1, bandwidth=2.4e6, frequency=1.4204e9, num_samp=2 ** 18, nbins=2 ** 12,
                 gain=49.6, mode='SPECTRUM', loglevel='INFO',
                 source=None, device=0, max_num_samp=None, output_file=None, remove_dc=True, calibrate=True,
                 output_format='csv', batch=1):
        self.logger = logging.getLogger(__name__)
        self.logger.setLevel(getattr(logging, loglevel))
        self._max_num_samp = int(max_num_samp) if max_num_samp else Correlator._MAX_NUM_SAMP
        self.source = source if source is not None else SyntheticSource()
        self.device = device
        self.remove_dc = remove_dc
        self.calibrate = calibrate      # the reference always calibrates on the first chunk pair (effex.py:353,399-401)
        # batch > 1: the RUN state takes that many chunk pairs per device call through the double-buffered host-fed front
        # end (fxc_pipe_*) -- for replaying recordings and fast streams; 1 = the reference's one call per chunk pair
        if int(batch) < 1:
            raise ValueError("batch must be >= 1")
        self.batch = int(batch)
        self._fx_plan = None
        self._f_plans = {}
        self._rot_key = None
        self._filler_grace_s = 10.0     # _run_batched's error path: how long a closed source's read may take to return

        self.run_time = run_time
        self.bandwidth = bandwidth
        self.frequency = frequency
        self.num_samp = num_samp
        self.nbins = nbins
        self.gain = gain
        self._state = 'OFF'
        self.mode = mode
        self.start_time = -1

        # staging buffers the hot path reads — effex.py:109-110: mapped pinned memory there (cusignal.get_shared_mem) and
        # here (fxc_host_alloc; complex128 there, complex64 here: the HIP path computes in complex64).
        # gpu_iq_0 / gpu_iq_1 are the two halves of one pinned chunk pair: _pfb_xcorr hands that buffer to the library as
        # it is.  Rebinding them to other arrays (as effex.py:394-395 does) is allowed: those are copied in per call.
        self._alloc_staging(int(self.num_samp))

        self.ntaps = 4                                           # effex.py:115
        n_int = len(self._gpu_iq[0]) // self.ntaps // self.nbins   # effex.py:118-124
        assert (n_int >= 1), ('Assertion failed: there must be at least 1 window of length n_branches*ntaps '
                              'in each input timeseries.\ntimeseries len: {}\nn_branches: {}\nntaps: {}\n'
                              'n_branches*ntaps: {}').format(len(self.gpu_iq_0), self.nbins, self.ntaps,
                                                             self.nbins * self.ntaps)
        self.window = design_window(self.ntaps, self.nbins)      # effex.py:126-127

        self.calibrated_delay = 0                                # effex.py:132
        # 'csv': the reference's file (effex.py:136, 667-696).  'bin': the binary sidecar of effex_amd.rowsink (same header
        # line, rows as the device produced them; tools/rows_to_csv.py turns it into the reference's csv byte for byte)
        if output_format not in ('csv', 'bin'):
            raise ValueError("output_format must be 'csv' or 'bin'")
        self.output_format = output_format
        self.output_file = output_file or (time.strftime('visibilities_%Y%m%d-%H%M%S') +
                                           ('.csv' if output_format == 'csv' else '.fxb'))
        crit_delay = 1 / self.frequency                          # effex.py:151-155
        self.test_delay_sweep_step = crit_delay / 2
        self.test_delay_offset = self.test_delay_sweep_step * 1600


    # -- staging buffers (effex.py:109-110) -------------------------------------------------
    def _alloc_staging(self, n):
        self._pair_buf, self._pinned = _staging_empty((1, 2, n), np.complex64)
        self._pair_buf[...] = 0
        self._gpu_iq = [self._pair_buf[0, 0], self._pair_buf[0, 1]]
        self._row_bufs = {}
        self._dc_pending = False    # the pair _stage left in the pinned buffers still carries its mean (see _settle_dc)

    def _row_buf(self, shape, dtype):
        key = (shape, np.dtype(dtype).str)
        buf = self._row_bufs.get(key)
        if buf is None:
            buf = self._row_bufs[key] = _staging_empty(shape, dtype)[0]
        return buf

    def _settle_dc(self):
        """The RUN loop's _stage leaves the raw chunk pair in the pinned buffers and lets the device take the mean off on
        the way in (``_dc_pending``; effex.py:394-395 without a host pass).  The reference's ``gpu_iq_0/1`` hold de-meaned
        samples at that point, and its ``_pfb_xcorr`` never removes anything: so the moment anybody looks at the buffers
        through the public names -- to read them, to write into them in place (``cor.gpu_iq_0[:] = data``, effex.py:391) or
        to rebind them -- the mean comes off here, on the host, in place, and the device is no longer asked to.  What
        ``_run_task`` then computes is exactly what the buffers hold."""
        if self._dc_pending:
            self._dc_pending = False
            for a in range(2):
                view = self._pair_buf[0, a]
                view -= np.complex64(complex(view.real.mean(dtype=np.float64), view.imag.mean(dtype=np.float64)))

    @property
    def gpu_iq_0(self):
        self._settle_dc()
        return self._gpu_iq[0]

    @gpu_iq_0.setter
    def gpu_iq_0(self, value):
        self._settle_dc()
        self._gpu_iq[0] = value

    @property
    def gpu_iq_1(self):
        self._settle_dc()
        return self._gpu_iq[1]

    @gpu_iq_1.setter
    def gpu_iq_1(self, value):
        self._settle_dc()
        self._gpu_iq[1] = value

    def _staged_pair(self):
        """The pinned chunk pair [1, 2, n] complex64 holding gpu_iq_0 / gpu_iq_1: as it is when they still are its two
        halves, else filled from whatever they were rebound to (one narrowing pass per stream)."""
        n = len(self._gpu_iq[0])
        if self._pair_buf.shape[2] != n:
            held = list(self._gpu_iq)       # (never with _dc_pending: _stage sizes the pair itself)
            self._alloc_staging(n)
            self._gpu_iq = held
        for a in range(2):
            view = self._pair_buf[0, a]
            if self._gpu_iq[a] is not view:
                np.copyto(view, np.asarray(self._gpu_iq[a]).reshape(n), casting='same_kind')
        return self._pair_buf

    # -- lifecycle --------------------------------------------------------------------------
    def close(self):
        """effex.py:176-180 — release the receivers (and the device plans)."""
        self.source.close()
        if self._fx_plan is not None:
            self._fx_plan.close()
            self._fx_plan = None
        for p in self._f_plans.values():
            p.close()
        self._f_plans = {}

    # -- properties (effex.py:199-320) ------------------------------------------------------
    @property
    def state(self):
        return self._state

    @state.setter
    def state(self, input_state):
        if input_state not in self._states:
            self.close()
            raise ValueError('State {} is not in known states: {}'.format(input_state, self._states))
        if input_state not in self._transitions[self._state]:
            self.close()
            raise self.StateTransitionError(self._state, input_state)
        self._state = input_state

    @property
    def run_time(self):
        return self._run_time

    @run_time.setter
    def run_time(self, run_time):
        if run_time < 1:
            self.close()
            raise ValueError('run time {} is not allowed; run times must be >= 1 second.'.format(run_time))
        self._run_time = run_time

    @property
    def bandwidth(self):
        return self._bandwidth

    @bandwidth.setter
    def bandwidth(self, value):
        threshold = 2.8e6
        if value > threshold:
            self.logger.warning('Bandwidth value {} is greater than {}, and RtlSdrs may not be stable.'.format(
                value, threshold))
        self._bandwidth = value
        self.source.rs = value

    @property
    def frequency(self):
        return self._frequency

    @frequency.setter
    def frequency(self, value):
        self._frequency = value
        self.source.fc = value

    @property
    def num_samp(self):
        return self._num_samp

    @num_samp.setter
    def num_samp(self, value):
        int_val = int(round(value))
        if int_val < 2 ** 8:
            value = 2 ** 8
        elif int_val > self._max_num_samp:
            value = self._max_num_samp
        self._num_samp = value           # clamped but not rounded, like the reference (:278-284)

    @property
    def nbins(self):
        return self._nbins

    @nbins.setter
    def nbins(self, value):
        self._nbins = value

    @property
    def gain(self):
        return self._gain

    @gain.setter
    def gain(self, value):
        self._gain = value
        self.source.gain = value

    @property
    def mode(self):
        return self._mode

    @mode.setter
    def mode(self, input_mode):
        input_mode = input_mode.upper()
        if input_mode not in self._modes:
            raise ValueError('Mode input {} is not in known modes: {}'.format(input_mode, self._modes))
        self._mode = input_mode

    # -- hot path ---------------------------------------------------------------------------
    def _spectrometer_poly(self, x, ntaps, n_branches, window):
        """effex.py:530-555 — (len(x)//n_branches, n_branches) complex, natural bin order.

        ``x`` may be a host array (result: numpy complex128 view of the complex64 device result) or a
        CUDA complex64 tensor (result: tensor).  ``ntaps > 32`` raises NotImplementedError like cusignal.
        """
        n_branches = int(n_branches)
        window = np.asarray(window, dtype=np.float64)
        ntaps_w = int(len(window) / n_branches)      # what channelize_poly actually uses
        is_tensor = type(x).__module__.startswith("torch")
        length = int(x.shape[0])
        key = (length, n_branches, ntaps_w, hash(window.tobytes()))
        plan = self._f_plans.get(key)
        if plan is None:
            plan = FxPlan(1, n_branches, ntaps_w, length, window=window[:ntaps_w * n_branches], device=self.device)
            self._f_plans[key] = plan
        if is_tensor:
            return plan.channelize(x)[0]
        out = plan.channelize(np.asarray(x).astype(np.complex64))[0]
        return out.astype(np.complex128)

    def _plan(self):
        n = int(self.num_samp)
        nb = int(self.nbins)
        if self._fx_plan is None or (self._fx_plan.num_samp, self._fx_plan.nchan) != (n, nb):
            if self._fx_plan is not None:
                self._fx_plan.close()
            # The window is designed once, for the constructor's nbins (effex.py:126-127), and the nbins setter only
            # stores the new value (effex.py:287-294): after ``cor.nbins = ...`` the reference channelises with the stale
            # window, and channelize_poly derives its tap count from it (len(h) / n_chans, first ntaps * n_chans taps).
            ntaps_w = int(len(self.window) / nb)
            if ntaps_w < 1:
                raise ValueError("nbins = {} exceeds the {} taps of the window designed at construction "
                                 "(effex.py:126-127)".format(nb, len(self.window)))
            self._fx_plan = FxPlan(2, nb, ntaps_w, n, window=np.asarray(self.window)[:ntaps_w * nb], device=self.device)
            if getattr(self.source, "fmt", None) == "u8":      # a byte source: its kernel build now (startup), not inside the first RUN task
                self._fx_plan.warm_bytes()
            self._rot_key = None
        key = (self.bandwidth, self.frequency, self.calibrated_delay)
        if key != self._rot_key:      # rot only changes on calibration / TEST sweep
            self._fx_plan.set_rot(rot_table(int(self.nbins), self.bandwidth, self.frequency, self.calibrated_delay))
            self._rot_key = key
        return self._fx_plan

    def _pfb_xcorr(self):
        """effex.py:497-527 — one visibility from the chunk pair in ``gpu_iq_0`` / ``gpu_iq_1``."""
        plan = self._plan()
        u8 = getattr(self, "_u8_pair", None)
        if u8 is not None:          # byte source: convert + de-mean + F+X in one device call
            if self.mode in ('CONTINUUM', 'TEST'):
                out = self._row_buf((1, 1), np.complex128)
                return plan.fx_rows_u8(u8, 'CONTINUUM', self.bandwidth, remove_dc=self.remove_dc, out=out)[0, 0]
            out = self._row_buf((1, 1, int(self.nbins)), np.complex64)
            return plan.fx_rows_u8(u8, 'SPECTRUM', remove_dc=self.remove_dc, out=out)[0, 0].astype(np.complex128)
        # the pinned pair goes over PCIe by DMA, the row comes back written by the device into a pinned row buffer; a pair
        # _stage left there and nobody has looked at since is de-meaned on the device on the way (effex.py:394-395)
        dc = self._dc_pending
        pair = self._staged_pair()
        if self.mode in ('CONTINUUM', 'TEST'):
            out = self._row_buf((1, 1), np.complex128)
            return plan.fx_rows(pair, 'CONTINUUM', self.bandwidth, remove_dc=dc, out=out)[0, 0]
        out = self._row_buf((1, 1, int(self.nbins)), np.complex64)
        return plan.fx_rows(pair, 'SPECTRUM', remove_dc=dc, out=out)[0, 0].astype(np.complex128)

    def _run_task(self):
        """effex.py:490-494."""
        return self._pfb_xcorr()

    # -- delay calibration (effex.py:476-487, 558-627) --------------------------------------
    def _estimate_delay_gaussian(self, iq_0, iq_1, rate):
        """effex.py:583-627 — sub-sample delay between the channels in seconds (FFT cross-correlation,
        arg-max, 3-point log-Gaussian peak), computed on the device."""
        assert len(iq_0) == len(iq_1), ('Algorithm assumes input complex timeseries are of equal length.')
        return self._plan().estimate_delay(iq_0, iq_1, rate)

    def _estimate_delay(self, iq_0, iq_1, rate):
        """effex.py:558-580."""
        total_delay = self._estimate_delay_gaussian(iq_0, iq_1, rate)
        if self.mode in ['TEST']:
            total_delay -= self.test_delay_offset
        return total_delay

    def _calibrate_task(self):
        """effex.py:476-487 — estimate and store the delay from the chunk pair currently staged."""
        self.calibrated_delay = self._estimate_delay(self.gpu_iq_0, self.gpu_iq_1, self.bandwidth)
        self.logger.info('Estimated delay (us): {}'.format(1e6 * self.calibrated_delay))

    def integrate(self, chunks):
        """Build extension: integrate a whole batch [n_chunks, 2, num_samp] (CUDA tensor
        or host array) into one visibility spectrum / scalar — same definition as averaging the
        reference's rows."""
        plan = self._plan()
        plan.acc_reset()
        plan.fx_accumulate(chunks)
        mode = 'CONTINUUM' if self.mode in ('CONTINUUM', 'TEST') else 'SPECTRUM'
        out = plan.finalize(mode, self.bandwidth)
        return out[0]

    # -- output (effex.py:667-696) ----------------------------------------------------------
    def _header_line(self):
        return rowsink.header_line(self.run_time, self.bandwidth, self.frequency, self.num_samp, self.nbins, self.gain,
                                   self.mode)

    def _write_metadata(self):
        with open(self.output_file, 'w') as fh:
            fh.write(self._header_line() + '\n')
            if 'SPECTRUM' == self.mode:
                np.savetxt(fh, [rowsink.spectrum_freqs(self.nbins, self.bandwidth, self.frequency)], delimiter=',')

    def _open_bin_sink(self):
        spectrum = 'SPECTRUM' == self.mode
        return rowsink.BinSink(self.output_file, self._header_line(),
                               rowsink.spectrum_freqs(self.nbins, self.bandwidth, self.frequency) if spectrum else None,
                               int(self.nbins) if spectrum else 1, np.complex64 if spectrum else np.complex128)

    def _write_row(self, fh, vis):
        np.savetxt(fh, [np.asarray(vis, dtype=np.complex128)], delimiter=',')

    # -- control loop (host orchestration only; effex.py:326-417 without the hardware) ------
    def _stage(self, pair):
        iq_0, iq_1 = pair
        self._u8_pair = None
        if np.asarray(iq_0).dtype == np.uint8:
            # raw receiver bytes [num_samp, 2]: the RUN state hands them to the device as they are; CALIBRATE (one
            # chunk pair per run) needs samples, converted as pyrtlsdr does (effex.py:652)
            b0 = np.ascontiguousarray(iq_0, dtype=np.uint8).reshape(-1, 2)
            b1 = np.ascontiguousarray(iq_1, dtype=np.uint8).reshape(-1, 2)
            if 'CALIBRATE' != self.state:
                n = len(b0)
                buf = getattr(self, "_u8_buf", None)
                if buf is None or buf.shape[2] != n:
                    buf = self._u8_buf = _staging_empty((1, 2, n, 2), np.uint8)[0]
                buf[0, 0], buf[0, 1] = b0, b1
                self._u8_pair = buf
                return
            iq_0 = ((b0[:, 0].astype(np.float64) - 127.5) + 1j * (b0[:, 1].astype(np.float64) - 127.5)) / 127.5
            iq_1 = ((b1[:, 0].astype(np.float64) - 127.5) + 1j * (b1[:, 1].astype(np.float64) - 127.5)) / 127.5
        if 'CALIBRATE' == self.state:
            # one chunk pair per run, read by the delay estimate on the host side of the ABI: de-meaned as effex.py:394-395
            self.gpu_iq_0 = _without_mean(iq_0) if self.remove_dc else np.asarray(iq_0)
            self.gpu_iq_1 = _without_mean(iq_1) if self.remove_dc else np.asarray(iq_1)
            return
        # RUN: one narrowing pass per stream into the pinned pair; the mean (effex.py:394-395) comes off on the device,
        # in front of the F+X kernels, when _pfb_xcorr hands the pair over
        n = len(iq_0)
        if self._pair_buf.shape[2] != n:
            self._alloc_staging(n)
        for a, iq in enumerate((iq_0, iq_1)):
            np.copyto(self._pair_buf[0, a], np.asarray(iq).reshape(n), casting='same_kind')
        self._gpu_iq = [self._pair_buf[0, 0], self._pair_buf[0, 1]]
        self._dc_pending = bool(self.remove_dc)

    def _run_batched(self, first_pair, sink, fh):
        """The RUN state for ``batch`` > 1: chunk pairs go ``batch`` at a time through a three-slot ``FxPipeline`` -- the
        source fills a pinned slot in place (``IQSource.read_into``), the copy in, the F+X call and the rows' copy out of
        successive batches overlap, and with the binary sidecar the rows land in a mapped window of the file.  Same rows
        as the one-call-per-pair loop up to float32 summation order (a batch is split over the workgroups differently);
        a last, short batch goes through one blocking call.  Returns the number of rows written."""
        from .plan import FxPipeline
        plan = self._plan()
        n, K = int(self.num_samp), self.batch
        u8 = np.asarray(first_pair[0]).dtype == np.uint8
        mode = 'CONTINUUM' if 'CONTINUUM' == self.mode else 'SPECTRUM'
        rows = 0

        def emit(out):          # out: [k, 1, nchan] complex64 or [k, 1] complex128
            for row in out[:, 0]:
                self._write_row(fh, row)

        def fill(view, pair):
            """view[0] from ``pair`` when there is one, the rest from the source; returns the chunk pairs filled."""
            k = 0
            if pair is not None:
                view[0, 0][...] = np.asarray(pair[0]).reshape(view[0, 0].shape)
                view[0, 1][...] = np.asarray(pair[1]).reshape(view[0, 1].shape)
                k = 1
            k += self.source.read_many_into(n, view[k:])
            return k

        # complex sources: samples are rounded to complex64 in the pinned slot and de-meaned on the device
        # (effex.py:394-395 through fxc_pipe_create_iq) -- no host pass over the samples
        with FxPipeline(plan, K, depth=3, mode=mode, bandwidth=self.bandwidth, u8=u8, remove_dc=self.remove_dc) as pipe:
            def drain():
                if sink is not None:
                    pipe.pop(out=sink.reserve(K))
                    sink.commit(K)
                else:
                    emit(pipe.pop())
                return K

            # the fill of a slot (file / socket -> pinned memory, no library call) runs beside the main thread, which
            # meanwhile collects the oldest batch in flight; every fxc_* call stays on the main thread
            import concurrent.futures
            filler = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            job = None
            try:
                pending, k = first_pair, 0
                while True:
                    view = pipe.acquire()
                    job = filler.submit(fill, view, pending)
                    pending = None
                    if pipe.in_flight == 2:
                        rows += drain()
                    k = job.result()
                    if k < K:
                        break
                    pipe.submit()
            except BaseException:
                # the filler may sit in a blocking read (a socket source) that writes into a pinned slot of the pipe:
                # closing the source ends that read; the pipe (and its slots) may only be freed once the read has
                # returned.  A read that does not come back within the bound keeps its memory: the pipe is abandoned
                # (leaked, with its plan) rather than freed under a live writer.
                self.source.close()
                filler.shutdown(wait=False, cancel_futures=True)
                if job is not None and not job.cancelled():
                    done, _ = concurrent.futures.wait([job], timeout=self._filler_grace_s)
                    if not done:
                        self.logger.error('the source did not return from its read within {} s of being closed: '
                                          'leaking the staging slots it writes into'.format(self._filler_grace_s))
                        pipe.abandon()
                        self._fx_plan = None      # (the plan object keeps the abandoned handle alive; close() skips it)
                raise
            filler.shutdown(wait=True)
            while pipe.in_flight:
                rows += drain()
            if k:               # the stream ended inside a batch
                tail = view[:k]
                out = (plan.fx_rows_u8(tail, mode, self.bandwidth, remove_dc=self.remove_dc) if u8
                       else plan.fx_rows(tail, mode, self.bandwidth, remove_dc=self.remove_dc))
                if sink is not None:
                    sink.write_rows(out[:, 0])
                else:
                    emit(out)
                rows += k
        return rows

    def run_state_machine(self):
        """OFF -> STARTUP -> CALIBRATE -> RUN ... -> SHUTDOWN -> OFF over the source's chunk pairs; the first
        pair calibrates the delay (unless ``calibrate=False``), every further pair writes one csv row."""
        rows = 0
        fh = sink = None
        try:
            while True:
                if 'OFF' == self.state:
                    self.state = 'STARTUP'
                elif 'STARTUP' == self.state:
                    if 'bin' == self.output_format:
                        sink = self._open_bin_sink()
                    else:
                        self._write_metadata()
                        fh = open(self.output_file, 'a')
                    self.start_time = time.time()
                    self.state = 'CALIBRATE' if self.calibrate else 'RUN'
                elif self.state in ('CALIBRATE', 'RUN'):
                    pair = self.source.read(int(self.num_samp))
                    if pair is None:
                        self.state = 'SHUTDOWN'
                        continue
                    if 'RUN' == self.state and self.batch > 1 and self.mode not in ['TEST']:
                        rows += self._run_batched(pair, sink, fh)      # to the end of the stream
                        self.state = 'SHUTDOWN'
                        continue
                    self._stage(pair)
                    if 'CALIBRATE' == self.state:      # consumes a chunk pair, writes no row (effex.py:399-401)
                        self._calibrate_task()
                        self.state = 'RUN'
                        continue
                    if self.mode in ['TEST']:
                        self.calibrated_delay += self.test_delay_sweep_step      # effex.py:403-404
                    if sink is not None:
                        sink.write(self._run_task())
                    else:
                        self._write_row(fh, self._run_task())
                    rows += 1
                elif 'SHUTDOWN' == self.state:
                    self.close()
                    self._state = 'OFF'
                    break
        finally:
            if fh is not None:
                fh.close()
            if sink is not None:
                sink.close()
        return rows
