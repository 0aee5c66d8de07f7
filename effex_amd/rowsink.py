"""Visibility row sinks: the reference's ``.csv`` and a binary sidecar for throughput.

The reference writes one text row per chunk pair — ``np.savetxt(fh, [row], delimiter=',')`` of 4096 complex numbers
as ``%.18e`` pairs, ≈ 200 KiB and ≈ 10 ms of formatting per row (``/root/reference/effex/effex.py:667-696``; readers
``effex.py:798``, ``post_process.py:201-219``).  The device path produces rows three to four orders of magnitude faster
than that, so next to the byte-exact csv (``CsvSink``) there is ``BinSink``:

    line 1      the csv's own header line, text, '\\n'-terminated  (effex.py:672-678)
    preamble    b"FXB2", uint32 bytes per element (8 = complex64, 16 = complex128), uint64 elements per row,
                uint64 data_offset (absolute, 64-byte aligned), uint64 n_freqs, uint64 rows committed
    freqs       float64[n_freqs]   — the csv's second line in SPECTRUM mode (effex.py:679-682), else n_freqs = 0
    (padding to data_offset)
    rows        row after row, native little-endian, no separators

A reader trusts the committed-row count, not the file size: the file is extended *before* rows are written through a
mapped window (``reserve``) or by several ranks at once (``create_shared``), and a live reader or a file left behind by a
killed writer must not show the zero-filled tail as visibilities.  (``FXB1`` files, without the count, are still read by
size.)

Several ranks, one file (the reference-faithful time-series mode: "ranks own disjoint rows"):
``create_shared`` (one rank) sizes the file for all rows, every rank maps its own ``RowWindow`` of it and fills it in
place, ``commit_shared`` (one rank, after a barrier) publishes the count -- no collective on the data path.

Rows can be written one at a time, a batch at a time, or by filling a memory-mapped window of the file in place
(``reserve`` / ``commit``: ``FxPipeline.pop(out=...)`` copies straight from the pinned result slot into the page cache).
``to_csv`` (``tools/rows_to_csv.py``) turns a sidecar back into exactly the bytes the reference's writer would have
produced for the same rows, so ``post_process.py``-style readers keep working.
"""
import os
import threading
import time
import struct

import numpy as np

MAGIC = b"FXB2"
MAGIC_V1 = b"FXB1"                       # round-3 files: no committed-row count, rows = what the file size holds
_PRE = struct.Struct("<4sIQQQQ")
_PRE_V1 = struct.Struct("<4sIQQQ")
_COUNT_OFFSET_IN_PRE = _PRE.size - 8     # the committed-row count is the preamble's last field


def header_line(run_time, bandwidth, frequency, num_samp, resolution, gain, mode):
    """The csv's first line without the newline — effex.py:672-678 (Python ``str()`` of the values)."""
    fields = (('run_time', run_time), ('bandwidth', bandwidth), ('frequency', frequency), ('num_samp', num_samp),
              ('resolution', resolution), ('gain', gain), ('mode', mode))
    return ','.join('{}:{}'.format(k, v) for k, v in fields)


def spectrum_freqs(nbins, bandwidth, frequency):
    """The csv's second line in SPECTRUM mode — effex.py:679-682."""
    return np.fft.fftshift(np.fft.fftfreq(int(nbins), d=1 / bandwidth)) + frequency


class CsvSink(object):
    """The reference's writer: header, frequency row (SPECTRUM), then ``np.savetxt`` of one complex128 row per call."""

    def __init__(self, path, header, freqs=None):
        self.path = path
        with open(path, 'w') as fh:
            fh.write(header + '\n')
            if freqs is not None:
                np.savetxt(fh, [np.asarray(freqs, dtype=np.float64)], delimiter=',')
        self._fh = open(path, 'a')
        self.rows = 0

    def write(self, row):
        np.savetxt(self._fh, [np.atleast_1d(np.asarray(row, dtype=np.complex128))], delimiter=',')
        self.rows += 1

    def write_rows(self, rows):
        for row in np.asarray(rows):
            self.write(row)

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class BinSink(object):
    """The binary sidecar (format in the module docstring)."""

    def __init__(self, path, header, freqs, row_len, row_dtype=np.complex64):
        self.path = path
        self.row_len = int(row_len)
        self.row_dtype = np.dtype(row_dtype)
        if self.row_dtype not in (np.dtype(np.complex64), np.dtype(np.complex128)):
            raise ValueError("row_dtype must be complex64 or complex128")
        freqs = np.zeros(0) if freqs is None else np.ascontiguousarray(freqs, dtype=np.float64)
        head = (header + '\n').encode()
        unaligned = len(head) + _PRE.size + freqs.nbytes
        self.data_offset = (unaligned + 63) // 64 * 64
        self._count_at = len(head) + _COUNT_OFFSET_IN_PRE
        self._fh = open(path, 'w+b')
        self._fh.write(head)
        self._fh.write(_PRE.pack(MAGIC, self.row_dtype.itemsize, self.row_len, self.data_offset, freqs.size, 0))
        self._fh.write(freqs.tobytes())
        self._fh.write(b'\0' * (self.data_offset - unaligned))
        self._fh.flush()
        self.rows = 0                 # rows committed
        self._published_at = 0.0
        self._published_rows = 0
        self._timer = None            # a pending late publication (write_rows)
        self._lock = threading.RLock()
        self._map = None
        self._map_rows = 0            # rows the file is currently sized for beyond `rows`
        self.row_bytes = self.row_len * self.row_dtype.itemsize

    def write(self, row):
        self.write_rows(np.asarray(row).reshape(1, -1))

    def write_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=self.row_dtype).reshape(-1, self.row_len)
        with self._lock:
            self._write_rows_locked(rows)

    def _write_rows_locked(self, rows):
        self._drop_map()
        self._fh.seek(self.data_offset + self.rows * self.row_bytes)
        self._fh.write(rows.view(np.uint8).data)
        self.rows += len(rows)
        # the per-row writer path (one visibility per call, effex.py:689-693) publishes the count at most every 0.1 s -- the
        # reference's own writer wakes that often -- instead of two system calls per row; batches and close() always publish
        now = time.monotonic()
        if len(rows) > 1 or now - self._published_at >= 0.1:
            self._publish()
        elif self._timer is None:
            # ... and a row whose publication was put off does not wait for the NEXT row to arrive (a source that stalls would leave
            # it in this process's buffer, its count stale, for as long as the stall lasts): a one-shot timer publishes it 0.1 s on
            self._timer = threading.Timer(0.1, self._publish_late)
            self._timer.daemon = True
            self._timer.start()

    def _publish_late(self):
        with self._lock:
            self._timer = None
            if self._fh is not None and self.rows > self._published_rows:
                self._publish()

    def flush(self):
        """Publish every committed row now (the Correlator calls it when its source blocks)."""
        with self._lock:
            if self._fh is not None and self.rows > self._published_rows:
                self._publish()

    def _publish(self):
        """The committed-row count goes into the preamble behind the rows it counts (same file object: ordered)."""
        self._fh.flush()
        os.pwrite(self._fh.fileno(), struct.pack("<Q", self.rows), self._count_at)
        self._published_at = time.monotonic()
        self._published_rows = self.rows

    def reserve(self, n_rows):
        """A writable [n_rows, row_len] view of the file just behind the committed rows (the file grows to hold it);
        fill it in place, then ``commit(k)`` the first k rows."""
        self._drop_map()
        n_rows = int(n_rows)
        start = self.data_offset + self.rows * self.row_bytes
        self._fh.flush()
        os.ftruncate(self._fh.fileno(), start + n_rows * self.row_bytes)
        self._map = np.memmap(self._fh, dtype=self.row_dtype, mode='r+', offset=start, shape=(n_rows, self.row_len))
        self._map_rows = n_rows
        return self._map

    def commit(self, n_rows):
        if n_rows > self._map_rows:
            raise ValueError("commit of {} rows, {} reserved".format(n_rows, self._map_rows))
        self.rows += int(n_rows)
        self._map_rows -= int(n_rows)
        if self._map_rows == 0:
            self._drop_map()
        self._publish()

    def _drop_map(self):
        # no msync here: rows written through the window are in the page cache, where every reader of the file sees them
        # (and only counts them once ``commit`` has published them); forcing them to the disk after every batch cost more
        # than producing them
        self._map = None
        self._map_rows = 0

    def close(self):
        timer, self._timer = self._timer, None
        if timer is not None:
            timer.cancel()
        with self._lock:
            self._close_locked()

    def _close_locked(self):
        if self._fh is not None:
            self._drop_map()
            self._fh.flush()
            os.ftruncate(self._fh.fileno(), self.data_offset + self.rows * self.row_bytes)     # drop rows reserved, never committed
            self._publish()
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _read_preamble(fh):
    """-> (header line bytes, itemsize, row_len, data_offset, n_freqs, committed rows or None (FXB1), offset of the count)."""
    line = fh.readline()
    pre = fh.read(_PRE_V1.size)
    if len(pre) != _PRE_V1.size:
        raise ValueError("not a visibility sidecar (truncated)")
    magic, itemsize, row_len, data_offset, n_freqs = _PRE_V1.unpack(pre)
    committed = None
    if magic == MAGIC:
        tail = fh.read(8)
        if len(tail) != 8:
            raise ValueError("not a visibility sidecar (truncated)")
        committed = struct.unpack("<Q", tail)[0]
    elif magic != MAGIC_V1:
        raise ValueError("not a visibility sidecar")
    if itemsize not in (8, 16) or row_len < 1:
        raise ValueError("not a visibility sidecar")
    return line, itemsize, row_len, data_offset, n_freqs, committed, len(line) + _COUNT_OFFSET_IN_PRE


class RowFile(object):
    """A sidecar opened for reading: ``header`` (line 1), ``fields`` (its key:value pairs), ``freqs``, ``rows`` (memory map
    of the *committed* rows: what a writer has reserved or mapped but not yet committed is not shown)."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            try:
                line, itemsize, row_len, data_offset, n_freqs, committed, _ = _read_preamble(fh)
            except ValueError as exc:
                raise ValueError("{}: {}".format(path, exc))
            self.freqs = np.frombuffer(fh.read(8 * n_freqs), dtype=np.float64) if n_freqs else None
        self.header = line.decode().rstrip('\n')
        self.fields = dict(item.split(':', 1) for item in self.header.split(','))
        self.row_dtype = np.dtype(np.complex64 if itemsize == 8 else np.complex128)
        self.row_len = int(row_len)
        held = max(0, (os.path.getsize(path) - data_offset) // (self.row_len * itemsize))
        n_rows = held if committed is None else min(int(committed), held)
        self.rows = (np.memmap(path, dtype=self.row_dtype, mode='r', offset=data_offset, shape=(n_rows, self.row_len))
                     if n_rows > 0 else np.zeros((0, self.row_len), dtype=self.row_dtype))


# ------------------------------------------------------------------------------------------------------------------
# several writers, one file: disjoint row ranges, no collective on the data path (time-series mode)
# ------------------------------------------------------------------------------------------------------------------
def create_shared(path, header, freqs, row_len, row_dtype, n_rows):
    """One rank: the sidecar's head and room for ``n_rows`` rows, none of them committed yet."""
    sink = BinSink(path, header, freqs, row_len, row_dtype)
    size = sink.data_offset + int(n_rows) * sink.row_bytes
    os.ftruncate(sink._fh.fileno(), size)
    try:                                    # blocks allocated once, here, not one page fault (or extent) at a time under the writers
        os.posix_fallocate(sink._fh.fileno(), 0, size)
    except (AttributeError, OSError):
        pass
    sink._fh.close()
    sink._fh = None


class RowWindow(object):
    """Rows [lo, hi) of a sidecar made by ``create_shared``, mapped for writing: ``rows`` is [hi - lo, row_len]."""

    def __init__(self, path, lo, hi, mapped=True):
        lo, hi = int(lo), int(hi)
        self._fh = open(path, 'r+b')
        _, itemsize, row_len, data_offset, _, committed, _ = _read_preamble(self._fh)
        if committed is None:
            self._fh.close()
            raise ValueError("{}: an FXB1 sidecar has no row count to publish: shared writing needs the FXB2 layout".format(path))
        self.row_dtype = np.dtype(np.complex64 if itemsize == 8 else np.complex128)
        self.row_len = int(row_len)
        need = data_offset + hi * self.row_len * itemsize
        if hi < lo or lo < 0 or os.fstat(self._fh.fileno()).st_size < need:
            self._fh.close()
            raise ValueError("{}: rows [{}, {}) are outside the file".format(path, lo, hi))
        self.lo, self.hi = lo, hi
        self._base = data_offset + lo * self.row_len * itemsize
        self._row_bytes = self.row_len * itemsize
        # mapped=False: rows go in through write() (pwrite from the caller's -- pinned -- buffer; several threads may call it)
        self.rows = (np.memmap(self._fh, dtype=self.row_dtype, mode='r+', offset=self._base, shape=(hi - lo, self.row_len))
                     if (mapped and hi > lo) else np.zeros((0, self.row_len), self.row_dtype))

    def write(self, first, rows):
        """rows [n, row_len] -> rows [first, first + n) of the window (relative to its first row); thread-safe."""
        rows = np.ascontiguousarray(rows, dtype=self.row_dtype).reshape(-1, self.row_len)
        if first < 0 or first + rows.shape[0] > self.hi - self.lo:
            raise ValueError("rows [{}, {}) are outside the window".format(first, first + rows.shape[0]))
        buf = memoryview(rows).cast('B')
        at, done = self._base + first * self._row_bytes, 0
        while done < len(buf):
            done += os.pwrite(self._fh.fileno(), buf[done:], at + done)

    def close(self):
        if self._fh is not None:
            if isinstance(self.rows, np.memmap):
                self.rows.flush()      # this rank's rows are in the file before it tells the others so (the barrier)
            self.rows = None
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def commit_shared(path, n_rows):
    """One rank, after every writer has closed its window: publish the count."""
    with open(path, 'r+b') as fh:
        pre = _read_preamble(fh)
        if pre[5] is None:          # FXB1: those eight bytes are the frequency row's (or padding), not a count
            raise ValueError("{}: an FXB1 sidecar has no row count to publish".format(path))
        os.pwrite(fh.fileno(), struct.pack("<Q", int(n_rows)), pre[6])


def to_csv(path_in, path_out):
    """The csv the reference's writer produces for the sidecar's rows, byte for byte (effex.py:667-696).  Returns the
    number of rows."""
    src = RowFile(path_in)
    with CsvSink(path_out, src.header, src.freqs) as out:
        out.write_rows(src.rows)
        return out.rows
