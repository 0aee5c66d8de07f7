"""Visibility row sinks: the reference's ``.csv`` and a binary sidecar for throughput.

The reference writes one text row per chunk pair — ``np.savetxt(fh, [row], delimiter=',')`` of 4096 complex numbers
as ``%.18e`` pairs, ≈ 200 KiB and ≈ 10 ms of formatting per row (``/root/reference/effex/effex.py:667-696``; readers
``effex.py:798``, ``post_process.py:201-219``).  The device path produces rows three to four orders of magnitude faster
than that, so next to the byte-exact csv (``CsvSink``) there is ``BinSink``:

    line 1      the csv's own header line, text, '\\n'-terminated  (effex.py:672-678)
    preamble    b"FXB1", uint32 bytes per element (8 = complex64, 16 = complex128), uint64 elements per row,
                uint64 data_offset (absolute, 64-byte aligned), uint64 n_freqs
    freqs       float64[n_freqs]   — the csv's second line in SPECTRUM mode (effex.py:679-682), else n_freqs = 0
    (padding to data_offset)
    rows        row after row, native little-endian, no separators

Rows can be written one at a time, a batch at a time, or by filling a memory-mapped window of the file in place
(``reserve`` / ``commit``: ``FxPipeline.pop(out=...)`` copies straight from the pinned result slot into the page cache).
``to_csv`` (``tools/rows_to_csv.py``) turns a sidecar back into exactly the bytes the reference's writer would have
produced for the same rows, so ``post_process.py``-style readers keep working.
"""
import os
import struct

import numpy as np

MAGIC = b"FXB1"
_PRE = struct.Struct("<4sIQQQ")


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
        self._fh = open(path, 'w+b')
        self._fh.write(head)
        self._fh.write(_PRE.pack(MAGIC, self.row_dtype.itemsize, self.row_len, self.data_offset, freqs.size))
        self._fh.write(freqs.tobytes())
        self._fh.write(b'\0' * (self.data_offset - unaligned))
        self._fh.flush()
        self.rows = 0                 # rows committed
        self._map = None
        self._map_rows = 0            # rows the file is currently sized for beyond `rows`
        self.row_bytes = self.row_len * self.row_dtype.itemsize

    def write(self, row):
        self.write_rows(np.asarray(row).reshape(1, -1))

    def write_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=self.row_dtype).reshape(-1, self.row_len)
        self._drop_map()
        self._fh.seek(self.data_offset + self.rows * self.row_bytes)
        self._fh.write(rows.view(np.uint8).data)
        self.rows += len(rows)

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

    def _drop_map(self):
        # no msync here: rows written through the window are in the page cache, where every reader of the file sees them;
        # forcing them to the disk after every batch cost more than producing them
        self._map = None
        self._map_rows = 0

    def close(self):
        if self._fh is not None:
            self._drop_map()
            self._fh.flush()
            os.ftruncate(self._fh.fileno(), self.data_offset + self.rows * self.row_bytes)     # drop rows reserved, never committed
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class RowFile(object):
    """A sidecar opened for reading: ``header`` (line 1), ``fields`` (its key:value pairs), ``freqs``, ``rows`` (memory map)."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            line = fh.readline()
            pre = fh.read(_PRE.size)
            if len(pre) != _PRE.size:
                raise ValueError("{}: not a visibility sidecar (truncated)".format(path))
            magic, itemsize, row_len, data_offset, n_freqs = _PRE.unpack(pre)
            if magic != MAGIC or itemsize not in (8, 16) or row_len < 1:
                raise ValueError("{}: not a visibility sidecar".format(path))
            self.freqs = np.frombuffer(fh.read(8 * n_freqs), dtype=np.float64) if n_freqs else None
        self.header = line.decode().rstrip('\n')
        self.fields = dict(item.split(':', 1) for item in self.header.split(','))
        self.row_dtype = np.dtype(np.complex64 if itemsize == 8 else np.complex128)
        self.row_len = int(row_len)
        n_rows = (os.path.getsize(path) - data_offset) // (self.row_len * itemsize)
        self.rows = (np.memmap(path, dtype=self.row_dtype, mode='r', offset=data_offset, shape=(n_rows, self.row_len))
                     if n_rows > 0 else np.zeros((0, self.row_len), dtype=self.row_dtype))


def to_csv(path_in, path_out):
    """The csv the reference's writer produces for the sidecar's rows, byte for byte (effex.py:667-696).  Returns the
    number of rows."""
    src = RowFile(path_in)
    with CsvSink(path_out, src.header, src.freqs) as out:
        out.write_rows(src.rows)
        return out.rows
