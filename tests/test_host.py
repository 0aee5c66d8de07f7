"""Host-side logic: window design, synthetic source, the Correlator shell (properties, state machine,
csv bytes).  Mirrors the reference's system-level tests (tests/test_effex.py:127-248) with a fake IQ
source instead of two RTL-SDRs.  No GPU compute is invoked."""
import hashlib
import io
import os

import numpy as np
import pytest

import golden_inputs as gi
from effex_amd import synth
from effex_amd.correlator import ArraySource, Correlator, SyntheticSource
from effex_amd.window import design_window


def test_window_sha256_matches_reference(golden):
    meta, _ = golden
    w = design_window(4, 4096)
    sha = hashlib.sha256(np.ascontiguousarray(w, dtype="<f8").tobytes()).hexdigest()
    # the closed form reproduces scipy to the last bit on this platform; tolerate a last-ulp platform
    # difference by falling back to a value check
    if sha != meta["window"]["4096_4"]["sha256"]:
        np.testing.assert_allclose(w[meta["window"]["4096_4"]["sample_idx"]], meta["window"]["4096_4"]["samples"],
                                   rtol=1e-12, atol=1e-24)


def test_synth_is_deterministic_and_sharded():
    a = synth.synth_iq(7, 3, 2, 1000)
    b = synth.synth_iq(7, 3, 2, 1000)
    np.testing.assert_array_equal(a, b)
    # a rank generating only its shard gets the same samples
    c = synth.synth_iq(7, 1, 2, 1000, first_chunk=2)
    np.testing.assert_array_equal(a[2:3], c)
    assert a.dtype == np.complex64 and np.abs(a).max() < 2.5
    # antenna streams share the delayed sky signal: correlation at lag d1 - d0 = 3
    x0, x1 = a[0, 0].astype(np.complex128), a[0, 1].astype(np.complex128)
    lag3 = np.abs(np.vdot(x0[:-3], x1[3:]))
    lag0 = np.abs(np.vdot(x0, x1))
    assert lag3 > 5 * lag0


# --- system-level (tests/test_effex.py:127-154) ------------------------------------------------
@pytest.fixture()
def cor(tmp_path):
    c = Correlator(source=SyntheticSource(n_chunks=2), output_file=str(tmp_path / "vis.csv"))
    yield c
    c.close()


def test_correlator_init_defaults(cor, golden):
    d = golden[0]["defaults"]
    assert cor.state == d["state"] == 'OFF'
    assert cor.mode == d["mode"] == 'SPECTRUM'
    assert cor.bandwidth == d["bandwidth"]
    assert cor.nbins == d["nbins"]
    assert cor.frequency == d["frequency"]
    assert cor.gain == d["gain"]
    assert cor.num_samp == d["num_samp"]
    assert cor.run_time == d["run_time"]
    assert cor.ntaps == d["ntaps"]
    assert list(Correlator._states) == d["states"]
    assert list(Correlator._modes) == d["modes"]
    np.testing.assert_allclose(cor.window, design_window(4, 4096))


def test_change_properties_reach_the_source(cor):
    cor.bandwidth = 1.2e6
    assert cor.bandwidth == 1.2e6 and cor.source.rs == 1.2e6
    cor.nbins = 2 ** 11
    assert cor.nbins == 2 ** 11
    cor.frequency = 1.3e9
    assert cor.frequency == 1.3e9 and cor.source.fc == 1.3e9
    cor.gain = 9.9
    assert cor.gain == 9.9 and cor.source.gain == 9.9


def test_num_samp_clamp(tmp_path):
    c = Correlator(num_samp=2 ** 20, source=SyntheticSource())
    assert c.num_samp == 2 ** 18                     # effex.py:282-283
    c2 = Correlator(num_samp=2 ** 20, max_num_samp=2 ** 20, source=SyntheticSource())
    assert c2.num_samp == 2 ** 20                    # documented deviation for the continuum config
    c3 = Correlator(num_samp=10, nbins=16, source=SyntheticSource())
    assert c3.num_samp == 2 ** 8


# --- state machine (tests/test_effex.py:157-219) -----------------------------------------------
def step_and_assert(cor, sequence):
    for state in sequence:
        cor.state = state
        assert state == cor.state


def test_nominal_state_transitions(cor):
    step_and_assert(cor, ['STARTUP', 'CALIBRATE', 'RUN', 'CALIBRATE', 'RUN', 'SHUTDOWN', 'OFF'])


def test_early_aborts(cor):
    step_and_assert(cor, ['STARTUP', 'SHUTDOWN', 'OFF', 'STARTUP', 'CALIBRATE', 'SHUTDOWN', 'OFF'])


@pytest.mark.parametrize("path,bad", [
    ([], 'RUN'), ([], 'CALIBRATE'), ([], 'SHUTDOWN'), ([], 'OFF'),
    (['STARTUP'], 'OFF'), (['STARTUP'], 'STARTUP'),
    (['STARTUP', 'RUN'], 'OFF'), (['STARTUP', 'RUN'], 'STARTUP'), (['STARTUP', 'RUN'], 'RUN'),
    (['STARTUP', 'CALIBRATE'], 'OFF'), (['STARTUP', 'CALIBRATE'], 'STARTUP'), (['STARTUP', 'CALIBRATE'], 'CALIBRATE'),
    (['STARTUP', 'SHUTDOWN'], 'RUN'),
])
def test_bad_transitions(cor, path, bad):
    step_and_assert(cor, path)
    with pytest.raises(Correlator.StateTransitionError):
        cor.state = bad
    assert cor.source.closed           # the reference closes the SDRs before raising (effex.py:210-228)


def test_unknown_state(cor):
    with pytest.raises(ValueError):
        cor.state = 'BOGUS'


# --- off-nominal init (tests/test_effex.py:225-248) --------------------------------------------
def test_bad_run_time_init():
    with pytest.raises(ValueError):
        Correlator(run_time=0, source=SyntheticSource())


def test_bad_bandwidth_init_only_warns(caplog):
    c = Correlator(bandwidth=3e6, source=SyntheticSource())
    assert c.bandwidth == 3e6
    assert any("greater than" in r.message for r in caplog.records)


def test_bad_mode_init():
    with pytest.raises(ValueError):
        Correlator(mode='nonsense', source=SyntheticSource())


def test_alt_mode_init():
    for mode in ('continuum', 'Spectrum', 'TEST'):
        assert Correlator(mode=mode, source=SyntheticSource()).mode == mode.upper()


def test_too_short_chunk_asserts():
    with pytest.raises(AssertionError):
        Correlator(num_samp=2 ** 12, nbins=2 ** 12, source=SyntheticSource())     # n_int = S//4//N < 1


# --- csv header bytes (effex.py:667-684) -------------------------------------------------------
@pytest.mark.parametrize("mode", ["SPECTRUM", "CONTINUUM"])
def test_csv_bytes_match_reference(tmp_path, golden, mode):
    meta, _ = golden
    path = str(tmp_path / "vis.csv")
    c = Correlator(mode=mode, nbins=gi.CSV_NBINS, num_samp=gi.CSV_S, source=SyntheticSource(), output_file=path)
    c._write_metadata()
    with open(path, 'a') as fh:
        c._write_row(fh, gi.csv_row(mode))
    assert open(path).read() == meta["csv"][mode]
    skip = 2 if mode == "SPECTRUM" else 1                 # post_process.py:206-209
    back = np.loadtxt(path, dtype=np.complex128, delimiter=',', skiprows=skip)
    np.testing.assert_allclose(np.atleast_1d(back), gi.csv_row(mode), rtol=1e-15)


def test_default_output_name():
    c = Correlator(source=SyntheticSource())
    assert c.output_file.startswith('visibilities_') and c.output_file.endswith('.csv')


def test_file_source_reads_chunk_pairs(tmp_path):
    """FileSource (recorded streams instead of the reference's live dongles, effex.py:81-82, 630-664): chunk by chunk,
    bytes stay bytes, a trailing partial chunk is dropped."""
    from effex_amd.correlator import FileSource
    rng = np.random.default_rng(3)
    raw = [rng.integers(0, 256, size=(1000 * 3 + 17, 2), dtype=np.uint8) for _ in range(2)]
    for a in range(2):
        raw[a].tofile(str(tmp_path / ("rx%d.u8" % a)))
        (raw[a][:, 0] + 1j * raw[a][:, 1]).astype(np.complex64).tofile(str(tmp_path / ("rx%d.c64" % a)))
    src = FileSource(str(tmp_path / "rx0.u8"), str(tmp_path / "rx1.u8"), fmt='u8', rs=2.4e6)
    assert src.n_samples == 3017 and src.rs == 2.4e6
    for c in range(3):
        b0, b1 = src.read(1000)
        assert b0.dtype == np.uint8 and b0.shape == (1000, 2)
        np.testing.assert_array_equal(b0, raw[0][1000 * c:1000 * (c + 1)])
        np.testing.assert_array_equal(b1, raw[1][1000 * c:1000 * (c + 1)])
    assert src.read(1000) is None
    src.close()
    assert src.closed
    src = FileSource(str(tmp_path / "rx0.c64"), str(tmp_path / "rx1.c64"), fmt='c64')
    z0, z1 = src.read(2000)
    assert z0.dtype == np.complex64 and len(z1) == 2000 and z1[5] == raw[1][5, 0] + 1j * raw[1][5, 1]
    assert src.read(2000) is None
    with pytest.raises(ValueError):
        FileSource(str(tmp_path / "rx0.u8"), str(tmp_path / "rx1.u8"), fmt='s16')


def test_sources_fill_staging_windows_in_place(tmp_path):
    """IQSource.read_into (what the batched run feeds its pinned slots with): file and in-memory sources write the chunk
    pairs read() returns into caller-owned windows, and report the end of the stream the same way."""
    from effex_amd.correlator import ArraySource, FileSource
    rng = np.random.default_rng(4)
    n, n_chunks = 96, 3
    raw = rng.integers(0, 256, size=(2, n_chunks * n + 10, 2), dtype=np.uint8)
    for a in range(2):
        raw[a].tofile(str(tmp_path / ("rx%d.u8" % a)))
    cplx = (rng.standard_normal((n_chunks, 2, n)) + 1j * rng.standard_normal((n_chunks, 2, n))).astype(np.complex64)
    for a in range(2):
        cplx[:, a].reshape(-1).tofile(str(tmp_path / ("rx%d.c64" % a)))
    cases = [(lambda: FileSource(str(tmp_path / "rx0.u8"), str(tmp_path / "rx1.u8"), fmt='u8'), np.uint8, (n, 2)),
             (lambda: FileSource(str(tmp_path / "rx0.c64"), str(tmp_path / "rx1.c64"), fmt='c64'), np.complex64, (n,)),
             (lambda: ArraySource(cplx), np.complex64, (n,))]
    for make, dtype, shape in cases:
        by_read, by_fill = make(), make()
        slot = np.zeros((n_chunks + 1, 2) + shape, dtype=dtype)         # windows of one larger buffer, like a staging slot
        for c in range(n_chunks):
            pair = by_read.read(n)
            assert by_fill.read_into(n, slot[c, 0], slot[c, 1]) is True
            np.testing.assert_array_equal(slot[c, 0], np.asarray(pair[0]).reshape(shape))
            np.testing.assert_array_equal(slot[c, 1], np.asarray(pair[1]).reshape(shape))
        assert by_read.read(n) is None
        assert by_fill.read_into(n, slot[n_chunks, 0], slot[n_chunks, 1]) is False
        assert not slot[n_chunks].any()
        many = make()                                                   # a batch window at a time; a short last batch
        batch = np.zeros((2, 2) + shape, dtype=dtype)
        assert many.read_many_into(n, batch) == 2
        np.testing.assert_array_equal(batch, slot[:2])
        batch[...] = 0
        assert many.read_many_into(n, batch) == n_chunks - 2
        np.testing.assert_array_equal(batch[0], slot[2])
        assert not batch[1].any() and many.read_many_into(n, batch) == 0
        for src in (by_read, by_fill, many):
            src.close()


class _FakePipe(object):
    """The host-visible behaviour of FxPipeline (fxc_pipe_*): ``depth`` pinned slots, acquire refuses when all are in flight,
    rows come back in submission order.  A 'row' is the per-stream byte / sample sum, so order and content are checkable."""
    made = []

    def __init__(self, plan, chunks_per_batch, depth=2, mode="SPECTRUM", bandwidth=1.0, u8=False, remove_dc=True):
        self.chunks, self.depth, self.mode, self.u8 = int(chunks_per_batch), int(depth), mode, u8
        shape = (self.chunks, 2, plan.num_samp) + ((2,) if u8 else ())
        self.slots = [np.zeros(shape, dtype=np.uint8 if u8 else np.complex64) for _ in range(self.depth)]
        self.queue, self.pushed, self.popped, self.plan = [], 0, 0, plan
        self.max_in_flight = 0
        _FakePipe.made.append(self)

    @property
    def in_flight(self):
        return self.pushed - self.popped

    def acquire(self):
        assert self.in_flight < self.depth, "acquire with every slot in flight"
        return self.slots[self.pushed % self.depth]

    def submit(self):
        assert self.in_flight < self.depth
        self.queue.append(self.plan.rows_of(self.slots[self.pushed % self.depth], self.mode))
        self.pushed += 1
        self.max_in_flight = max(self.max_in_flight, self.in_flight)

    def pop(self, out=None):
        assert self.in_flight > 0
        rows = self.queue.pop(0)
        self.popped += 1
        if out is None:
            return rows
        out.reshape(rows.shape)[...] = rows
        return out

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _FakePlan(object):
    def __init__(self, num_samp, nchan):
        self.num_samp, self.nchan, self.tail_calls = num_samp, nchan, 0

    def rows_of(self, batch, mode):
        sums = batch.reshape(batch.shape[0], -1).astype(np.complex128).sum(axis=1)
        if mode == "SPECTRUM":
            return (sums[:, None, None] * np.ones((1, 1, self.nchan))).astype(np.complex64)
        return sums[:, None].astype(np.complex128)

    def fx_rows_u8(self, x, mode, bandwidth, remove_dc=True):
        self.tail_calls += 1
        return self.rows_of(np.asarray(x), mode)

    def fx_rows(self, x, mode, bandwidth):
        self.tail_calls += 1
        return self.rows_of(np.asarray(x), mode)

    def close(self):
        pass


@pytest.mark.parametrize("mode,fmt,n_chunks,batch", [("SPECTRUM", "bin", 11, 4), ("CONTINUUM", "bin", 9, 4), ("SPECTRUM", "csv", 7, 3),
                                                    ("CONTINUUM", "csv", 5, 8), ("SPECTRUM", "bin", 8, 4), ("SPECTRUM", "bin", 1, 4)])
def test_batched_run_control_flow(tmp_path, monkeypatch, mode, fmt, n_chunks, batch):
    """Correlator(batch=K) without a device (a stand-in pipeline and plan): every chunk pair of the recordings becomes one
    row, in order, whole batches through the pipeline (never more in flight than it has slots), the short last batch
    through one blocking call, the trailing partial chunk dropped, rows in the sidecar or the csv."""
    import effex_amd.plan as plan_module
    from effex_amd import rowsink
    from effex_amd.correlator import FileSource
    num_samp, nbins = 256, 16
    rng = np.random.default_rng(n_chunks * 10 + batch)
    raw = rng.integers(0, 256, size=(2, n_chunks * num_samp + 9, 2), dtype=np.uint8)
    for a in range(2):
        raw[a].tofile(str(tmp_path / ("rx%d.u8" % a)))
    fake = _FakePlan(num_samp, nbins)
    _FakePipe.made = []
    monkeypatch.setattr(plan_module, "FxPipeline", _FakePipe)
    monkeypatch.setattr(Correlator, "_plan", lambda self: fake)
    path = str(tmp_path / ("rows." + fmt))
    src = FileSource(str(tmp_path / "rx0.u8"), str(tmp_path / "rx1.u8"), fmt='u8')
    cor = Correlator(num_samp=num_samp, nbins=nbins, source=src, output_file=path, mode=mode, output_format=fmt, batch=batch,
                     calibrate=False)
    assert cor.num_samp == num_samp
    assert cor.run_state_machine() == n_chunks
    assert src.closed and cor.state == 'OFF'
    want = np.array([raw[:, c * num_samp:(c + 1) * num_samp].astype(np.float64).sum() for c in range(n_chunks)])
    if fmt == "bin":
        rf = rowsink.RowFile(path)
        got = np.asarray(rf.rows)[:, 0].real
    else:
        skip = 2 if mode == "SPECTRUM" else 1
        got = np.loadtxt(path, dtype=np.complex128, delimiter=',', skiprows=skip).reshape(n_chunks, -1)[:, 0].real
    np.testing.assert_allclose(got, want, rtol=1e-6)
    pipe, = _FakePipe.made
    assert pipe.chunks == batch and pipe.u8 and pipe.pushed == pipe.popped == n_chunks // batch
    assert pipe.max_in_flight <= pipe.depth
    assert fake.tail_calls == (1 if n_chunks % batch else 0)


@pytest.mark.parametrize("stuck", [False, True])
def test_batched_run_error_path_never_frees_slots_under_a_live_reader(tmp_path, monkeypatch, stuck):
    """Correlator._run_batched when the device side fails while the filler thread sits in the source's read: the source is
    closed, and the pipe (whose pinned slot the read writes into) is closed only once the read has returned; a read that
    does not come back within the grace period keeps its memory -- the pipe is abandoned, not freed."""
    import threading
    import effex_amd.plan as plan_module
    num_samp, nbins, batch = 256, 16, 4
    events = []
    release = threading.Event()

    class Source(object):
        rs = fc = gain = None
        closed = False

        def read(self, n):
            return np.zeros((n, 2), np.uint8), np.zeros((n, 2), np.uint8)

        def read_many_into(self, n, view):
            if getattr(self, "calls", 0) == 0:       # first batch fills at once, the second read blocks
                self.calls = 1
                return len(view)
            release.wait(timeout=30 if stuck else None)
            events.append("read returned")
            view[...] = 7                              # the write a freed slot must never see
            return len(view)

        def close(self):
            self.closed = True
            if not stuck:                              # a source that notices being closed ends its read
                release.set()

    class Pipe(_FakePipe):
        def close(self):
            events.append("pipe closed")            # (FxPipeline.close after abandon() frees nothing: its handle is gone)

        def abandon(self):
            events.append("pipe abandoned")

    fake = _FakePlan(num_samp, nbins)
    monkeypatch.setattr(plan_module, "FxPipeline", Pipe)
    monkeypatch.setattr(Correlator, "_plan", lambda self: fake)
    cor = Correlator(num_samp=num_samp, nbins=nbins, source=Source(), output_file=str(tmp_path / "rows.fxb"), output_format="bin",
                     batch=batch, calibrate=False)
    cor._filler_grace_s = 0.5
    # the second batch's read blocks; the main thread, waiting for it, fails (stands in for a failing device call)
    import concurrent.futures
    real_result = concurrent.futures.Future.result

    def result(self, timeout=None):
        try:
            return real_result(self, timeout=0.3)
        except concurrent.futures.TimeoutError:
            raise RuntimeError("device call failed")
    monkeypatch.setattr(concurrent.futures.Future, "result", result)
    with pytest.raises(RuntimeError, match="device call failed"):
        cor.run_state_machine()
    assert cor.source.closed
    if stuck:
        assert events == ["pipe abandoned", "pipe closed"], events       # given up before anything could be freed under the read
    else:
        assert events[0] == "read returned" and "pipe abandoned" not in events and events[-1] == "pipe closed", events
    release.set()


def test_socket_source_reads_chunk_pairs_from_two_streams():
    """SocketSource (a network stream in place of effex.py:630-664's live dongles): two TCP streams of
    rtl_tcp-style bytes -- a 12-byte greeting, then interleaved uint8 I,Q -- read chunk pair by chunk pair whatever the
    senders' packet sizes; a stream that ends inside a chunk ends the run."""
    import socket
    import threading
    from effex_amd.correlator import SocketSource
    rng = np.random.default_rng(5)
    raw = [rng.integers(0, 256, size=(700 * 3 + 123, 2), dtype=np.uint8) for _ in range(2)]
    servers = []
    for a in range(2):
        srv = socket.socket()
        srv.bind(("127.0.0.1", 0))
        srv.listen(1)
        servers.append(srv)

    def serve(srv, data, piece):
        conn, _ = srv.accept()
        payload = b"RTL0" + bytes(8) + data.tobytes()
        for lo in range(0, len(payload), piece):       # odd-sized pieces: chunk boundaries fall inside packets
            conn.sendall(payload[lo:lo + piece])
        conn.close()
        srv.close()

    threads = [threading.Thread(target=serve, args=(servers[a], raw[a], 977 + 500 * a)) for a in range(2)]
    for t in threads:
        t.start()
    src = SocketSource([srv.getsockname() for srv in servers], fmt='u8', rs=2.4e6, skip=12, timeout=20)
    assert src.rs == 2.4e6
    for c in range(3):
        b0, b1 = src.read(700)
        assert b0.dtype == np.uint8 and b0.shape == (700, 2)
        np.testing.assert_array_equal(b0, raw[0][700 * c:700 * (c + 1)])
        np.testing.assert_array_equal(b1, raw[1][700 * c:700 * (c + 1)])
    assert src.read(700) is None                         # 123 samples left: a short stream ends the run
    src.close()
    assert src.closed
    for t in threads:
        t.join()
    a, b = socket.socketpair()                           # connected sockets handed over; complex64 samples
    c, d = socket.socketpair()
    z = (rng.standard_normal(64) + 1j * rng.standard_normal(64)).astype('<c8')
    b.sendall(z.tobytes())
    d.sendall((2 * z).tobytes())
    src = SocketSource([a, c], fmt='c64')
    z0, z1 = src.read(64)
    np.testing.assert_array_equal(z0, z)
    np.testing.assert_array_equal(z1, 2 * z)
    slot = np.zeros((2, 2, 64), dtype=np.complex64)      # read_into: straight into windows of a staging slot
    b.sendall((3 * z).tobytes())
    d.sendall((4 * z).tobytes())
    assert src.read_into(64, slot[1, 0], slot[1, 1]) is True
    np.testing.assert_array_equal(slot[1, 0], 3 * z)
    np.testing.assert_array_equal(slot[1, 1], 4 * z)
    assert not slot[0].any()
    b.sendall(z.tobytes()[:100])                         # ends inside a chunk
    b.close()
    d.close()
    assert src.read_into(64, slot[0, 0], slot[0, 1]) is False
    src.close()
    with pytest.raises(ValueError):
        SocketSource([("127.0.0.1", 1)], fmt='u8')


# --- binary row sink and its way back to the reference's csv ------------------------
@pytest.mark.parametrize("mode", ["SPECTRUM", "CONTINUUM"])
def test_binary_sidecar_round_trips_to_the_reference_csv(tmp_path, golden, mode):
    """rowsink.BinSink holds the csv's header line, the frequency row and the rows; tools/rows_to_csv.py must give back the
    bytes the reference's own writer produced (tests/golden csv, effex.py:667-696) -- rows kept as complex128 here because
    the golden row is; and, rows kept as the device's complex64, the bytes the csv sink writes for those rows."""
    import subprocess
    import sys
    from effex_amd import rowsink
    meta, _ = golden
    c = Correlator(mode=mode, nbins=gi.CSV_NBINS, num_samp=gi.CSV_S, source=SyntheticSource(), output_file=str(tmp_path / "x.csv"))
    freqs = rowsink.spectrum_freqs(c.nbins, c.bandwidth, c.frequency) if mode == "SPECTRUM" else None
    row = gi.csv_row(mode)
    fxb = str(tmp_path / "vis128.fxb")
    with rowsink.BinSink(fxb, c._header_line(), freqs, len(row), np.complex128) as sink:
        sink.write(row)
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rows_to_csv.py")
    out = str(tmp_path / "back.csv")
    subprocess.run([sys.executable, tool, fxb, out], check=True, capture_output=True)
    assert open(out).read() == meta["csv"][mode]
    # complex64 rows, several of them, three ways of writing: one by one, a batch, in place through a mapped window
    rows = np.stack([(row * (k + 1)).astype(np.complex64) for k in range(7)])
    fxb64, ref_csv = str(tmp_path / "vis64.fxb"), str(tmp_path / "ref.csv")
    with rowsink.BinSink(fxb64, c._header_line(), freqs, rows.shape[1], np.complex64) as sink:
        sink.write(rows[0])
        sink.write_rows(rows[1:3])
        view = sink.reserve(10)                     # more than will be committed: close() drops the rest
        view[:4] = rows[3:7]
        sink.commit(4)
        assert sink.rows == 7
    back = rowsink.RowFile(fxb64)
    assert back.header == c._header_line() and back.fields["mode"] == mode and back.rows.shape == rows.shape
    np.testing.assert_array_equal(np.asarray(back.rows), rows)
    if mode == "SPECTRUM":
        np.testing.assert_array_equal(back.freqs, freqs)
    with rowsink.CsvSink(ref_csv, c._header_line(), freqs) as sink:
        sink.write_rows(rows)
    assert rowsink.to_csv(fxb64, out) == 7
    assert open(out, 'rb').read() == open(ref_csv, 'rb').read()


def test_binary_sidecar_shows_committed_rows_only(tmp_path):
    """The file is extended before rows are committed (reserve): a reader must not take the zero-filled tail for
    visibilities -- the preamble carries the committed-row count, rewritten by commit / write_rows / close.  Round-3
    files (FXB1, no count) are still read by size."""
    import struct
    from effex_amd import rowsink
    path = str(tmp_path / "live.fxb")
    sink = rowsink.BinSink(path, "h:1", None, 4, np.complex64)
    view = sink.reserve(6)
    assert len(rowsink.RowFile(path).rows) == 0              # reserved, nothing committed
    view[:3] = 1
    sink.commit(3)
    assert np.asarray(rowsink.RowFile(path).rows).shape == (3, 4)
    view[3:5] = 2                                            # (a writer killed here leaves six rows' worth of file)
    assert len(rowsink.RowFile(path).rows) == 3
    sink.commit(2)
    sink.write(np.full(4, 5, np.complex64))                  # single rows publish the count at most every 0.1 s (and on close)
    assert len(rowsink.RowFile(path).rows) in (5, 6)
    sink.close()
    rows = np.asarray(rowsink.RowFile(path).rows)
    assert rows.shape == (6, 4) and (rows[:3] == 1).all() and (rows[3:5] == 2).all() and (rows[5] == 5).all()
    # an FXB1 file: the same head without the count
    old = str(tmp_path / "old.fxb")
    head = b"h:1\n"
    data_offset = (len(head) + rowsink._PRE_V1.size + 63) // 64 * 64
    with open(old, "wb") as fh:
        fh.write(head + rowsink._PRE_V1.pack(rowsink.MAGIC_V1, 8, 4, data_offset, 0))
        fh.write(b"\0" * (data_offset - len(head) - rowsink._PRE_V1.size))
        fh.write(rows[:2].tobytes())
    np.testing.assert_array_equal(np.asarray(rowsink.RowFile(old).rows), rows[:2])


def test_binary_sidecar_rejects_other_files(tmp_path):
    from effex_amd import rowsink
    p = tmp_path / "not.fxb"
    p.write_text("run_time:1\n" + "x" * 100)
    with pytest.raises(ValueError):
        rowsink.RowFile(str(p))


def test_bin_sink_publishes_a_lone_row_without_waiting_for_the_next(tmp_path):
    """The per-row writer (one visibility per call, effex.py:689-693) puts publications off by up to 0.1 s; a row written just after a
    publication must still reach a follower of the live file when the source then stalls: a one-shot timer publishes it, and
    flush() does so at once."""
    import time
    from effex_amd import rowsink
    path = str(tmp_path / "live.fxb")
    sink = rowsink.BinSink(path, rowsink.header_line(1, 2.4e6, 1.4204e9, 4096, 8, 49.6, "SPECTRUM"), None, 8)
    sink.write(np.full(8, 1, np.complex64))
    sink.write(np.full(8, 2, np.complex64))          # within 0.1 s of the first: not published by the write itself
    assert len(rowsink.RowFile(path).rows) == 1
    time.sleep(0.35)                                 # the source stalls
    rows = np.asarray(rowsink.RowFile(path).rows)
    assert rows.shape[0] == 2 and rows[1, 0] == 2
    sink.write(np.full(8, 3, np.complex64))
    sink.flush()
    assert len(rowsink.RowFile(path).rows) == 3
    sink.close()
    assert len(rowsink.RowFile(path).rows) == 3
