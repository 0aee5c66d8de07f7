"""Parity bounds set from measurement.

Every comparison of the HIP path with the oracle or a reference golden goes through a ``Bound``: ``assert err < TOL_VIS``.
A bound has a *ceiling* -- the tolerance the north star states (1e-5 of max|vis|; 2e-6 for F-stage spectra) --
and, per test function, a *measured* value from ``tests/golden/tolerances.json``: twice the largest error that comparison
showed on an MI355X over the whole GPU suite (``tools/make_tolerances.py`` writes the file from the errors a suite run
records; the arithmetic is deterministic -- no atomics, fixed summation orders -- so a run reproduces them).  The bound in force
is the measured one where there is one, never above the ceiling; a test without an entry (a new test) runs against the
ceiling until the table is regenerated.  Every comparison is also recorded, so each GPU run leaves
``gpurun_out/observed_errors.json`` behind: observed error and bound per test and quantity.

FXC_TOL_MEASURE=1: compare against the ceilings only (the run that produces the numbers for the table).
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
TABLE_PATH = os.path.join(HERE, "golden", "tolerances.json")
FLOOR = 2.4e-7          # two float32 roundings of the reference value itself: below this a bound says nothing

_current = [None]       # the running test function's name (tests/conftest.py)
observed = {}           # test -> bound name -> largest error seen this session
_table = None


def table():
    global _table
    if _table is None:
        try:
            with open(TABLE_PATH) as fh:
                _table = json.load(fh)["bounds"]
        except (OSError, ValueError, KeyError):
            _table = {}
    return _table


class Bound(object):
    """``err < bound`` -> records err for the running test and compares it with the bound in force for that test."""
    __array_priority__ = 1000.0
    __array_ufunc__ = None          # numpy scalars on the left defer to the reflected comparison below

    def __init__(self, name, ceiling):
        self.name, self.ceiling = name, float(ceiling)

    def in_force(self):
        if os.environ.get("FXC_TOL_MEASURE") == "1":
            return self.ceiling
        measured = table().get(_current[0] or "", {}).get(self.name)
        return self.ceiling if measured is None else min(self.ceiling, float(measured))

    def _see(self, err):
        err = float(err)
        slot = observed.setdefault(_current[0] or "?", {})
        if not (slot.get(self.name, -1.0) >= err):       # (a NaN sticks)
            slot[self.name] = err
        return err

    def __gt__(self, err):          # err < bound
        return self._see(err) < self.in_force()

    def __ge__(self, err):          # err <= bound
        return self._see(err) <= self.in_force()

    def __float__(self):
        return self.in_force()

    def __lt__(self, other):        # (arrays against the bound in a failure message)
        return self.in_force() < other

    def __le__(self, other):
        return self.in_force() <= other

    def __mul__(self, scale):       # err_abs < bound * scale   (absolute error against a scaled bound)
        return _Scaled(self, float(scale))

    __rmul__ = __mul__

    def __repr__(self):
        return "%s(%.3g)" % (self.name, self.in_force())


class _Scaled(object):
    __array_ufunc__ = None

    def __init__(self, bound, scale):
        self.bound, self.scale = bound, scale

    def __gt__(self, err_abs):
        if self.scale <= 0.0:
            return False
        return self.bound > (float(err_abs) / self.scale)

    def __add__(self, slack):       # ... + an absolute slack term (a value that may be zero by symmetry)
        return _ScaledPlus(self, float(slack))


class _ScaledPlus(object):
    __array_ufunc__ = None

    def __init__(self, scaled, slack):
        self.scaled, self.slack = scaled, slack

    def __gt__(self, err_abs):
        return self.scaled > max(0.0, float(err_abs) - self.slack)


# the ceilings: what the north star states
TOL_VIS = Bound("TOL_VIS", 1e-5)            # visibilities against the float64 oracle, of max|vis|
TOL_SPEC = Bound("TOL_SPEC", 2e-6)          # F-stage spectra at the powers of two against the oracle, of max|spec|
TOL_SPEC_ANY = Bound("TOL_SPEC_ANY", 1e-5)  # F-stage spectra at any channel count (mixed radix, chirp-z rows)
TOL_CONT = Bound("TOL_CONT", 1e-5)          # CONTINUUM scalars, of |ref| (effex.py:523-524)
TOL_TONE = Bound("TOL_TONE", 5e-6)          # the reference test's tones: sampled spectra against the reference-executed goldens


def dump(path):
    rows = {}
    for test, per in sorted(observed.items()):
        rows[test] = {}
        for name, err in sorted(per.items()):
            _current[0] = test
            b = globals().get(name)
            rows[test][name] = {"observed": err, "bound": b.in_force() if isinstance(b, Bound) else None}
    _current[0] = None
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        json.dump({"measure_mode": os.environ.get("FXC_TOL_MEASURE") == "1", "floor": FLOOR, "tests": rows}, fh, indent=1, sort_keys=True)
