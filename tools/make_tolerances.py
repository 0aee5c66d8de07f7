#!/usr/bin/env python3
"""tests/golden/tolerances.json from the errors a GPU suite run recorded (gpurun_out/observed_errors.json, written by
tests/conftest.py): per test function and bound, twice the largest error that comparison showed -- rounded up to two digits,
not below tests/tolerances.py::FLOOR, never above the bound's ceiling.  Also prints the per-family table of DESIGN.md.

    gpurun -- 'FXC_TOL_MEASURE=1 python -m pytest tests -q -m gpu'      # ceilings only, records the errors
    python tools/make_tolerances.py [gpurun_out/observed_errors.json]
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tolerances  # noqa: E402

# test function -> the path family its comparisons exercise (for the table in DESIGN.md)
FAMILIES = [
    ("fused (nchan 4096, 4 taps: fx_fused4096_kernel)", ("test_drop_in_run_task", "test_fused", "test_headline", "test_batched_integration", "test_dc_removal_inside_the_fused",
                                                         "test_drop_in_stage", "test_fx_straight_from_rtlsdr", "test_integration_doc", "test_documented", "test_eight_antennas", "test_x_engine")),
    ("tiled / wave-local (nchan 16 .. 8192 powers of two)", ("test_tiled", "test_small", "test_other_nfft", "test_drop_in_any_resolution", "test_drop_in_nbins", "test_prefilter", "test_channelize_matches")),
    ("specialised per channel count (fx_spec.h, hiprtc)", ("test_specialised",)),
    ("any-shape mixed radix / chirp-z (k_generic.h)", ("test_any_channel_count", "test_mixed_radix", "test_powers_of_two_no_tuned")),
    ("continuum streaming limit (nchan 1)", ("test_stream", "test_continuum")),
]


def round_up(x, digits=2):
    if x <= 0:
        return 0.0
    e = math.floor(math.log10(x)) - (digits - 1)
    return math.ceil(x / 10 ** e - 1e-9) * 10 ** e


def unseeded_draws():
    """Lines of the test suite that draw random inputs without a seed: a bound measured over such a run does not reproduce."""
    import re
    bad = []
    tests = os.path.join(ROOT, "tests")
    for name in sorted(os.listdir(tests)):
        if not name.endswith(".py"):
            continue
        lines = open(os.path.join(tests, name)).read().split("\n")
        for i, line in enumerate(lines):
            code = line.split("#")[0]
            if re.search(r"torch\.(randint|randn|rand|normal)\(", code) and "generator=" not in code + lines[min(i + 1, len(lines) - 1)]:
                bad.append("%s:%d: %s" % (name, i + 1, line.strip()))
            if re.search(r"np\.random\.(rand|randn|randint|random|normal|uniform)\(", code) or re.search(r"default_rng\(\s*\)", code):
                bad.append("%s:%d: %s" % (name, i + 1, line.strip()))
    return bad


def main():
    bad = unseeded_draws()
    if bad:
        sys.exit("refusing to build the table: the suite draws unseeded inputs (measured bounds would not reproduce):\n  " + "\n  ".join(bad))
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "observed_errors.json")
    with open(src) as fh:
        seen = json.load(fh)
    bounds, rows = {}, []
    for test, per in sorted(seen["tests"].items()):
        for name, rec in sorted(per.items()):
            ceiling = getattr(tolerances, name).ceiling
            obs = rec["observed"]
            b = min(ceiling, max(tolerances.FLOOR, round_up(2.0 * obs)))
            bounds.setdefault(test, {})[name] = b
            rows.append((test, name, obs, b, ceiling))
    out = {"source": "errors recorded by tests/conftest.py over `pytest tests -m gpu` on an MI355X (tools/make_tolerances.py)",
           "rule": "bound = min(ceiling, max(%g, 2 x largest observed error rounded up to two digits))" % tolerances.FLOOR,
           "bounds": bounds, "observed": {t: {n: r["observed"] for n, r in per.items()} for t, per in seen["tests"].items()}}
    with open(tolerances.TABLE_PATH, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", os.path.relpath(tolerances.TABLE_PATH, ROOT), "-", len(rows), "bounds over", len(bounds), "tests")
    print("\n| path family | quantity | largest observed error | bound in force (2 x) | ceiling |\n|---|---|---|---|---|")
    for fam, prefixes in FAMILIES:
        per = {}
        for test, name, obs, b, ceiling in rows:
            if test.startswith(prefixes):
                o, bb, c = per.get(name, (0.0, 0.0, ceiling))
                per[name] = (max(o, obs), max(bb, b), ceiling)
        for name, (o, bb, c) in sorted(per.items()):
            print("| %s | %s | %.2g | %.2g | %.0e |" % (fam, name, o, bb, c))
    other = sorted({t for t, *_ in rows if not any(t.startswith(p) for _, p in FAMILIES)})
    if other:
        print("\n(other tests with bounds of their own: %s)" % ", ".join(other))


if __name__ == "__main__":
    main()
