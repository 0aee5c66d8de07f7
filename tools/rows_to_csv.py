#!/usr/bin/env python3
"""Binary visibility sidecar (effex_amd.rowsink.BinSink, Correlator(output_format='bin')) -> the csv the reference's
writer produces for the same rows, byte for byte (effex/effex.py:667-696), so that post_process.py-style readers work.

    python tools/rows_to_csv.py visibilities_20260101-000000.fxb [out.csv]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) not in (2, 3):
        raise SystemExit(__doc__)
    from effex_amd import rowsink
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) == 3 else os.path.splitext(src)[0] + '.csv'
    n = rowsink.to_csv(src, dst)
    print("{}: {} rows -> {}".format(src, n, dst))


if __name__ == "__main__":
    main()
