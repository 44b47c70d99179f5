#!/usr/bin/env python3
"""Step periods of a traced bench.py run: the distance between the ends of consecutive `adam_update_kernel` launches (one per training step), in call order --
shows whether the first timed region's steps differ from the later ones.  usage: step_periods.py <dir with the rocprofv3 output> [marker kernel substring]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_summary import rows_from      # noqa: E402


def main():
    rows = sorted(rows_from(sys.argv[1]), key=lambda r: r[1])
    mark = sys.argv[2] if len(sys.argv) > 2 else "adam_update_kernel"
    ends = [b for n, a, b in rows if mark in n]
    per = [(ends[i + 1] - ends[i]) / 1e6 for i in range(len(ends) - 1)]
    print(len(ends), "marker launches; periods in ms:")
    print(" ".join("%.2f" % p for p in per))


if __name__ == "__main__":
    main()
