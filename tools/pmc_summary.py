#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counters per kernel name over the counter_collection csv files found under
the given directories.  Usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "?").split("(")[0]
                    cell = acc[name][row.get("Counter_Name", "?")]
                    cell[0] += float(row.get("Counter_Value", 0) or 0)
                    cell[1] += 1
    print("# rocprofv3 --pmc: per-kernel average counter value per dispatch")
    for name in sorted(acc):
        print(name)
        for counter in sorted(acc[name]):
            total, n = acc[name][counter]
            print(f"    {counter:28s} avg {total / max(n, 1):18.1f}   dispatches {n}")


if __name__ == "__main__":
    main(sys.argv[1:])
