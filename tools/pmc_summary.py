#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (and grid size):
mean counter values per dispatch and mean duration.  Usage: pmc_summary.py <csv> [<csv> ...]"""
import csv
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for path in paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                key = (short(row["Kernel_Name"]), int(row["Grid_Size"]))
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[key][row["Dispatch_Id"] + path] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    for key in sorted(acc, key=lambda k: -sum(dur[k].values())):
        d = list(dur[key].values())
        print(f"{key[0]}  grid={key[1]}  dispatches={len(d)}  avg_us={sum(d) / len(d):.1f}")
        for c, v in sorted(acc[key].items()):
            print(f"    {c:32s} {sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1:])
