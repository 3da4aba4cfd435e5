"""Condenses rocprofv3 --pmc counter CSVs (one run per counter set) into one table: per kernel name the
mean of every counter over its dispatches and the mean profiled duration.
Usage: python tools/pmc_kernel_table.py out.csv name_filter set1.csv [set2.csv ...]"""
import csv
import re
import sys


def short(name):
    m = re.search(r"aesmc::(\w+)<([^>]*)>", name)
    return "{}<{}>".format(m.group(1), m.group(2).replace(" ", "")) if m else name[:60]


def main(out_path, name_filter, *paths):
    table = {}
    for path in paths:
        for r in csv.DictReader(open(path)):
            kernel = r.get("Kernel_Name", "")
            if name_filter not in kernel:
                continue
            row = table.setdefault(short(kernel), {})
            row.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if r.get("End_Timestamp"):
                row.setdefault("duration_us_profiled", []).append(
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    counters = sorted({c for row in table.values() for c in row})
    with open(out_path, "w") as out:
        out.write("kernel," + ",".join(counters) + "\n")
        for kernel, row in table.items():
            out.write('"{}",'.format(kernel) + ",".join(
                "{:.5g}".format(sum(row[c]) / len(row[c])) if c in row else "nan" for c in counters) + "\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *sys.argv[3:])
