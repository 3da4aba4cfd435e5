"""Condenses the counter CSVs of tools/pmc_step.py (one rocprofv3 --pmc run per counter set) into
one table: per configuration (5 launches each) the mean of every counter, plus derived ratios.
Usage: python tools/pmc_step_summarize.py out.csv set1.csv [set2.csv ...]"""
import csv
import sys

CONFIGS = ["c2 B=256 K=1024 parts=1", "c2 B=256 K=1024 parts=2", "c4s B=128 K=4096 parts=1",
           "c4s B=128 K=4096 parts=2", "c4 B=1024 K=4096 parts=1"]


def main(out_path, *paths):
    table = {}
    for path in paths:
        rows = [r for r in csv.DictReader(open(path)) if "ancestor_index_inv_kernel" in r.get("Kernel_Name", "")]
        by_counter = {}
        for r in rows:
            by_counter.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                                                  int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0))))
        for counter, values in by_counter.items():
            values.sort()
            for i, config in enumerate(CONFIGS):
                chunk = values[5 * i:5 * i + 5]
                if chunk:
                    table.setdefault(config, {})[counter] = sum(v for _, v, _ in chunk) / len(chunk)
                    table[config]["duration_us_profiled"] = sum(d for _, _, d in chunk) / len(chunk) / 1e3
    counters = sorted({c for row in table.values() for c in row})
    with open(out_path, "w") as out:
        out.write("configuration," + ",".join(counters) + "\n")
        for config in CONFIGS:
            if config in table:
                out.write(config + "," + ",".join("{:.4g}".format(table[config].get(c, float("nan"))) for c in counters) + "\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
