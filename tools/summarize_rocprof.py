"""Condenses a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats file into the
short table committed under profiles/ (top kernels by total time + every aesmc:: kernel)."""
import csv
import sys


def main(path, top=12):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    print("# {}\n# total kernel time {:.3f} ms over {} launches".format(path.split("/")[-1], total / 1e6, calls))
    print("kernel,calls,avg_us,min_us,max_us,total_ms,percent")
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    shown = 0
    for r in rows:
        ours = "aesmc::" in r["Name"]
        if shown >= top and not ours:
            continue
        shown += 1
        name = r["Name"].split("(")[0][:100].replace(",", ";")
        print("{},{},{:.2f},{:.2f},{:.2f},{:.3f},{:.2f}".format(
            name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
