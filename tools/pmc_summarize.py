"""Reads the two rocprofv3 counter CSVs (FETCH_SIZE run, WRITE_SIZE run) of tools/pmc_gather.py and
writes profiles/pmc_traffic.json + a readable table.  gfx950 calibration per MI355X_MICROARCH.md
(HBM section): counters are in KiB-like units of 1024 B; FETCH_SIZE can under-report coalesced reads
by 2x, so the identity-index launches (bytes known exactly) give the correction factor for THIS
access pattern, applied to the other launches."""
import csv
import json
import sys


def per_launch(path, counter, kernel="resample_gather_kernel"):
    rows = [r for r in csv.DictReader(open(path))
            if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
    return [float(r["Counter_Value"]) * 1024.0 for r in rows]


def main(fetch_csv, write_csv, out_json):
    fetch, write = per_launch(fetch_csv, "FETCH_SIZE"), per_launch(write_csv, "WRITE_SIZE")
    assert len(fetch) == 18 and len(write) == 18, (len(fetch), len(write))
    # fused step: per shape 2 launches of K2 alone, then 3 + 3 of the step with payload
    step_fetch = per_launch(fetch_csv, "FETCH_SIZE", "ancestor_index_inv_kernel")
    step_write = per_launch(write_csv, "WRITE_SIZE", "ancestor_index_inv_kernel")
    assert len(step_fetch) == 16 and len(step_write) == 16, (len(step_fetch), len(step_write))
    result = {}
    for s, (name, B, K, d) in enumerate([("c2", 256, 1024, 10), ("c4", 1024, 4096, 10)]):
        base = s * 9
        payload = B * K * d * 4
        known_read, known_write = payload + B * K * 8, payload
        mean = lambda xs: sum(xs) / len(xs)
        f_cal, w_cal = mean(fetch[base:base + 3]), mean(write[base:base + 3])
        f_corr, w_corr = known_read / f_cal, known_write / w_cal
        entry = {"algorithmic_bytes_per_launch": B * K * (8 * d + 8),
                 "calibration": {"fetch_raw": f_cal, "fetch_known": known_read, "fetch_factor": f_corr,
                                 "write_raw": w_cal, "write_known": known_write, "write_factor": w_corr}}
        for label, off in (("workload_s1", 3), ("degenerate_s5", 6)):
            f = mean(fetch[base + off:base + off + 3]) * f_corr
            w = mean(write[base + off:base + off + 3]) * w_corr
            entry[label] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w}
        entry["resample_gather_bytes_per_launch"] = entry["workload_s1"]["hbm_bytes"]
        # the fused step on the same operands (same correction factors: same access widths)
        entry["step_algorithmic_bytes_per_launch"] = B * K * (12 + 8 * d + 8) + 8 * B
        sbase = s * 8 + 2
        for label, off in (("step_workload_s1", 0), ("step_degenerate_s5", 3)):
            f = mean(step_fetch[sbase + off:sbase + off + 3]) * f_corr
            w = mean(step_write[sbase + off:sbase + off + 3]) * w_corr
            entry[label] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w}
        entry["resample_step_bytes_per_launch"] = entry["step_workload_s1"]["hbm_bytes"]
        # K5 / K6: 3 launches each per shape, streaming kernels (same correction factors)
        for key, kernel, algorithmic in (
                ("normal_logweight", "normal_logweight_kernel", B * K * (4 * d + 1) * 4 + B * d * 4),
                ("normal_rsample", "normal_rsample_dense_kernel", B * K * d * 4 * 3)):
            f_all = per_launch(fetch_csv, "FETCH_SIZE", kernel)
            w_all = per_launch(write_csv, "WRITE_SIZE", kernel)
            if len(f_all) >= 3 * (s + 1) and len(w_all) >= 3 * (s + 1):
                f = mean(f_all[3 * s:3 * s + 3]) * f_corr
                w = mean(w_all[3 * s:3 * s + 3]) * w_corr
                entry[key] = {"algorithmic_bytes": algorithmic, "fetch_bytes": f, "write_bytes": w,
                              "hbm_bytes": f + w}
        result[name] = entry
    json.dump(result, open(out_json, "w"), indent=1)
    print(json.dumps(result, indent=1))


def condense(raw_csv, out_csv):
    """The aesmc:: rows of a rocprofv3 counter_collection CSV, one line per dispatch."""
    with open(out_csv, "w") as out:
        out.write("dispatch_id,kernel,grid_size,counter,value_x1024B,duration_ns\n")
        for r in csv.DictReader(open(raw_csv)):
            name = r.get("Kernel_Name", "")
            if "aesmc::" not in name:
                continue
            duration = ""
            if r.get("End_Timestamp") and r.get("Start_Timestamp"):
                duration = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            out.write("{},{},{},{},{},{}\n".format(r.get("Dispatch_Id", ""), name.replace(",", ";").split("(")[0],
                                                  r.get("Grid_Size", ""), r.get("Counter_Name", ""),
                                                  r.get("Counter_Value", ""), duration))


if __name__ == "__main__":
    main(*sys.argv[1:4])
    if len(sys.argv) >= 6:
        condense(sys.argv[1], sys.argv[4])
        condense(sys.argv[2], sys.argv[5])
