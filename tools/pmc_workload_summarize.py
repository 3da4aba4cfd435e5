"""Reads the FETCH_SIZE and WRITE_SIZE counter CSVs of one tools/pmc_workload.py run pair and merges
the result into a traffic table (profiles/pmc_traffic.json):

    { "<workload>:<proposal>": { "<kernel key>": {"hbm_bytes_per_launch", "fetch_bytes", "write_bytes",
                                                  "algorithmic_bytes_per_launch", "launches", "source"} } }

Calibration: the first three resample_gather_kernel dispatches are identity-index gathers whose read
and write bytes are known exactly; their ratio to the raw counter gives the factor applied to every
other dispatch (gfx950: FETCH_SIZE reads 0.50 of a coalesced stream, WRITE_SIZE is exact).
Usage: python tools/pmc_workload_summarize.py <workload> <proposal> fetch.csv write.csv out.json"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

KERNELS = {"resample_step": "ancestor_index_",      # (K2: the lean form ancestor_index_rows_kernel or ancestor_index_inv_kernel)
            "resample_gather": "resample_gather_kernel",
           "normal_logweight": "normal_logweight", "normal_rsample": "normal_rsample",
           "affine_normal_rsample": "affine_rsample_kernel", "affine_normal_logweight": "affine_logweight_kernel",
           "affine_normal_propagate": "affine_logweight_kernel",      # K15 = K10's kernel in DRAW mode: it also writes x_t
           "affine_normal_propagate_drawn": "affine_propagate_item_kernel",      # K16 (round 5's form: one item per workgroup)
           "affine_normal_propagate_resampled": "affine_logweight_kernel",      # K15 fetching x_{t-1} through the ancestors
           "philox_normal_fill": "philox_normal_fill_kernel",
           "affine_step_backward_resampled": "affine_step_backward_rows_kernel"}      # K14 (pmc_workload.py with a backward)


def per_dispatch(path, counter, kernel):
    rows = [r for r in csv.DictReader(open(path))
            if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    return [float(r["Counter_Value"]) * 1024.0 for r in rows]


def main(workload, proposal, fetch_csv, write_csv, out_json):
    description, kind, dim, B, K, T, _ = bench.WORKLOADS[workload]
    payload = B * K * dim * 4
    mean = lambda xs: sum(xs) / len(xs)
    gather_fetch = per_dispatch(fetch_csv, "FETCH_SIZE", KERNELS["resample_gather"])
    gather_write = per_dispatch(write_csv, "WRITE_SIZE", KERNELS["resample_gather"])
    assert len(gather_fetch) >= 3 and len(gather_write) >= 3, (len(gather_fetch), len(gather_write))
    f_factor = (payload + B * K * 8) / mean(gather_fetch[:3])
    w_factor = payload / mean(gather_write[:3])
    source = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) over tools/pmc_workload.py {} {}: the first "
              "timesteps of bench.py's seeded workload; FETCH_SIZE x{:.3f}, WRITE_SIZE x{:.3f} as calibrated on "
              "identity-index gathers of the same shape").format(workload, proposal, f_factor, w_factor)
    entry = {"calibration": {"fetch_factor": f_factor, "write_factor": w_factor}}
    algorithmic = {"resample_step": B * K * (20 + 8 * dim) + 8 * B, "resample_gather": B * K * (8 + 8 * dim),
                   "affine_normal_rsample": B * K * 12 * dim, "affine_normal_logweight": B * K * (8 * dim + 4),
                   "affine_normal_propagate": B * K * (12 * dim + 4),
                   "affine_normal_propagate_drawn": B * K * (8 * dim + 12),      # indices, surviving rows in; x_t, lw out
                   "affine_normal_propagate_resampled": B * K * (12 * dim + 12), "philox_normal_fill": B * K * 4 * dim,
                   # K14 with the children folded in: ancestors, x_{t-1} rows, x_t, lw, the next step's per-child gradient and
                   # ranges in; the gradient of the resampled rows out
                   "affine_step_backward_resampled": B * K * (16 * dim + 16)}
    for key, kernel in KERNELS.items():
        skip = 3 if key == "resample_gather" else 0
        fetch = per_dispatch(fetch_csv, "FETCH_SIZE", kernel)[skip:]
        write = per_dispatch(write_csv, "WRITE_SIZE", kernel)[skip:]
        if key == "affine_step_backward_resampled":      # (backward order: the last timestep's launch has no children to fold)
            fetch, write = fetch[1:], write[1:]
        n = min(len(fetch), len(write))
        if n == 0:
            continue
        if key == "affine_normal_logweight":      # K10 proper writes the log-weights only
            pairs = [(f, w) for f, w in zip(fetch, write) if w * w_factor <= 0.5 * payload]
            if not pairs:
                continue
            fetch, write = [p[0] for p in pairs], [p[1] for p in pairs]
            n = len(pairs)
        if key in ("resample_step", "affine_normal_propagate", "affine_normal_propagate_resampled"):
            # launches with a payload only (time 0 has none; K2 alone writes 12 B/particle); K15: the draw
            pairs = [(f, w) for f, w in zip(fetch, write) if w * w_factor > 0.5 * payload]
            if not pairs and key == "resample_step":
                # round 3: the newest latent stays un-gathered (the propagation launch fetches the rows): K2 alone,
                # 4 B in, 8 B of indices out per particle (+ 4 B of children ranges when a backward will follow)
                pairs = list(zip(fetch, write))
                ranges = mean(write) * w_factor > 10 * B * K
                algorithmic = dict(algorithmic, resample_step=B * K * (12 + (4 if ranges else 0)) + 8 * B)
            if not pairs:
                continue
            fetch, write = [p[0] for p in pairs], [p[1] for p in pairs]
            n = len(pairs)
        f, w = mean(fetch[:n]) * f_factor, mean(write[:n]) * w_factor
        entry[key] = {"hbm_bytes_per_launch": f + w, "fetch_bytes": f, "write_bytes": w, "launches": n,
                      "source": source}
        if key in algorithmic:
            entry[key]["algorithmic_bytes_per_launch"] = algorithmic[key]
    table = {}
    if os.path.exists(out_json):
        table = json.load(open(out_json))
    table["{}:{}".format(workload, proposal)] = entry
    json.dump(table, open(out_json, "w"), indent=1, sort_keys=True)
    print(json.dumps({"{}:{}".format(workload, proposal): entry}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
