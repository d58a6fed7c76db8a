"""Summarise rocprofv3 --pmc passes for one kernel into a small JSON (kept under profiles/).

    python tools/pmc_summary.py <kernel substring> <out.json> <pass dir> [<pass dir> ...]

Every pass directory holds the *_counter_collection.csv of one `rocprofv3 --pmc ...` run.  Per
counter: number of launches of the kernel and the mean value per launch.  HBM traffic per launch
follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) reads at 64 bytes, so reads are
doubled; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys


def main():
    kernel, out_path, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    counters = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = {}
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    if kernel not in row["Kernel_Name"]:
                        continue
                    key = (row["Counter_Name"], row["Dispatch_Id"])
                    per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
            for (name, _), v in per_dispatch.items():
                counters.setdefault(name, []).append(v)
    summary = {name: {"launches": len(v), "mean_per_launch": sum(v) / len(v), "median_per_launch": sorted(v)[len(v) // 2],
                      "min_per_launch": min(v), "max_per_launch": max(v)} for name, v in sorted(counters.items())}
    out = {"kernel": kernel, "counters": summary}
    if os.environ.get("PMC_SOURCE_SHA16"):   # tools/kernel_only.py: hash of the kernel's sources the counters belong to
        out["source_sha16"] = os.environ["PMC_SOURCE_SHA16"]
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        rd = 2.0 * summary["FETCH_SIZE"]["mean_per_launch"] * 1024.0
        wr = summary["WRITE_SIZE"]["mean_per_launch"] * 1024.0
        out["hbm_bytes_per_launch_median"] = {"read": 2.0 * summary["FETCH_SIZE"]["median_per_launch"] * 1024.0,
                                              "write": summary["WRITE_SIZE"]["median_per_launch"] * 1024.0}
        out["hbm_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr,
                                       "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide reads), WRITE_SIZE KiB x 1024"}
    if "TCC_HIT_sum" in summary and "TCC_MISS_sum" in summary:
        h, m = summary["TCC_HIT_sum"]["mean_per_launch"], summary["TCC_MISS_sum"]["mean_per_launch"]
        out["l2_hit_rate"] = h / (h + m) if h + m else None
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
