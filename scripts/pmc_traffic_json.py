#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from the two rocprofv3 PMC passes of the bench command.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR_F -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR_W -- python3 bench.py ... (same)
    python scripts/pmc_traffic_json.py FETCH.csv WRITE.csv profiles/r03_pmc_traffic.json

Per steady-state launch of the rollout kernel (the first launch of a run carries no covariance pass
and is left out): mean of the counter over the launches, KB -> bytes, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950.  The file records the sha256 of the kernel sources the
passes ran on (bench.kernel_source_hash); bench.py refuses to quote it for another tree."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def per_launch(path, counter, kernel_substr):
    rows = [r for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows][1:]     # steady state: drop the first launch
    return sum(vals) / len(vals) * 1024.0, len(vals)


def main(fetch_csv, write_csv, out_json, n=3, directions=512, H=1000):
    kern = bench.rollout_kernel_name(n)
    sub = kern.split("<")[0]
    fetch, nf = per_launch(fetch_csv, "FETCH_SIZE", sub)
    write, nw = per_launch(write_csv, "WRITE_SIZE", sub)
    fetch *= 2.0                                             # gfx950 FETCH_SIZE correction (the guide)
    d = 2 * n + 2
    doc = {
        "_comment": "HBM traffic per steady-state launch from rocprofv3 PMC passes (one counter per pass, "
                    "--kernel-trace only), mean over the launches after the first; FETCH_SIZE doubled as "
                    "MI355X_MICROARCH.md prescribes for gfx950.  A steady-state rollout launch also carries the "
                    "covariance pass over the previous iteration's trajectories (extra workgroups of the same "
                    "grid): its reads are in this launch's counters.",
        "source_sha256": bench.kernel_source_hash(),
        "inputs": [os.path.basename(fetch_csv), os.path.basename(write_csv)],
        kern: {
            "workload": f"n={n}, {directions} directions x 2 rollouts, H={H}, trajectory capture + moments; "
                        "+ covariance pass over the previous iteration's trajectories",
            "launches_averaged": [nf, nw],
            "fetch_bytes": int(round(fetch)), "write_bytes": int(round(write)),
            "traffic_bytes": int(round(fetch + write)),
            "breakdown": {
                "rollout_workgroups: trajectory stores + returns/moments": int(round(write)),
                "covariance_workgroups: reads of the previous trajectory buffer + tile rows": int(round(fetch)),
            },
            "algorithmic_bytes_rollouts": bench.rollout_algorithmic_bytes(n, directions, H),
            "algorithmic_bytes_covariance_pass": 2 * directions * H * 8 * d,
        },
    }
    with open(out_json, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc[kern]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
