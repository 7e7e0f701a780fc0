"""Fold two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes, as the TCC slot budget requires) into
profiles/pmc_traffic.json: HBM bytes per launch for every kernel.

    python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload> <tag> [<kernel_stats.csv>] [--merge]

--merge: the passes were taken with another solver of the same sources (the direct chain's k_ldl_* kernels beside the default solver's): kernels the
workload's entry does not hold yet are added to it, the ones it holds stay.

With the fifth argument (the rocprofv3 --kernel-trace --stats summary of the same command) the per-kernel average durations are
recorded as well (`_rocprofv3_avg_us`); bench.py prints them beside its own HIP-event averages.

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): the counters are in KB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced read stream, so the read side is doubled; WRITE_SIZE is
exact for 16-byte stores and float atomics.  `raw_*` keeps the uncorrected values.
"""
import collections
import csv
import json
import os
import sys


def per_kernel(path):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("aar::", "").split("<")[0]
        acc[k][0] += float(row["Counter_Value"])
        acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    merge = "--merge" in sys.argv
    argv = [a for a in sys.argv if a != "--merge"]
    fetch, write, workload, tag = argv[1:5]
    stats = argv[5] if len(argv) > 5 else None
    f, w = per_kernel(fetch), per_kernel(write)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_path = os.path.join(root, "profiles", "pmc_traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    sys.path.insert(0, root)
    import bench
    entry = {"_source": tag, "_kernel_source_sha1": bench.kernel_source_hash(), "_note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts half of a coalesced read stream)"}
    rows = []
    for k in sorted(f):
        if not k.startswith("k_"):
            continue
        fk, n = f[k]
        wk = w.get(k, (0.0, 0))[0]
        entry[k] = (2.0 * fk + wk) * 1024.0
        entry["raw_" + k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "launches": n, "bytes_uncorrected": (fk + wk) * 1024.0}
        rows.append((k, n, fk, wk, entry[k]))
    if stats:
        entry["_rocprofv3_avg_us"] = {}
        for row in csv.DictReader(open(stats)):
            k = row["Name"].split("(")[0].replace("void ", "").replace("aar::", "").split("<")[0]
            if k.startswith("k_"):
                entry["_rocprofv3_avg_us"][k] = float(row["AverageNs"]) * 1e-3
    old = data.get("workload_%s" % workload)
    if merge and old and old.get("_kernel_source_sha1") == entry["_kernel_source_sha1"]:
        for k, v in entry.items():
            if k == "_rocprofv3_avg_us":
                for kk, vv in v.items():
                    old.setdefault("_rocprofv3_avg_us", {}).setdefault(kk, vv)
            elif not k.startswith("_"):
                old.setdefault(k, v)
        old["_source"] = "%s + %s" % (old["_source"], tag)
        entry = old
    data["workload_%s" % workload] = entry
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    with open(os.path.join(root, "profiles", "%s_pmc_summary.csv" % tag), "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch_corrected\n")
        for r in rows:
            fh.write("%s,%d,%.3f,%.3f,%.0f\n" % r)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
