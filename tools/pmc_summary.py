"""Summarise rocprofv3 --pmc counter CSVs into profiles/: per kernel, mean counter value per launch.

usage: pmc_summary.py <out.json> <counter_collection.csv> [more csv ...]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  The calibration launches (k_diag_read<T>, 1 GiB each) give
the factor known_bytes / counted_bytes for their lane width (MI355X_MICROARCH.md, HBM section: the counter sees half
of a 16 B/lane streaming read on gfx950); the factor of the 4 B/lane read -- the access width of the similar-k-mer
table rows that dominate k_match -- is applied to the search kernels' FETCH_SIZE."""
import csv
import json
import sys
from collections import defaultdict

GIB = 1 << 30


def short(name):
    for k in ("k_match", "k_prep", "k_score", "k_select", "k_scan_counts", "k_reorder_hits", "k_sw_byte1p", "k_sw_byte1", "k_sw_byte", "k_sw_word", "k_aln_assemble", "k_aln_lens", "k_tantan", "k_candidates", "k_select_merged",
              "k_sw_prep", "k_sw_gate", "k_pairs_from_hits", "k_sw_trace", "k_rescore"):
        if k in name:
            return k
    if "k_diag_read" in name:
        return "k_diag_read<16B>" if ("uint4" in name or "HIP_vector_type<unsigned int, 4" in name) else (
            "k_diag_read<8B>" if ("uint2" in name or "HIP_vector_type<unsigned int, 2" in name) else "k_diag_read<4B>")
    return name[:48]


def binary_identity():
    """what the counters were collected on: sha256 of the library the workload loaded and of the device sources it was built from (bench.py compares the latter with the
    tree it runs from and marks `roofline.traffic` stale when they differ -- a host-only change rebuilds the library but not its kernels)"""
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "spacepharer_amd", os.environ.get("SP_LIB_NAME", "libspacepharer_amd.so"))
    if not os.path.exists(lib):
        lib = os.path.join(root, "spacepharer_amd", "csrc", "build", os.path.basename(lib))
    out = {"library": os.path.basename(lib), "library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None}
    h = hashlib.sha256()
    src = os.path.join(root, "spacepharer_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f.endswith(".hip") or f.endswith(".inc"):
            h.update(f.encode()); h.update(open(os.path.join(src, f), "rb").read())
    out["device_sources_sha256"] = h.hexdigest()
    return out


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for p in paths:
        seen = set()
        for row in csv.DictReader(open(p)):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    res = {}
    for k, ctrs in acc.items():
        if not k.startswith("k_"):
            continue
        res[k] = {"launches": max(len(v) for v in ctrs.values()), "mean_us_under_pmc": sum(dur[k]) / max(1, len(dur[k]))}
        for c, v in ctrs.items():
            v = v[1:] if len(v) > 1 else v          # drop the first (cold) launch
            res[k][c] = sum(v) / len(v)
    calib = {}
    for width, name in ((4, "k_diag_read<4B>"), (16, "k_diag_read<16B>")):
        if name in res and "FETCH_SIZE" in res[name]:
            calib["%dB_per_lane" % width] = GIB / (res[name]["FETCH_SIZE"] * 1024.0)
    stream = {}
    for width, name in ((4, "k_diag_read<4B>"), (16, "k_diag_read<16B>")):
        if name in res and res[name].get("mean_us_under_pmc"):
            stream["%dB_per_lane_GBps" % width] = GIB / (res[name]["mean_us_under_pmc"] * 1e-6) / 1e9
    f4 = calib.get("4B_per_lane", 1.0)
    summary = {"source": paths, "binary": binary_identity(), "unit_note": "FETCH_SIZE/WRITE_SIZE in KiB as printed by rocprofv3; *_bytes fields are bytes per launch",
               "calibration_factor_known_over_counted": calib, "streaming_read_of_1GiB": stream, "kernels": res}
    for k in res:
        if "FETCH_SIZE" in res[k] and not k.startswith("k_diag"):
            res[k]["fetch_bytes_raw"] = res[k]["FETCH_SIZE"] * 1024.0
            res[k]["fetch_bytes_corrected"] = res[k]["FETCH_SIZE"] * 1024.0 * f4
        if "WRITE_SIZE" in res[k]:
            res[k]["write_bytes_raw"] = res[k]["WRITE_SIZE"] * 1024.0
    json.dump(summary, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(summary["calibration_factor_known_over_counted"]))
    for k in sorted(res):
        print(k, {c: round(v, 1) for c, v in res[k].items()})


if __name__ == "__main__":
    main()
