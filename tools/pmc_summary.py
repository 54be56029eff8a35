#!/usr/bin/env python3
"""Summarises rocprofv3 outputs (kernel stats + PMC counter CSVs) into one JSON per round under profiles/.

    python tools/pmc_summary.py --out profiles/r1/summary_r1d.json --stats gpurun_out/prof_r1d/r1d_kernel_stats.csv \
        --pmc gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv ...

HBM-side traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB of
fabric requests behind the L2 (Infinity-Cache hits included); on gfx950 FETCH_SIZE counts 128-B requests as 64 B for
wide (16 B/lane) streams, so the read side is doubled. Counters were collected in separate --pmc passes, each with
--kernel-trace only."""
import argparse
import collections
import csv
import json
import re


def kname(k):
    m = re.search(r"conv3d_f16_mfma<([^>]*)>", k)
    if m:
        return "conv3d_f16_mfma<" + m.group(1).replace(" ", "") + ">"
    m = re.search(r"sn::([A-Za-z0-9_]+)", k) or re.search(r"_ZN2sn\d+([A-Za-z0-9_]+?)(?:I|E)", k)
    return m.group(1) if m else k[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*", default=[])
    a = ap.parse_args()
    out = {"kernels": {}}
    if a.stats:
        for r in csv.DictReader(open(a.stats)):
            k = kname(r["Name"])
            out["kernels"].setdefault(k, {})["rocprof_avg_us"] = round(float(r["AverageNs"]) / 1e3, 2)
            out["kernels"][k]["rocprof_calls"] = int(r["Calls"])
            out["kernels"][k]["rocprof_pct"] = float(r["Percentage"])
    for path in a.pmc:
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            k = kname(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
        for k in acc:
            d = out["kernels"].setdefault(k, {}).setdefault("pmc_per_launch", {})
            for c, v in acc[k].items():
                d[c] = v / len(cnt[k])
    for k, d in out["kernels"].items():
        p = d.get("pmc_per_launch", {})
        if "FETCH_SIZE" in p:
            d["hbm_read_bytes_per_launch"] = p["FETCH_SIZE"] * 1024 * 2      # gfx950 correction (x2), KiB -> B
        if "WRITE_SIZE" in p:
            d["hbm_write_bytes_per_launch"] = p["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in p:
            d["l2_hit_rate"] = round(p["TCC_HIT_sum"] / max(1.0, p["TCC_HIT_sum"] + p["TCC_MISS_sum"]), 4)
        if "SQ_LDS_BANK_CONFLICT" in p and p.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = round(p["SQ_LDS_BANK_CONFLICT"] / p["SQ_LDS_IDX_ACTIVE"], 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in p and "GRBM_GUI_ACTIVE" in out["kernels"][k].get("pmc_per_launch", {}):
            d["mfma_util"] = round(p["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * p["GRBM_GUI_ACTIVE"] / 8), 4)      # 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        if "SQ_LDS_IDX_ACTIVE" in p and "GRBM_GUI_ACTIVE" in p:
            d["lds_busy"] = round(p["SQ_LDS_IDX_ACTIVE"] / (256 * p["GRBM_GUI_ACTIVE"] / 8), 4)                  # 256 CUs, one LDS each
    json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
    print("wrote", a.out, "with", len(out["kernels"]), "kernels")


if __name__ == "__main__":
    main()
