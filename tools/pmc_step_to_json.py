"""Merge the step-level PMC passes (tools/pmc_step.sh -> profiles/<tag>_pmc_step_counters.csv) into profiles/pmc_traffic.json:
per-kernel HBM-side bytes of the kernels bench.py prices inside the step, and step_total_bytes = the sum over the step's kernels of
(2 FETCH_SIZE + WRITE_SIZE) KB x launches per step.   usage: python tools/pmc_step_to_json.py profiles/r6_pmc_step_counters.csv [steps=11]"""
import collections
import csv
import json
import sys

KEYS = ("nce_fwd_h2_kernel", "nce_bwd_g_kernel", "gru2_persist_bwd_kernel", "gru2_persist_fwd_h2_kernel", "conv_dgrad_dma_kernel",
        "conv0_bwd_kernel")


def main():
    src = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    rows = [r for r in csv.reader(open(src)) if len(r) == 4]
    k = collections.defaultdict(dict)
    for name, c, n, v in rows:
        k[name][c] = float(v)
        k[name]["n"] = int(n)
    d = json.load(open("profiles/pmc_traffic.json"))
    tot = 0.0
    for name, c in k.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        tb = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        tot += tb * c["n"] / steps
        s = next((key for key in KEYS if key in name), None)
        if s:
            e = {"FETCH_SIZE": c["FETCH_SIZE"], "WRITE_SIZE": c["WRITE_SIZE"], "launches": c["n"], "traffic_bytes": tb,
                 "from": src + " (inside the train step)"}
            if c.get("GRBM_GUI_ACTIVE"):
                e["mfma_busy_frac"] = (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0)
            d["kernels"][s] = e
    d["step_total_bytes"] = tot
    d["step_total_source"] = src + ": sum over the step's kernels of (2 FETCH_SIZE + WRITE_SIZE) KB x launches per step (tools/pmc_step.sh)"
    json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(f"{tot / 1e9:.3f} GB per step")


if __name__ == "__main__":
    main()
