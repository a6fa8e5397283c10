"""Per-kernel PMC means (tools/rocpd_pmc.py rows: "kernel",counter,launches,mean) -> profiles/pmc_traffic.json, the file
bench.py reads `roofline.traffic` from.  traffic_bytes = 2 * FETCH_SIZE + WRITE_SIZE in bytes: rocprofv3 reports both in KB,
and on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section).
usage: python tools/pmc_to_json.py pmc_counters.csv B out.json"""
import csv
import json
import sys


def short(name):
    if "conv_fwd_dma_kernel<256" in name:
        return "conv_fwd_dma_kernel<256>"
    if "conv_fwd_dma_kernel<128" in name:
        return "conv_fwd_dma_kernel<128>"
    if "conv0_fwd_kernel" in name:
        return "conv0_fwd_kernel"
    if "nce_fwd_kernel" in name:
        return "nce_fwd_kernel"
    return name


def main():
    rows = list(csv.reader(open(sys.argv[1])))
    kernels = {}
    for r in rows:
        if len(r) != 4:
            continue
        k = kernels.setdefault(short(r[0]), {})
        k[r[1]] = float(r[3])
        k["launches"] = int(r[2])
    for k, c in kernels.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["traffic_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            # MFMA busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            c["mfma_busy_frac"] = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0)
    out = {"batch": int(sys.argv[2]), "source": "tools/pmc_collect.sh (rocprofv3 --kernel-trace --pmc, one counter group per pass)",
           "note": "traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled per the gfx950 correction",
           "kernels": kernels}
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
