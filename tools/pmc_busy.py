"""MFMA-busy fraction per kernel family from a rocprofv3 --pmc summary (tools/pmc_summary.py output of a pass with
SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE): busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCD x 1024 SIMD)
(the busy counter sums over all SIMDs of the chip; GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles).
usage: python tools/pmc_busy.py SUMMARY.csv OUT.json"""
import csv
import json
import sys


def main(summary_csv, out_json):
    per = {}
    with open(summary_csv) as fh:
        for r in csv.DictReader(fh):
            per.setdefault(r["kernel"], {})[r["counter"]] = (int(r["dispatches"]), float(r["total"]))
    from pmc_families import FAMILIES as fams
    out = {"source": summary_csv, "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)", "kernels": {}}
    acc = {k: [0.0, 0.0, 0] for k in fams}
    for k, c in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c or c["GRBM_GUI_ACTIVE"][1] <= 0:
            continue
        busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"][1], c["GRBM_GUI_ACTIVE"][1]
        out["kernels"][k] = {"dispatches": c["GRBM_GUI_ACTIVE"][0], "mfma_busy": round(busy / (act / 8.0 * 1024.0), 4)}
        for name, prefixes in fams.items():
            if k.startswith(prefixes):
                acc[name][0] += busy
                acc[name][1] += act
                acc[name][2] += c["GRBM_GUI_ACTIVE"][0]
    for name, (busy, act, n) in acc.items():
        if act > 0:
            out[name] = {"dispatches": n, "mfma_busy": round(busy / (act / 8.0 * 1024.0), 4)}
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:3])
