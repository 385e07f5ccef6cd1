"""HBM traffic per launch of the 3x3 conv family from two rocprofv3 --pmc summaries (tools/pmc_summary.py
output for a FETCH_SIZE pass and a WRITE_SIZE pass of the same `bench.py --steps 1 --warmup 1` command).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and on gfx950 FETCH_SIZE counts a
wide coalesced read at half its size (MI355X_MICROARCH.md, HBM / rocprofv3 section).
usage: python tools/pmc_traffic.py FETCH.summary.csv WRITE.summary.csv OUT.json"""
import csv
import json
import sys


def read(path, counter):
    out = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["counter"] == counter:
                out[r["kernel"]] = (int(r["dispatches"]), float(r["total"]))
    return out


def main(fetch_csv, write_csv, out_json):
    f, w = read(fetch_csv, "FETCH_SIZE"), read(write_csv, "WRITE_SIZE")
    kernels = {}
    from pmc_families import FAMILIES
    fams = {name: [prefixes, 0.0, 0] for name, prefixes in FAMILIES.items()}
    for k in f:
        if k not in w:
            continue
        n = f[k][0]
        b = (2.0 * f[k][1] + w[k][1]) * 1024.0
        kernels[k] = {"dispatches": n, "hbm_bytes_per_launch": b / max(n, 1)}
        for fam in fams.values():
            if k.startswith(fam[0]):
                fam[1] += b
                fam[2] += n
    out = {"source": [fetch_csv, write_csv], "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, separate --pmc passes",
           "kernels": kernels}
    for name, (_, b, n) in fams.items():
        out[name] = {"dispatches": n, "hbm_bytes_per_launch": b / max(n, 1)}
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
