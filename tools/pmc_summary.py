#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch).

usage: python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE ...
Each directory is one separate --pmc pass of the same bench command (the guide's rule:
FETCH_SIZE and WRITE_SIZE do not fit in one pass). Values are printed as reported by the
counter (KB for *_SIZE); corrections are applied by the reader, see DESIGN.md section 6.
"""
import csv, glob, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"mmvae::(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    table = defaultdict(dict)
    vg = {}
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            acc = defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k:
                    continue
                a = acc[(k, r["Counter_Name"])]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                vg[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
            for (k, c), (s, n) in acc.items():
                table[k][c] = (s / n, n)
    ctrs = sorted({c for v in table.values() for c in v})
    print("kernel," + ",".join(ctrs) + ",launches,vgpr,agpr,lds_bytes,scratch")
    for k in sorted(table):
        row = ['"%s"' % k] + ["%.1f" % table[k][c][0] if c in table[k] else "" for c in ctrs]
        n = max(v[1] for v in table[k].values())
        print(",".join(row + [str(n)] + list(vg[k])))


if __name__ == "__main__":
    main()
