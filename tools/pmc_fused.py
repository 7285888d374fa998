#!/usr/bin/env python3
"""Per-frame-iteration fabric traffic of k_rl_fused from rocprofv3 PMC passes (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950; KB = 1024 B).  usage: pmc_fused.py <dir prefix> <frames> <k>"""
import csv, glob, sys
pre, frames, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out = {}
for suffix, counter, mult in (('_f', 'FETCH_SIZE', 2.0), ('_w', 'WRITE_SIZE', 1.0)):
    vals = []
    for f in glob.glob(pre + suffix + '/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and 'k_rl_fused' in r['Kernel_Name']:
                vals.append(float(r['Counter_Value']))
    out[counter] = max(vals) * 1024 * mult / (frames * k) / 1e6 if vals else None   # the K-iteration launch is the largest
print(pre, {c: (round(v, 2) if v else v) for c, v in out.items()}, 'MB per frame-iteration')
