#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs:  pmc_summary.py <counter_collection.csv> [...]  -> table on stdout."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r'\(.*', '', r['Kernel_Name'])
            name = re.sub(r'^void rl::', '', name)
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name in sorted(acc):
    c = acc[name]
    n = max(len(v) for v in c.values())
    print('%-70s launches %4d  ' % (name[:70], n) + '  '.join('%s=%.4g' % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
