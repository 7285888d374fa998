#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 --pmc csv output.  usage: pmc_table.py DIR [name filter]"""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void rl::', '')
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in agg.items():
    print(name)
    for c, v in sorted(cs.items()):
        print('   %-32s n=%4d  avg %.4g' % (c, len(v), sum(v) / len(v)))
