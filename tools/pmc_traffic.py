#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes over bench.py into profiles/rNN/pmc_traffic.json (fabric bytes per launch).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- \
        python bench.py --steps 1 --warmup 0 --no-cpu-baseline --kernel-reps 2
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- (same)
    python tools/pmc_traffic.py <fetch counter csv> <write counter csv> <out json> [frames per launch] [size] [views] [dtype]

Only the most frequent launch shape of each kernel is kept: for the four RL kernels that is
the slice of the batch the RL loop launches them on (bench.py: roofline.frames_per_launch, pass it
as the 4th argument), the shape bench.py times for `roofline`.  FETCH_SIZE / WRITE_SIZE are reported in KB (1024 B); FETCH_SIZE is
doubled (MI355X_MICROARCH.md, HBM/rocprofv3 section: gfx950 reports half the bytes of coalesced
streaming reads; re-calibrated here on rowpass_FWD, which must read batch x ny x nx x 4 bytes).
"""
import csv
import json
import re
import sys
from collections import defaultdict

NAMES = (  # (regex on the kernel name, key in the json); first match wins
    (r'k_rowpair<\d+, \d+, 2,', 'rowpass_RATIO'), (r'k_rowpair<\d+, \d+, 3,', 'rowpass_UPDATE'), (r'k_rowpair<\d+, \d+, 0,', 'rowpair_FWD'),
    (r'k_rowpass<\d+, \d+, 0,', 'rowpass_FWD'), (r'k_rowpass<\d+, \d+, 1,', 'rowpass_INV'),
    (r'k_rowpass<\d+, \d+, 2,', 'rowpass_RATIO'), (r'k_rowpass<\d+, \d+, 3,', 'rowpass_UPDATE'),
    (r'k_rowpass<\d+, \d+, 4,', 'rowpass_ADJ'),
    # the split column pass of multi-view plans on the long transforms (conv_kernels.hpp COL_SPLIT_FWD / _INV / _INV_SUM)
    (r'k_colconv_outer<\d+, \d+, \w+, 3[,>]', 'colsplit_FWD'), (r'k_colconv_outer<\d+, \d+, \w+, 4[,>]', 'colsplit_INV'),
    (r'k_colconv_outer<\d+, \d+, \w+, 5[,>]', 'colsplit_INV_SUM'),
    (r'k_colconv<\d+, \d+, 2,', 'colconv_Ht'), (r'k_colconv<\d+, \d+, 1,', 'colconv_H'),
    (r'k_colconv', 'colconv'), (r'k_poisson_fast', 'poisson_fast'), (r'k_poisson_slow', 'poisson_slow'), (r'k_poisson', 'poisson'))


def per_kernel(path, counter):
    rows = defaultdict(list)   # key -> [(grid, value)]
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            for rx, key in NAMES:
                if re.search(rx, r['Kernel_Name']):
                    rows[key].append((int(r['Grid_Size']), float(r['Counter_Value'])))
                    break
    out = {}
    for key, v in rows.items():
        grids = [g for g, _ in v]
        common = max(set(grids), key=grids.count)     # the launch shape that dominates the run
        sel = [x for g, x in v if g == common]
        out[key] = sum(sel) / len(sel)
        if key == 'colsplit_FWD':
            # two shapes per RL iteration: the slice's frames (H) and its frames x views images (H_t)
            top = sorted(sorted(set(grids), key=grids.count)[-2:])
            for g, name in zip(top, ('colsplit_FWD_frames', 'colsplit_FWD_images')):
                sel = [x for gg, x in v if gg == g]
                out[name] = sum(sel) / len(sel)
    return out


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    fl = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    size = int(sys.argv[5]) if len(sys.argv) > 5 else 512
    views = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    dtype = sys.argv[7] if len(sys.argv) > 7 else 'f32'
    res = {'_how': __doc__.strip().split('\n\n')[0] + ' (tools/pmc_traffic.py; FETCH_SIZE doubled, KB = 1024 B)',
           'frames_per_launch': fl, 'dtype': dtype, 'shape': [size, size], 'n_psf': views}
    for key in sorted(set(fetch) | set(write)):
        f, w = fetch.get(key, 0.0), write.get(key, 0.0)
        # FETCH_SIZE / WRITE_SIZE are the L2's fabric-side request counters: Infinity Cache hits are counted too, so
        # this is fabric traffic (what leaves the XCD's L2), not bytes that reached HBM
        res[key] = {'fabric_bytes_per_launch': 2 * f * 1024 + w * 1024, 'fetch_kb_reported': f, 'write_kb_reported': w}
    if all(k in res for k in ('colsplit_FWD_frames', 'colsplit_FWD_images', 'colsplit_INV', 'colsplit_INV_SUM')):
        # a column pass of these plans is two launches: forward half + inverse half
        for k, parts in (('colconv_H', ('colsplit_FWD_frames', 'colsplit_INV')), ('colconv_Ht', ('colsplit_FWD_images', 'colsplit_INV_SUM'))):
            res[k] = {'fabric_bytes_per_launch': sum(res[q]['fabric_bytes_per_launch'] for q in parts),
                      'fetch_kb_reported': sum(res[q]['fetch_kb_reported'] for q in parts),
                      'write_kb_reported': sum(res[q]['write_kb_reported'] for q in parts), 'launches': list(parts)}
    if 'colconv' in res:   # the H and H_t column passes are the same kernel (single view / per-image launches)
        for k in ('colconv_H', 'colconv_Ht'):
            res.setdefault(k, res['colconv'])
    rl = ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')
    if all(k in res for k in rl):
        total = sum(res[k]['fabric_bytes_per_launch'] for k in rl)
        alg = (4 if dtype == 'f32' else 8) * size * size * (3 * views + 4) * fl
        res['rl_iteration'] = {'fabric_bytes': total, 'algorithmic_bytes': alg, 'ratio': total / alg,
                               'fabric_MB_per_frame_iteration': total / fl / 1e6}
    json.dump(res, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps({k: v['fabric_bytes_per_launch'] for k, v in res.items() if isinstance(v, dict) and 'fabric_bytes_per_launch' in v}))
    print(json.dumps(res.get('rl_iteration')))


if __name__ == '__main__':
    main()
