"""Manual helper (not a test): what this part sustains for plain streaming kernels whose data do not fit the Infinity Cache --
the yardstick for the "TB/s of fabric traffic" figures of DESIGN.md (torch elementwise kernels; bytes = reads + writes).

    python tools/gpu/gpu_stream_rate.py
"""
import time

import torch

dev = torch.device('cuda:0')
n = 1 << 29                                  # 2 GiB per float32 array
a = torch.rand(n, device=dev)
b = torch.rand(n, device=dev)
c = torch.empty_like(a)
cases = {'copy (1 read + 1 write)': (lambda: c.copy_(a), 2), 'add (2 reads + 1 write)': (lambda: torch.add(a, b, out=c), 3),
         'read only (sum)': (lambda: a.sum(), 1), 'write only (fill)': (lambda: c.fill_(1.0), 1)}
for name, (fn, arrays) in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print('%-28s %.2f TB/s' % (name, arrays * n * 4 / el / 1e12), flush=True)
