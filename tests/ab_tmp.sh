for v in c32q16 c16q16 c32q8 c16q8 c16q4 c32q4; do cp rescan_line_sted_amd/_lib/librlsted_$v.so rescan_line_sted_amd/_lib/librlsted.so; echo "$v: "; timeout -k 10 200 python tools/gpu/gpu_bench_small.py 2>&1 | grep "128" | cut -c1-70; done
cp rescan_line_sted_amd/_lib/librlsted_c32q16.so rescan_line_sted_amd/_lib/librlsted.so
