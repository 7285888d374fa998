# manual helper: PMC traffic of the fused kernel for a list of configurations (W:WGS:ACQ:S[:flags])
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-64:2:0:1 32:2:0:1 64:2:1:1}; do
  tag=$(echo $cfg | tr ':' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_f -- python tools/gpu/gpu_fused.py --batch 256 --reps 1 --no-ref --configs $cfg > gpurun_out/pmc_${tag}_f.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_w -- python tools/gpu/gpu_fused.py --batch 256 --reps 1 --no-ref --configs $cfg > gpurun_out/pmc_${tag}_w.log 2>&1
done
echo done
