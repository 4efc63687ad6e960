#!/bin/bash
# rocprofv3 PMC passes of the batched SVD of the chi=4096 theta list (run ON the GPU box from the repo root):
#   bash scripts/pmc_svd.sh <tag>
# Separate passes per counter group, --kernel-trace only, the probe program directly after "--" (as scripts/pmc_gemm.sh).
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
groups=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64")
names=(fetch write sq)
for i in 0 1 2; do
  out=gpurun_out/pmc_${tag}_svd_${names[$i]}
  rm -rf $out
  rocprofv3 --kernel-trace --pmc ${groups[$i]} -d $out -o run --output-format csv -- python3 scripts/svd_bench.py theta4096 > $out.log 2>&1
  echo "pass svd/${names[$i]} done"
done
