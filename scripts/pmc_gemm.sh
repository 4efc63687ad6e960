#!/bin/bash
# rocprofv3 PMC passes of the grouped GEMM (run ON the GPU box from the repo root, e.g. through gpurun):
#   bash scripts/pmc_gemm.sh <tag>
# Separate passes per counter group (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), --kernel-trace only
# (never combined with sys/runtime traces), the probe program directly after "--".
# Outputs gpurun_out/pmc_<tag>_{theta,uniform}_<group>/ ; summarise with scripts/pmc_summarize.py.
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
groups=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64")
names=(fetch write sq)
for w in theta uniform; do
  if [ $w == theta ]; then prog="scripts/theta_gemm_probe.py"; else prog="scripts/gemm_probe.py 4096"; fi
  for i in 0 1 2; do
    out=gpurun_out/pmc_${tag}_${w}_${names[$i]}
    rm -rf $out
    rocprofv3 --kernel-trace --pmc ${groups[$i]} -d $out -o run --output-format csv -- python3 $prog > $out.log 2>&1
    echo "pass $w/${names[$i]} done"
  done
done
