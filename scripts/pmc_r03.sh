#!/bin/bash
# rocprofv3 PMC passes of round 3 (run ON the GPU box from the repo root, e.g. through gpurun):  bash scripts/pmc_r03.sh
# Separate passes per counter group (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), --kernel-trace only (never
# combined with sys/runtime traces), the probe program directly after "--".  Summarise with scripts/pmc_summarize_r03.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
groups=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64")
names=(fetch write sq)
for w in gemm_u1 gemm_u1u1 gemm_uniform svd; do
  case $w in
    gemm_u1) prog="scripts/gemm_lists.py u1 reps=6";;
    gemm_u1u1) prog="scripts/gemm_lists.py u1u1 reps=6";;
    gemm_uniform) prog="scripts/gemm_lists.py uniform reps=6";;
    svd) prog="scripts/svd_bench.py theta4096";;
  esac
  for i in 0 1 2; do
    out=gpurun_out/pmc_r03_${w}_${names[$i]}
    rm -rf $out
    rocprofv3 --kernel-trace --pmc ${groups[$i]} -d $out -o run --output-format csv -- python3 $prog > $out.log 2>&1
    echo "pass $w/${names[$i]} done"
  done
done
