#!/bin/bash
# Round-3 measurement set (run ON the GPU box from the repo root, e.g. `gpurun --timeout 1100 -- 'bash scripts/measure_r03.sh'`).
# Everything lands under gpurun_out/r03/; copy what profiles/README.md names into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r03
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
python bench.py --symmetry u1u1 --no-cpu-baseline --no-extras > $O/bench_u1u1.json 2> $O/bench_u1u1.err
python bench.py --chi 1024 --no-cpu-baseline --no-extras > $O/bench_chi1024.json 2> $O/bench_chi1024.err
echo "bench variants done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
echo "bench profile done"
python scripts/svd_bench.py theta4096 single > $O/svd_theta4096.log 2>&1
python scripts/svd_bench.py cfg2 full3 > $O/svd_lists.log 2>&1
python scripts/shard_model.py > $O/shard_model.log 2>&1
python scripts/cfg5_bench.py > $O/cfg5.log 2>&1
python scripts/lanczos_bench.py > $O/lanczos.log 2>&1
python scripts/csvd_bench.py > $O/csvd.log 2>&1
echo "svd / shard / cfg5 / lanczos done"
python scripts/dmrg_profile.py 32 256 12 2 --no-profile > $O/dmrg_chi256.log 2>&1
python scripts/dmrg_profile.py 32 512 13 2 --no-profile > $O/dmrg_chi512.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dmrg -o run -- python3 scripts/dmrg_profile.py 32 256 12 2 --no-profile > $O/prof_dmrg.log 2>&1
rm -f $O/prof_dmrg/run_kernel_trace.csv    # (40 MB; the stats are what is kept)
echo "dmrg done"
