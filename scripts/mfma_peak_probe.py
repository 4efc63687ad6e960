"""The f64-MFMA issue-rate microbenchmark (`cyb_mfma_f64_peak`) in its variants: accumulators per wave x waves per SIMD.
    python3 scripts/mfma_peak_probe.py                     # plain: TFLOP/s per variant
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 \
        -d gpurun_out/pmc_peak -o run --output-format csv -- python3 scripts/mfma_peak_probe.py
    python3 scripts/mfma_peak_probe.py --summarize gpurun_out/pmc_peak      # clock and MFMA-busy of every launch
The question it answers (hygiene 11): the loop reads 47 TFLOP/s where the spec says 78.6 -- is the pipe idle
(issue-limited loop) or is the clock down (power management)?"""
import csv
import glob
import sys

N_CU = 256

if '--summarize' in sys.argv:
    d = sys.argv[sys.argv.index('--summarize') + 1]
    f = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)[0]
    per = {}
    for row in csv.DictReader(open(f)):
        if 'mfma_f64_peak' not in row['Kernel_Name']:
            continue
        k = int(row['Dispatch_Id'])
        e = per.setdefault(k, {'us': (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-3, 'grid': int(row['Grid_Size']),
                               'name': row['Kernel_Name']})
        e[row['Counter_Name']] = e.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
    for k in sorted(per):
        e = per[k]
        clk = e['GRBM_GUI_ACTIVE'] / 8 / (e['us'] * 1e-6) / 1e9
        busy = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * N_CU * e['GRBM_GUI_ACTIVE'] / 8)
        n = e['SQ_INSTS_VALU_MFMA_F64']
        nacc = e['name'].split('<')[1].split('>')[0] if '<' in e['name'] else '?'
        print(f"[peak-pmc] acc={nacc} waves/SIMD={e['grid'] // 256 // N_CU}: {e['us']:9.1f} us  {n * 2048 / (e['us'] * 1e-6) / 1e12:6.2f} TFLOP/s  "
              f"clock {clk:.3f} GHz  MFMA busy {busy:.3f}  busy cycles per MFMA {e['SQ_VALU_MFMA_BUSY_CYCLES'] / max(n, 1):.1f}  "
              f"-> at this clock the pipe's ceiling is {N_CU * 4 * 32 * clk / 1e3:.1f} TFLOP/s")
    sys.exit(0)

sys.path.insert(0, '.')
from cyten_amd.runtime import Context

ctx = Context(0)
for nacc in (1, 2, 4, 8):
    for wps in (1, 2, 4):
        tf, ms = ctx.mfma_f64_peak(400000, wps, nacc)
        print(f'[peak] {nacc} accumulators, {wps} wave(s)/SIMD: {tf:6.2f} TFLOP/s ({ms:.2f} ms)', flush=True)
