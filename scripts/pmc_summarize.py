"""Summarise the PMC passes written by scripts/pmc_gemm.sh into profiles/<tag>_gemm_pmc_summary.json.

Per-launch means over the last 5 dispatches of gemm_grouped_kernel in every pass.  HBM bytes follow
MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.  Clock =
GRBM_GUI_ACTIVE / 8 / kernel time.  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x n_cu x
GRBM_GUI_ACTIVE/8); v_mfma_f64_16x16x4 occupies the pipe 64 cycles.
"""
import csv, glob, json, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
N_CU = 256
ALG = {'theta': dict(flops=10798826696.0, bytes=210354688.0),
       'uniform': dict(flops=2.0 * 4096 ** 3, bytes=3 * 8.0 * 4096 ** 2)}
out = {'note': __doc__.strip()}
for w in ('theta', 'uniform'):
    vals, dur = {}, None
    for grp in ('fetch', 'write', 'sq'):
        files = glob.glob(f'gpurun_out/pmc_{tag}_{w}_{grp}/**/*counter_collection.csv', recursive=True)
        if not files:
            continue
        per = defaultdict(lambda: defaultdict(float))
        times = {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if 'gemm_grouped_kernel' not in row['Kernel_Name']:
                    continue
                d = int(row['Dispatch_Id'])
                per[d][row['Counter_Name']] += float(row['Counter_Value'])
                times[d] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6
        last = sorted(per)[-5:]
        for name in per[last[0]]:
            vals[name] = sum(per[d][name] for d in last) / len(last)
        if grp == 'sq':
            dur = sum(times[d] for d in last) / len(last)
    if not vals:
        continue
    r = dict(vals)
    r['launch_ms_under_pmc'] = dur
    r['algorithmic_flops'] = ALG[w]['flops']
    r['algorithmic_bytes'] = ALG[w]['bytes']
    if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
        r['hbm_bytes_corrected'] = (2 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024
    if 'GRBM_GUI_ACTIVE' in r and dur:
        r['clock_GHz'] = r['GRBM_GUI_ACTIVE'] / 8 / (dur * 1e-3) / 1e9
        r['mfma_busy_frac'] = r['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * N_CU * r['GRBM_GUI_ACTIVE'] / 8)
        r['achieved_TFLOPs_under_pmc'] = ALG[w]['flops'] / (dur * 1e-3) / 1e12
    if 'SQ_INSTS_VALU_MFMA_F64' in r:
        r['useful_flop_frac'] = ALG[w]['flops'] / (r['SQ_INSTS_VALU_MFMA_F64'] * 2048 * 64 / 64)
    if 'TCC_HIT_sum' in r:
        r['l2_hit_rate'] = r['TCC_HIT_sum'] / (r['TCC_HIT_sum'] + r['TCC_MISS_sum'])
    out[{'theta': 'theta_chi4096_u1', 'uniform': 'uniform_4096cubed'}[w]] = r
path = f'profiles/{tag}_gemm_pmc_summary.json'
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out, indent=1))
print('wrote', path)
