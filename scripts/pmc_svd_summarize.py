"""Summarise the PMC passes of scripts/pmc_svd.sh into profiles/<tag>_svd_pmc_summary.json: per kernel of the batched SVD
(chi=4096 theta list) the mean per launch of duration, HBM bytes ((2*FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of
MI355X_MICROARCH.md), L2 hit rate and MFMA busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE/8))."""
import csv, glob, json, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
KERNELS = ('jacobi_gram_kernel', 'jacobi_update_kernel', 'qr_panel_reg_kernel', 'gemm_grouped_kernel', 'svd_small_kernel')
acc = {k: defaultdict(float) for k in KERNELS}
cnt = {k: defaultdict(int) for k in KERNELS}
dur = {k: [0.0, 0] for k in KERNELS}
for grp in ('fetch', 'write', 'sq'):
    files = glob.glob(f'gpurun_out/pmc_{tag}_svd_{grp}/**/*counter_collection.csv', recursive=True)
    if not files:
        continue
    seen = set()
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            k = next((k for k in KERNELS if k in row['Kernel_Name']), None)
            if k is None:
                continue
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
            key = (row['Dispatch_Id'], row['Counter_Name'])
            if key not in seen:
                seen.add(key)
                cnt[k][row['Counter_Name']] += 1
            if grp == 'sq' and (row['Dispatch_Id'], 't') not in seen:
                seen.add((row['Dispatch_Id'], 't'))
                dur[k][0] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-3
                dur[k][1] += 1
out = {'note': __doc__.strip()}
for k in KERNELS:
    if not dur[k][1]:
        continue
    r = {name: acc[k][name] / max(cnt[k][name], 1) for name in acc[k]}
    r['launches'] = dur[k][1]
    r['mean_us_under_pmc'] = dur[k][0] / dur[k][1]
    if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
        r['hbm_bytes_per_launch'] = (2 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024
        r['hbm_GBps'] = r['hbm_bytes_per_launch'] / (r['mean_us_under_pmc'] * 1e-6) / 1e9
    if 'GRBM_GUI_ACTIVE' in r:
        r['mfma_busy_frac'] = r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (4 * 256 * r['GRBM_GUI_ACTIVE'] / 8)
    if 'TCC_HIT_sum' in r:
        r['l2_hit_rate'] = r['TCC_HIT_sum'] / max(r['TCC_HIT_sum'] + r['TCC_MISS_sum'], 1.0)
    out[k] = r
path = f'profiles/{tag}_svd_pmc_summary.json'
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out, indent=1))
print('wrote', path)
