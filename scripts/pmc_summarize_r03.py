"""Summarise the PMC passes written by scripts/pmc_r03.sh into profiles/r03_gemm_pmc_summary.json and
profiles/r03_svd_pmc_summary.json.

Per kernel, means per launch (GEMM: over the last 5 dispatches of gemm_grouped_kernel in every pass; SVD: over every
dispatch of each kernel of the batched call).  HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so hbm_bytes = (2*FETCH_SIZE +
WRITE_SIZE)*1024 (memory-side requests of the L2s: Infinity-Cache hits are counted).  Clock = GRBM_GUI_ACTIVE / 8 / kernel
time.  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x n_cu x GRBM_GUI_ACTIVE/8); v_mfma_f64_16x16x4 occupies
the pipe 64 cycles.  `source_sha16` records the kernel sources the passes ran on: bench.py quotes `traffic` only while
they are unchanged."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CU = 256


def sha16(name):
    return hashlib.sha256(open(os.path.join(ROOT, 'cyten_amd', 'csrc', name), 'rb').read()).hexdigest()[:16]


def passes(workload, kernels, last=None):
    """{kernel: {counter or derived: mean per launch}}"""
    acc = {k: defaultdict(float) for k in kernels}
    nd = {k: defaultdict(int) for k in kernels}
    dur = {k: [0.0, 0] for k in kernels}
    for grp in ('fetch', 'write', 'sq'):
        files = glob.glob(f'gpurun_out/pmc_r03_{workload}_{grp}/**/*counter_collection.csv', recursive=True)
        if not files:
            continue
        per = {k: defaultdict(lambda: defaultdict(float)) for k in kernels}
        times = {k: {} for k in kernels}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                k = next((k for k in kernels if k in row['Kernel_Name']), None)
                if k is None:
                    continue
                d = int(row['Dispatch_Id'])
                per[k][d][row['Counter_Name']] += float(row['Counter_Value'])
                times[k][d] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-3
        for k in kernels:
            ds = sorted(per[k])
            if last:
                ds = ds[-last:]
            for d in ds:
                for name, v in per[k][d].items():
                    acc[k][name] += v
                    nd[k][name] += 1
                if grp == 'sq':
                    dur[k][0] += times[k][d]
                    dur[k][1] += 1
    out = {}
    for k in kernels:
        if not dur[k][1]:
            continue
        r = {name: acc[k][name] / max(nd[k][name], 1) for name in acc[k]}
        r['launches_averaged'] = dur[k][1]
        r['mean_us_under_pmc'] = dur[k][0] / dur[k][1]
        if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
            r['hbm_bytes_corrected'] = (2 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024
            r['hbm_GBps'] = r['hbm_bytes_corrected'] / (r['mean_us_under_pmc'] * 1e-6) / 1e9
        if 'GRBM_GUI_ACTIVE' in r:
            r['clock_GHz'] = r['GRBM_GUI_ACTIVE'] / 8 / (r['mean_us_under_pmc'] * 1e-6) / 1e9
            r['mfma_busy_frac'] = r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (4 * N_CU * r['GRBM_GUI_ACTIVE'] / 8)
        if 'TCC_HIT_sum' in r:
            r['l2_hit_rate'] = r['TCC_HIT_sum'] / max(r['TCC_HIT_sum'] + r['TCC_MISS_sum'], 1.0)
        out[k] = r
    return out


ALG = {'theta_chi4096_u1': ('gemm_u1', 10798826696.0, 210354688.0), 'theta_chi4096_u1u1': ('gemm_u1u1', 5588732616.0, 281915368.0),
       'uniform_4096cubed': ('gemm_uniform', 2.0 * 4096 ** 3, 3 * 8.0 * 4096 ** 2)}
gemm = {'note': __doc__.strip(), 'source_sha16': {'gemm_grouped.hip': sha16('gemm_grouped.hip')}}
for key, (w, flops, nbytes) in ALG.items():
    r = passes(w, ('gemm_grouped_kernel',), last=5).get('gemm_grouped_kernel')
    if not r:
        continue
    r['algorithmic_flops'], r['algorithmic_bytes'] = flops, nbytes
    r['achieved_TFLOPs_under_pmc'] = flops / (r['mean_us_under_pmc'] * 1e-6) / 1e12
    if 'hbm_bytes_corrected' in r:
        r['traffic_over_algorithmic'] = r['hbm_bytes_corrected'] / nbytes
    if 'SQ_INSTS_VALU_MFMA_F64' in r:
        r['useful_flop_frac'] = flops / (r['SQ_INSTS_VALU_MFMA_F64'] * 2048)
    gemm[key] = r
json.dump(gemm, open(os.path.join(ROOT, 'profiles', 'r03_gemm_pmc_summary.json'), 'w'), indent=1)

SVD_KERNELS = ('jacobi_sweep_kernel', 'jacobi_round_kernel', 'jacobi_gram_kernel', 'jacobi_update_kernel', 'qr_panel_reg_kernel', 'qr_panel_wave', 'reflector_strip_kernel',
               'gemm_grouped_kernel', 'svd_small_kernel')
svd = {'note': __doc__.strip(), 'source_sha16': {s: sha16(s) for s in ('jacobi_engine.hip', 'svd_jacobi.hip', 'blocked_qr.hip')}}
per_kernel = passes('svd', SVD_KERNELS)
svd['kernels'] = per_kernel
# whole batched call: sum over kernels of (mean bytes per launch x launches) / calls of the probe (1 warm + 2 timed + ... = what ran)
calls = 3.0   # scripts/svd_bench.py theta4096: one warm call + reps = 2
tot = sum(r.get('hbm_bytes_corrected', 0.0) * r['launches_averaged'] for r in per_kernel.values()) / calls
svd['theta_chi4096_u1'] = {'hbm_bytes_corrected': tot, 'calls_in_the_passes': calls,
                           'launches_per_call': sum(r['launches_averaged'] for r in per_kernel.values()) / calls,
                           'algorithmic_bytes': 202779216.0}
json.dump(svd, open(os.path.join(ROOT, 'profiles', 'r03_svd_pmc_summary.json'), 'w'), indent=1)
for name, d in (('gemm', gemm), ('svd', svd)):
    print(name, json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in (
        'mean_us_under_pmc', 'hbm_bytes_corrected', 'traffic_over_algorithmic', 'l2_hit_rate', 'mfma_busy_frac', 'clock_GHz',
        'useful_flop_frac', 'launches_per_call', 'achieved_TFLOPs_under_pmc')}) for k, v in d.items() if k not in ('note', 'kernels')}, indent=1))
if 'kernels' in svd:
    for k, r in svd['kernels'].items():
        print(f"  {k}: {r['launches_averaged'] / calls:.0f} launches/call, {r['mean_us_under_pmc']:.1f} us, "
              f"{r.get('hbm_bytes_corrected', 0) / 1e6:.1f} MB/launch, L2 hit {r.get('l2_hit_rate', 0):.2f}, MFMA busy {r.get('mfma_busy_frac', 0):.3f}")
