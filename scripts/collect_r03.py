"""Copy what `bash scripts/pmc_r03.sh && python3 scripts/pmc_summarize_r03.py` and `bash scripts/measure_r03.sh` left under
gpurun_out/ into profiles/ (the files profiles/README.md lists) and print the numbers DESIGN.md section 6 quotes."""
import csv, json, os, shutil
R = 'gpurun_out/r03'
for f in ('r03_gemm_pmc_summary.json', 'r03_svd_pmc_summary.json'):
    if os.path.exists(f'gpurun_out/pmc_json/{f}'):
        shutil.copy(f'gpurun_out/pmc_json/{f}', f'profiles/{f}')
shutil.copy(f'{R}/prof_bench/run_kernel_stats.csv', 'profiles/r03_bench_kernel_stats.csv')
shutil.copy(f'{R}/prof_dmrg/run_kernel_stats.csv', 'profiles/r03_dmrg_chi256_kernel_stats.csv')
for src, dst in (('bench', 'r03_bench.json.log'), ('bench_u1u1', 'r03_bench_u1u1.json.log')):
    open(f'profiles/{dst}', 'w').write(open(f'{R}/{src}.json').read().strip().splitlines()[-1] + '\n')
if os.path.exists(f'{R}/csvd.log'):
    open('profiles/r03_complex_bench.log', 'w').write(''.join(l for l in open(f'{R}/csvd.log') if l.startswith('[c')))
open('profiles/r03_shard_model.log', 'w').write(''.join(l for l in open(f'{R}/shard_model.log') if l.startswith('[shard]')))
for f in ('bench', 'bench_u1u1', 'bench_chi1024'):
    d = json.loads(open(f'{R}/{f}.json').read().strip().splitlines()[-1])
    r, s = d.get('roofline', {}), d.get('roofline_svd', {})
    print(f, 'ms/step', d['ms_per_step'], 'GFLOP/s', d['value'], '| gemm frac', r.get('frac'), 'ms', r.get('avg_launch_ms'), 'traffic', r.get('traffic'),
          '| svd ms', s.get('avg_call_ms'), 'frac', s.get('frac'), 'traffic', s.get('traffic'))
d = json.loads(open(f'{R}/bench.json').read().strip().splitlines()[-1])
cb = d['cpu_baseline']
print('cpu', cb['seconds_per_step'], cb['sweep_seconds_per_step'], 'best', cb['best_threads'], 'single', cb['single_thread']['seconds_per_step_estimate'],
      'split', cb['split_best'], 'speedup', d['speedup_vs_cpu'])
print('torch svd', d['roofline_svd']['reference_same_hw']['seconds_per_list'], d['roofline_svd']['reference_same_hw']['speedup'],
      '| u1u1 gemm', d['roofline_u1u1']['frac'], d['roofline_u1u1']['avg_launch_ms'], '| truncating caller', d['truncating_caller']['ms_per_step'])
g = json.load(open('profiles/r03_gemm_pmc_summary.json'))
for k in ('theta_chi4096_u1', 'theta_chi4096_u1u1', 'uniform_4096cubed'):
    r = g[k]
    print(k, 'us', round(r['mean_us_under_pmc'], 1), 'TF', round(r['achieved_TFLOPs_under_pmc'], 1), 'busy', round(r['mfma_busy_frac'], 2), 'L2',
          round(r['l2_hit_rate'], 2), 'MB', round(r['hbm_bytes_corrected'] / 1e6), 'x', round(r['traffic_over_algorithmic'], 2))
sv = json.load(open('profiles/r03_svd_pmc_summary.json'))
for k, r in sv['kernels'].items():
    print(k, r['launches_averaged'] / 3, 'us', round(r['mean_us_under_pmc'], 1), 'MB', round(r.get('hbm_bytes_corrected', 0) / 1e6, 1), 'L2',
          round(r.get('l2_hit_rate', 0), 2), 'busy', round(r.get('mfma_busy_frac', 0), 3), 'GB/s', round(r.get('hbm_GBps', 0)))
print(sv['theta_chi4096_u1'])
for f, div, label in (('profiles/r03_bench_kernel_stats.csv', 4, 'step'), ('profiles/r03_dmrg_chi256_kernel_stats.csv', 806, 'bond')):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    n = sum(int(r['Calls']) for r in rows)
    print(f, 'kernel ms per', label, round(tot / 1e6 / div, 3), 'launches', round(n / div, 1))
    for r in rows[:7]:
        nm = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
        print(f"   {nm[:40]:40s} {int(r['Calls']) / div:8.1f} {float(r['TotalDurationNs']) / 1e6 / div:8.3f} ms avg {float(r['AverageNs']) / 1e3:8.1f} us")
for f in ('svd_theta4096.log', 'svd_lists.log', 'shard_model.log', 'dmrg_chi256.log', 'dmrg_chi512.log', 'cfg5.log', 'lanczos.log'):
    if not os.path.exists(f'{R}/{f}'):
        continue
    for l in open(f'{R}/{f}'):
        if l.startswith('[') :
            print(l.strip()[:260])
