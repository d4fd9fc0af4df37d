"""Summaries of the rocprofv3 passes of tools/prof_r03.sh (gpurun_out/r03_*) -> small tracked JSON / CSV files under profiles/."""
import csv, glob, json, os, shutil, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')


def rows(pattern):
    f = glob.glob(os.path.join(G, pattern), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def dur_ms(r):
    return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6


# ---- kernel trace: stats table + per-launch durations of the B = 65536 projection launches (grid 131072 = 512 workgroups x 256)
st = glob.glob(os.path.join(G, 'r03_trace', '**', '*kernel_stats.csv'), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(P, 'r03_kernel_stats.csv'))
tr = rows('r03_trace/**/*kernel_trace.csv')
if tr:
    proj = [r for r in tr if r['Kernel_Name'].startswith('void (anonymous namespace)::proj_kernel<2, 0, true, true, false>')]
    big = [dur_ms(r) for r in proj if int(r.get('Grid_Size', r.get('Grid_Size_X', 0))) == 131072]
    alg = 8 * (65536 * 4884 + 4884 * 30 + 4884 + 65536 * 30)
    out = {'source': 'rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline --steps 3 --warmup 1` (tools/prof_r03.sh); the launches of '
                     'proj_kernel<2,0,true,true,false> with grid 131072 = the B = 65536 x n_f = 4884 x r = 30 projections of the bench step and of pod_shapes',
           'algorithmic_bytes_per_launch': alg, 'launches': len(big), 'ms': [round(x, 6) for x in big],
           'ms_mean': statistics.mean(big) if big else None, 'ms_median': statistics.median(big) if big else None,
           'first_16_mean_ms (the timed bench steps + warm-up)': statistics.mean(big[:16]) if len(big) >= 16 else None}
    if big:
        out['achieved_GBs_mean'] = alg / (out['ms_mean'] * 1e-3) / 1e9
        out['frac_of_8TBs'] = out['achieved_GBs_mean'] / 8000.0
    json.dump(out, open(os.path.join(P, 'r03_proj_launches.json'), 'w'), indent=1)
    agg = {}
    for r in tr:
        k = r['Kernel_Name'][:90]
        agg.setdefault(k, []).append(dur_ms(r))
    top = sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]
    json.dump({'source': 'same trace', 'kernels': [{'kernel': k, 'calls': len(v), 'total_ms': sum(v), 'max_ms': max(v), 'last_ms': v[-4:]} for k, v in top]},
              open(os.path.join(P, 'r03_kernel_launches_top.json'), 'w'), indent=1)

# ---- PMC: HBM traffic of the projection kernel
def counter_rows(d, name):
    rs = rows(d + '/**/*counter_collection.csv')
    return [r for r in rs if r.get('Counter_Name') == name]


f, wv = counter_rows('r03_pmc_fetch', 'FETCH_SIZE'), counter_rows('r03_pmc_write', 'WRITE_SIZE')
sel = lambda rs: [float(r['Counter_Value']) for r in rs if r['Kernel_Name'].startswith('void (anonymous namespace)::proj_kernel<2, 0, true, true, false>')]
fv, wvv = sel(f), sel(wv)
if fv and wvv:
    fm, wm = statistics.median(fv), statistics.median(wvv)
    json.dump({'kernel': 'proj_kernel<2, 0, true, true, false>', 'workload': 'B=65536, n_f=4884, r=30 (tools/pmc_kernels.py)',
               'FETCH_SIZE_KiB_median': fm, 'WRITE_SIZE_KiB_median': wm,
               'fetch_correction': 'x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section)',
               'traffic_bytes_per_launch': (2 * fm + wm) * 1024, 'algorithmic_bytes_per_launch': 8 * (65536 * 4884 + 4884 * 30 + 4884 + 65536 * 30),
               'launches_sampled': len(fv), 'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_r03.sh), round 3 tree'},
              open(os.path.join(P, 'r03_proj_pmc.json'), 'w'), indent=1)

# ---- PMC: MFMA-busy fraction of the SCP kernels of one bench step
m = rows('r03_pmc_mfma/**/*counter_collection.csv')
if m:
    byk = {}
    for r in m:
        key = (r['Kernel_Name'][:80], r.get('Dispatch_Id'))
        byk.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
    res = {}
    for (k, did), cnt in byk.items():
        if 'gusto' not in k and 'proj_kernel' not in k:
            continue
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in cnt and cnt.get('GRBM_GUI_ACTIVE', 0) > 0:
            res.setdefault(k, []).append({'dispatch': did, 'SQ_VALU_MFMA_BUSY_CYCLES': cnt['SQ_VALU_MFMA_BUSY_CYCLES'], 'GRBM_GUI_ACTIVE': cnt['GRBM_GUI_ACTIVE'],
                                          'mfma_busy_fraction': cnt['SQ_VALU_MFMA_BUSY_CYCLES'] / ((cnt['GRBM_GUI_ACTIVE'] / 8) * 256 * 4)})
    json.dump({'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0',
               'formula': 'SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs)', 'kernels': res},
              open(os.path.join(P, 'r03_mfma_util.json'), 'w'), indent=1)
for name in ('r03_bench_under_rocprof.log',):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, name))
print('summaries written')
