"""Summaries of the rocprofv3 passes of tools/prof_round.sh <tag> (gpurun_out/<tag>_*) -> small tracked JSON / CSV files under profiles/.
Usage: python3 tools/prof_summarise.py r04   (prof_r03_summarise.py is this script frozen at the round-3 tag)."""
import csv, glob, json, os, shutil, statistics, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r04'
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')


def rows(pattern):
    f = glob.glob(os.path.join(G, pattern), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def dur_ms(r):
    return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6


# ---- kernel trace: stats table + per-launch durations of the B = 65536 projection launches (grid 131072 = 512 workgroups x 256)
st = glob.glob(os.path.join(G, TAG + '_trace', '**', '*kernel_stats.csv'), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(P, TAG + '_kernel_stats.csv'))
tr = rows(TAG + '_trace/**/*kernel_trace.csv')
if tr:
    proj = [r for r in tr if r['Kernel_Name'].startswith('void (anonymous namespace)::proj_kernel<2, 0, true, true, false>')]
    big = [dur_ms(r) for r in proj if int(r.get('Grid_Size', r.get('Grid_Size_X', 0))) == 131072]
    alg = 8 * (65536 * 4884 + 4884 * 30 + 4884 + 65536 * 30)
    out = {'source': 'rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline --steps 3 --warmup 1` (tools/prof_round.sh); the launches of '
                     'proj_kernel<2,0,true,true,false> with grid 131072 = the B = 65536 x n_f = 4884 x r = 30 projections of the bench step and of pod_shapes',
           'algorithmic_bytes_per_launch': alg, 'launches': len(big), 'ms': [round(x, 6) for x in big],
           'ms_mean': statistics.mean(big) if big else None, 'ms_median': statistics.median(big) if big else None,
           'first_16_mean_ms (the timed bench steps + warm-up)': statistics.mean(big[:16]) if len(big) >= 16 else None}
    if big:
        out['achieved_GBs_mean'] = alg / (out['ms_mean'] * 1e-3) / 1e9
        out['frac_of_8TBs'] = out['achieved_GBs_mean'] / 8000.0
    json.dump(out, open(os.path.join(P, TAG + '_proj_launches.json'), 'w'), indent=1)
    agg = {}
    for r in tr:
        k = r['Kernel_Name'][:90]
        agg.setdefault(k, []).append(dur_ms(r))
    top = sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]
    json.dump({'source': 'same trace', 'kernels': [{'kernel': k, 'calls': len(v), 'total_ms': sum(v), 'max_ms': max(v), 'last_ms': v[-4:]} for k, v in top]},
              open(os.path.join(P, TAG + '_kernel_launches_top.json'), 'w'), indent=1)

# ---- the other POD launches of the same trace (pod_shapes of bench.py): per-launch durations by kernel instantiation and grid size
if tr:
    shapes = {}
    for r in tr:
        k = r['Kernel_Name']
        if not any(s in k for s in ('proj_kernel<', 'lift_kernel<', 'utmu_reduce_kernel', 'splitk_reduce')):
            continue
        name = k.replace('void (anonymous namespace)::', '').split('(')[0]
        shapes.setdefault((name, int(r.get('Grid_Size', r.get('Grid_Size_X', 0))), int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 0)))), []).append(dur_ms(r))
    rows_out = []
    for (name, grid, wgs), v in sorted(shapes.items(), key=lambda kv: -sum(kv[1])):
        rows_out.append({'kernel': name, 'grid_threads': grid, 'workgroup': wgs, 'workgroups': grid // wgs if wgs else None, 'launches': len(v),
                         'ms_mean': statistics.mean(v), 'ms_median': statistics.median(v), 'ms_min': min(v), 'ms_max': max(v),
                         'ms_last_32': [round(x, 6) for x in v[-32:]]})
    json.dump({'source': 'rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline --steps 3 --warmup 1` (tools/prof_round.sh): every launch of the POD kernels '
                         '(projection r = 30 / 36, both state forms; U^T M U = proj_kernel<..., true> + utmu_reduce_kernel; lift), grouped by instantiation and grid',
               'template_arguments': 'proj_kernel<NTF full 16-column tiles, NQ 4-column tiles, HAS_REF, VEC2, UTMU>; lift_kernel<KS, MT>',
               'algorithmic_bytes': {'project B=65536 r=30': 8 * (65536 * 4884 + 4884 * 30 + 4884 + 65536 * 30), 'project B=65536 r=36': 8 * (65536 * 4884 + 4884 * 36 + 4884 + 65536 * 36),
                                     'utmu': 8 * 4884 * 4884},
               'launch_groups': rows_out}, open(os.path.join(P, TAG + '_pod_shapes_launches.json'), 'w'), indent=1)

# ---- PMC: HBM traffic of the projection kernel
def counter_rows(d, name):
    rs = rows(d + '/**/*counter_collection.csv')
    return [r for r in rs if r.get('Counter_Name') == name]


f, wv = counter_rows(TAG + '_pmc_fetch', 'FETCH_SIZE'), counter_rows(TAG + '_pmc_write', 'WRITE_SIZE')
sel = lambda rs: [float(r['Counter_Value']) for r in rs if r['Kernel_Name'].startswith('void (anonymous namespace)::proj_kernel<2, 0, true, true, false>')]
fv, wvv = sel(f), sel(wv)
if fv and wvv:
    fm, wm = statistics.median(fv), statistics.median(wvv)
    json.dump({'kernel': 'proj_kernel<2, 0, true, true, false>', 'workload': 'B=65536, n_f=4884, r=30 (tools/pmc_kernels.py)',
               'FETCH_SIZE_KiB_median': fm, 'WRITE_SIZE_KiB_median': wm,
               'fetch_correction': 'x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section)',
               'traffic_bytes_per_launch': (2 * fm + wm) * 1024, 'algorithmic_bytes_per_launch': 8 * (65536 * 4884 + 4884 * 30 + 4884 + 65536 * 30),
               'launches_sampled': len(fv), 'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_round.sh), ' + TAG + ' tree'},
              open(os.path.join(P, TAG + '_proj_pmc.json'), 'w'), indent=1)

# the one-pass U^T M U of tools/pmc_kernels.py (M 4884 x 4884; the UTMU instantiation of the projection kernel + its reduction)
selu = lambda rs, pat: [float(r['Counter_Value']) for r in rs if pat in r['Kernel_Name']]
# tools/pmc_kernels.py launches the UTMU instantiation six times with one matrix, then six times with four (blockIdx.z = matrix): split by dispatch order
def _split(rs):
    u = sorted((r for r in rs if 'true, true>' in r['Kernel_Name']), key=lambda r: int(r['Dispatch_Id']))
    grids = [int(r['Grid_Size']) for r in u]
    if len(set(grids)) < 2:
        return [float(r['Counter_Value']) for r in u], []
    first = grids[0]
    return [float(r['Counter_Value']) for r in u if int(r['Grid_Size']) == first], [float(r['Counter_Value']) for r in u if int(r['Grid_Size']) != first]
_ug = [0, 1]
_one = lambda rs, pat, big: _split(rs)[1 if big else 0]
fu, wu = _one(f, 'true, true>', False), _one(wv, 'true, true>', False)
f4, w4 = (_one(f, 'true, true>', True), _one(wv, 'true, true>', True)) if len(_ug) > 1 else ([], [])
if f4 and w4:
    fm4, wm4 = statistics.median(f4), statistics.median(w4)
    json.dump({'kernel': 'proj_kernel<..., UTMU> with blockIdx.z = matrix (srom_reduce_matrices_dev, four matrices) -- the streaming launch of the pair',
               'workload': 'four M 4884 x 4884 f64, r = 30 (tools/pmc_kernels.py)', 'FETCH_SIZE_KiB_median': fm4, 'WRITE_SIZE_KiB_median': wm4,
               'fetch_correction': 'x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section)',
               'traffic_bytes_per_launch': (2 * fm4 + wm4) * 1024, 'algorithmic_bytes_per_launch': 4 * 8 * (4884 * 4884 + 2 * 4884 * 30 + 30 * 30),
               'traffic_over_algorithmic': (2 * fm4 + wm4) * 1024 / (4 * 8 * (4884 * 4884 + 2 * 4884 * 30 + 30 * 30)), 'launches_sampled': len(f4),
               'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_round.sh), ' + TAG + ' tree'},
              open(os.path.join(P, TAG + '_utmu4_pmc.json'), 'w'), indent=1)
if fu and wu:
    fm, wm = statistics.median(fu), statistics.median(wu)
    fr, wr = selu(f, 'utmu_reduce_kernel'), selu(wv, 'utmu_reduce_kernel')
    json.dump({'kernel': 'proj_kernel<2, 0, false, true, true> (U^T M U, r = 30) + utmu_reduce_kernel', 'workload': 'M 4884 x 4884 f64, r = 30 (tools/pmc_kernels.py)',
               'FETCH_SIZE_KiB_median': fm, 'WRITE_SIZE_KiB_median': wm, 'fetch_correction': 'x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section)',
               'traffic_bytes_per_launch': (2 * fm + wm) * 1024, 'algorithmic_bytes_per_launch': 8 * 4884 * 4884,
               'reduce_kernel_FETCH_SIZE_KiB_median': statistics.median(fr) if fr else None, 'reduce_kernel_WRITE_SIZE_KiB_median': statistics.median(wr) if wr else None,
               'launches_sampled': len(fu), 'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_round.sh), ' + TAG + ' tree'},
              open(os.path.join(P, TAG + '_utmu_pmc.json'), 'w'), indent=1)

# ---- PMC: MFMA-busy fraction of the SCP kernels of one bench step
m = rows(TAG + '_pmc_mfma/**/*counter_collection.csv')
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
              open(os.path.join(P, TAG + '_mfma_util.json'), 'w'), indent=1)
for name in (TAG + '_bench_under_rocprof.log',):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, name))

# ---- PMC: instruction mix, wave cycles (parked / issue-stalled / active), LDS conflicts, instruction cache of the lean GuSTO kernel
mix = {}
for sub in ('mix_a', 'mix_b', 'mix_c', 'mix_d', 'mix_e'):
    for r in rows(TAG + '_' + sub + '/**/*counter_collection.csv'):
        if 'gusto_lean_kernel' not in r['Kernel_Name']:
            continue
        mix.setdefault((r['Kernel_Name'], r['Dispatch_Id'], sub), {}).setdefault(r['Counter_Name'], 0.0)
        mix[(r['Kernel_Name'], r['Dispatch_Id'], sub)][r['Counter_Name']] += float(r['Counter_Value'])
if mix:
    counters, kernel = {}, None
    for sub in ('mix_a', 'mix_b', 'mix_c', 'mix_d', 'mix_e'):
        cand = {k: v for k, v in mix.items() if k[2] == sub}
        if cand:
            best = max(cand.items(), key=lambda kv: sum(kv[1].values()))        # the largest dispatch = the timed 4096-rollout launch
            counters.update(best[1]); kernel = best[0][0]
    c = counters
    tot = sum(c.get(k, 0.0) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM'))
    der = {}
    if tot:
        der.update(instructions_counted=tot, share_valu=c.get('SQ_INSTS_VALU', 0) / tot, share_salu=c.get('SQ_INSTS_SALU', 0) / tot,
                   share_lds=c.get('SQ_INSTS_LDS', 0) / tot, share_vmem_rd=c.get('SQ_INSTS_VMEM_RD', 0) / tot,
                   mfma_per_valu=c.get('SQ_INSTS_MFMA', 0) / max(1.0, c.get('SQ_INSTS_VALU', 0)))
    if c.get('SQ_WAVE_CYCLES'):
        wc = c['SQ_WAVE_CYCLES']
        der.update(wave_cycles_parked_share=c.get('SQ_WAIT_ANY', 0) / wc, wave_cycles_issue_stall_share=c.get('SQ_WAIT_INST_ANY', 0) / wc,
                   wave_cycles_active_share=c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                   quad_cycles_per_instruction=wc / tot if tot else None)
    if c.get('SQ_LDS_IDX_ACTIVE'):
        der['lds_bank_conflict_cycles_per_lds_active_cycle'] = c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']
    if c.get('SQC_ICACHE_REQ'):
        der['icache_miss_rate'] = c.get('SQC_ICACHE_MISSES', 0) / c['SQC_ICACHE_REQ']
    if c.get('SQ_INSTS_MFMA'):
        der['mfma_flop_per_launch'] = c['SQ_INSTS_MFMA'] * 2048.0
    json.dump({'source': 'rocprofv3 --pmc (separate passes, tools/prof_round.sh) over `python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0`; '
                         'the timed 4096-rollout launch of the lean GuSTO kernel (largest dispatch of that kernel); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles',
               'kernel': kernel, 'counters': counters, 'derived': der}, open(os.path.join(P, TAG + '_lean_instruction_mix.json'), 'w'), indent=1)
print('summaries written')
