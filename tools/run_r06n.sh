#!/bin/bash
# round 6, call n: PMC passes of the r = 30 / r = 36 projection (MFMA-busy, instruction issue) -- which pipe saturates at the shipped basis size
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r06n_mfma -o m -- python3 $R/tools/pmc_proj_r36.py > $R/gpurun_out/r06n_a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r06n_sq -o s -- python3 $R/tools/pmc_proj_r36.py > $R/gpurun_out/r06n_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/r06n_mix -o x -- python3 $R/tools/pmc_proj_r36.py > $R/gpurun_out/r06n_c.log 2>&1
cd $R; python3 - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ('mfma', 'sq', 'mix'):
    f = glob.glob('gpurun_out/r06n_%s/**/*counter_collection.csv' % tag, recursive=True)
    if not f: print(tag, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k = row['Kernel_Name']
        if 'proj_kernel' not in k: continue
        name = 'r36' if 'proj_kernel<2, 1' in k else ('r30' if 'proj_kernel<2, 0' in k else k[:40])
        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
        acc[name]['_ns'].append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
    for name, cs in acc.items():
        for cn, vals in cs.items():
            vals = sorted(vals)
            out.setdefault(name, {})[cn + ('' if cn != '_ns' else '_' + tag)] = vals[len(vals) // 2]
for name, cs in out.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs:
        cs['mfma_busy_fraction'] = cs['SQ_VALU_MFMA_BUSY_CYCLES'] / ((cs['GRBM_GUI_ACTIVE'] / 8) * 256 * 4)
    if 'SQ_WAVE_CYCLES' in cs:
        cs['wave_cycles_waiting_share'] = cs.get('SQ_WAIT_INST_ANY', 0) / cs['SQ_WAVE_CYCLES']
        cs['wave_cycles_issuing_share'] = cs.get('SQ_ACTIVE_INST_ANY', 0) / cs['SQ_WAVE_CYCLES']
json.dump({'source': 'tools/run_r06n.sh: rocprofv3 --pmc passes over tools/pmc_proj_r36.py (B = 65536, n_f = 4884; medians over 8 launches; _ns_* = kernel duration under the pass)', 'kernels': out}, open('gpurun_out/r06n_proj_r36_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
