import csv, glob, sys
f = glob.glob('gpurun_out/r05f/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print(len(rows), 'kernels')
prev_end = None
for i, r in enumerate(rows):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (e - s) / 1e6
    name = r['Kernel_Name'][:70]
    if ('gusto' in name or 'locp' in name) and d > 20 and d < 200:
        ctx = rows[max(0, i - 2):i + 2]
        print('--- %.2f ms %s grid %s lds %s scratch %s' % (d, name, r.get('Grid_Size'), r.get('LDS_Block_Size'), r.get('Scratch_Size')))
        for c in ctx:
            print('     %.3f ms  %s (gap before: %.3f ms)' % ((int(c['End_Timestamp']) - int(c['Start_Timestamp'])) / 1e6, c['Kernel_Name'][:60], 0))
