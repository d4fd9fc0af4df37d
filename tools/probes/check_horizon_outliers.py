import json,sys
for f in sys.argv[1:]:
    try:
        l=[x for x in open(f).read().splitlines() if x.startswith('{')][-1]
        d=json.loads(l)
        print(f.split('/')[-1], {k:(round(v['ms_per_solve_max'],2), round(v['ms_per_solve_median'],3)) for k,v in d['secondary']['scp_reference_horizons'].items()}, 'c5 calls', [round(x,1) for x in d['secondary']['scp_c5']['ms_all_calls']])
    except Exception as e:
        print(f, 'ERR', e)
