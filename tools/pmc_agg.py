import csv, glob, sys, collections
d=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'conv_igemm' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    print(k[:90])
    for c,x in sorted(v.items()): print(f"   {c:32s} {x:16.0f}")
    if 'SQ_WAVE_CYCLES' in v:
        w=v['SQ_WAVE_CYCLES']
        for c in ('SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_WAIT_INST_LDS'):
            print(f"   {c}/WAVE_CYCLES = {v[c]/w:.3f}")
        print(f"   LDS conflict frac = {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.3f};  LDS_IDX_ACTIVE/BUSY_CYCLES = {v['SQ_LDS_IDX_ACTIVE']/max(v['SQ_BUSY_CYCLES'],1):.3f}")
