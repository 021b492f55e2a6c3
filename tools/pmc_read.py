import csv, glob, os, sys, collections
ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag in sys.argv[1:]:
    for f in glob.glob(os.path.join(ROOT, 'gpurun_out/pmc_%s/*/*counter_collection.csv' % tag)):
        d=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'render_pass' not in r['Kernel_Name']: continue
            dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
            if dur>150: d[r['Counter_Name']].append((float(r['Counter_Value']),dur))
        for k,v in d.items(): print(tag, k, "%.4g"%v[-1][0], "%.1f ms"%v[-1][1])
