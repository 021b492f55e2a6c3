"""per-layer duration of the SR stage's convolutions for each forced row count of conv3x3_limb16_kernel (NVSR_CV16_ROWS=2|3|4) and for the
launcher's own choice: reads the rocprofv3 kernel traces that tools/sr_rows_per_layer.sh wrote and prints, per launch of a step, the
durations and which variant wins -- i.e. what a perfect per-layer choice would gain over the cost model."""
import csv, glob, os, sys
root = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def launches(tag):
    f = glob.glob(os.path.join(root, 'gpurun_out', 'srrows_%s' % tag, '*', '*kernel_trace.csv'))[0]
    rows = [r for r in csv.DictReader(open(f)) if 'conv3x3' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    per_step = 70
    last = rows[-per_step:]                      # the last step of the run
    return [((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][11:45], r.get('Grid_Size_X', r.get('Grid_Size', ''))) for r in last]
tags = ['auto', '2', '3', '4']
data = {t: launches(t) for t in tags}
tot = {t: sum(x[0] for x in data[t]) for t in tags}
best = 0.0
for i in range(70):
    d = {t: data[t][i][0] for t in tags}
    b = min(d[t] for t in ('2', '3', '4'))
    best += b if 'limb16' in data['4'][i][1] else d['auto']
    if i < 8 or i > 60 or abs(d['auto'] - b) > 0.03 * b:
        print('%2d %-28s auto %7.1f us | rows2 %7.1f rows3 %7.1f rows4 %7.1f | auto picked %s' % (i, data['4'][i][1][:28], d['auto'], d['2'], d['3'], d['4'], data['auto'][i][1][-6:]))
print('totals (us):', {t: round(v) for t, v in tot.items()}, 'perfect per-layer choice: %d' % round(best))
