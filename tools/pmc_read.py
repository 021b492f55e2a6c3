"""read-out of tools/pmc.sh passes: for every tag, the counters of the LONGEST dispatch whose kernel name contains the filter
(default: render_pass).   python tools/pmc_read.py [--kernel substr] tag..."""
import collections
import csv
import glob
import os
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
kern = "render_pass"
if args and args[0] == "--kernel":
    kern, args = args[1], args[2:]
for tag in args:
    for f in glob.glob(os.path.join(ROOT, 'gpurun_out/pmc_%s/*/*counter_collection.csv' % tag)):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern not in r['Kernel_Name']:
                continue
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
            d[r['Counter_Name']].append((dur, float(r['Counter_Value']), r['Kernel_Name'][:40]))
        for k, v in d.items():
            dur, val, name = max(v)
            print("%-10s %-44s %16.6g   %.2f ms  %s" % (tag, k, val, dur, name))
