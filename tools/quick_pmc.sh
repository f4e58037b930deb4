#!/bin/bash
# usage (GPU box): tools/quick_pmc.sh "<time_scenes.py args>" <kernel substring> [ENV=VAL ...]
# ONE rocprofv3 --pmc pass (SQ instruction / cycle counters, --kernel-trace only) over tools/time_scenes.py (no torch import: seconds),
# averaged over the dispatches of the named kernel: wave-instructions, lane utilisation, issue share.  For A/B work on a schedule.
SC=$1; K=$2; shift 2
R=$GRAFT_REPO_ROOT; D=/tmp/qpmc_$$; cd /tmp; export TMPDIR=/tmp
env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace --output-format csv -d $D -- python $R/tools/time_scenes.py $SC > $D.log 2>&1
python - "$D" "$K" <<'PY'
import collections, csv, glob, sys
d, k = sys.argv[1:3]
c = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            c[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
dur = []
for f in glob.glob(d + '/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6)
s = {n: sum(v.values()) / len(v) for n, v in c.items()}
if not s:
    sys.exit("no dispatch of a kernel matching %r" % k)
print("%d dispatches, %.2f ms each (profiled)" % (len(dur), sum(dur) / max(1, len(dur))))
print("VALU wave-instructions %.3f G  SALU %.3f G  lanes %.1f of 64  active-VALU share of wave-cycles %.3f  wait-any %.3f  wait-inst %.3f" % (
    s['SQ_INSTS_VALU'] / 1e9, s['SQ_INSTS_SALU'] / 1e9, s['SQ_THREAD_CYCLES_VALU'] / s['SQ_ACTIVE_INST_VALU'],
    s['SQ_ACTIVE_INST_VALU'] / s['SQ_WAVE_CYCLES'], s['SQ_WAIT_ANY'] / s['SQ_WAVE_CYCLES'], s['SQ_WAIT_INST_ANY'] / s['SQ_WAVE_CYCLES']))
PY
rm -rf $D $D.log
