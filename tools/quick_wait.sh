#!/bin/bash
# usage (GPU box): tools/quick_wait.sh "<time_scenes.py args>" <kernel substring> [ENV=VAL ...]
# Where do the waves of a kernel wait?  Two rocprofv3 --pmc passes (--kernel-trace only) over tools/time_scenes.py: the in-flight level
# counters (SQ_INST_LEVEL_x / SQ_INSTS_x = mean latency of one x instruction; / SQ_WAVE_CYCLES = share of a wave's life with one outstanding).
SC=$1; K=$2; shift 2
R=$GRAFT_REPO_ROOT; D=/tmp/qw_$$; cd /tmp; export TMPDIR=/tmp
env "$@" rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d $D/a -- python $R/tools/time_scenes.py $SC > $D.log 2>&1
env "$@" rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d $D/b -- python $R/tools/time_scenes.py $SC >> $D.log 2>&1
env "$@" rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $D/c -- python $R/tools/time_scenes.py $SC >> $D.log 2>&1
python - "$D" "$K" <<'PY'
import collections, csv, glob, sys
d, k = sys.argv[1:3]
for sub in "abc":
    c = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + '/' + sub + '/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if k in r['Kernel_Name']:
                c[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    v = {n: sum(x.values()) / len(x) for n, x in c.items()}
    wc = v.get('SQ_WAVE_CYCLES', 0.0)
    for n in sorted(v):
        print("  %-22s %16.0f%s" % (n, v[n], ("   = %.1f %% of the wave-cycles" % (100 * v[n] / wc)) if wc and 'INSTS' not in n and n != 'SQ_IFETCH' else ""))
    for x in ("LDS", "VMEM", "SMEM"):
        if v.get('SQ_INSTS_' + x) and v.get('SQ_INST_LEVEL_' + x):
            print("  mean latency of one %-4s instruction: %.0f cycles in flight" % (x, v['SQ_INST_LEVEL_' + x] / v['SQ_INSTS_' + x]))
    if v.get('SQ_IFETCH') and v.get('SQ_IFETCH_LEVEL'):
        print("  mean latency of one instruction fetch: %.0f cycles" % (v['SQ_IFETCH_LEVEL'] / v['SQ_IFETCH']))
PY
rm -rf $D $D.log
