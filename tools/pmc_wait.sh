#!/bin/bash
# usage (GPU box): tools/pmc_wait.sh <out-dir> [bench args]  -- where do the waves of the render kernel WAIT?  One rocprofv3 --pmc pass
# (with --kernel-trace only) over the in-flight level counters: SQ_INST_LEVEL_x accumulates, every cycle, the x instructions a wave
# has issued and not yet got back, so LEVEL_x / INSTS_x = the mean latency of one x instruction and LEVEL_x / WAVE_CYCLES = the share
# of a wave's life it spends with an x instruction outstanding (an upper bound of what s_waitcnt can lose to x).
O=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/$O
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d $R/$O/raw -- python $R/bench.py "$@" --steps 2 --warmup 0 --no-cpu-baseline --no-also > $R/$O/run.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $R/$O/raw2 -- python $R/bench.py "$@" --steps 2 --warmup 0 --no-cpu-baseline --no-also > $R/$O/run2.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for kern in ("render_lean_pool<true, false", "render_full_pool<1, true, false", "render_full_pool2<true, false", "render_full_sync<1, false, false"):
    c = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(O + '/raw*/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                c[r['Counter_Name']][(f, r['Dispatch_Id'])] += float(r['Counter_Value'])
    if not c:
        continue
    v = {k: sum(d.values()) / len(d) for k, d in c.items()}
    wc = v.get('SQ_WAVE_CYCLES', 0.0)
    print("kernel %s...: per launch" % kern)
    for k in sorted(v):
        print("  %-22s %16.0f%s" % (k, v[k], ("   = %.1f %% of the wave-cycles" % (100 * v[k] / wc)) if wc and 'INSTS' not in k else ""))
    for x in ("LDS", "VMEM", "SMEM"):
        if v.get('SQ_INSTS_' + x):
            print("  mean latency of one %-4s instruction: %.0f cycles in flight (SQ_INST_LEVEL_%s / SQ_INSTS_%s)" % (x, v['SQ_INST_LEVEL_' + x] / v['SQ_INSTS_' + x], x, x))
PY
rm -rf $O/raw $O/raw2
