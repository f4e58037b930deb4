O=gpurun_out/r02h; mkdir -p $O
timeout 60 python tools/try_case.py book1 stats >> $O/try.txt 2>&1; echo "rc=$?" >> $O/try.txt
timeout 60 python tools/try_case.py book1 176 112 12 >> $O/try.txt 2>&1; echo "rc=$?" >> $O/try.txt
tail -6 $O/try.txt
one() { env $1 timeout 120 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', round(d['value'],1), round(d['roofline']['kernel_ms_avg'],3))" >> $O/ab.txt; }
one RTG_ILP=2 ilp2; one RTG_ILP=2 ilp2; one RTG_ILP=1 ilp1
for rm in 20 28 36 44; do one "RTG_ILP=2 RTG_REFILL_MIN=$rm" ilp2_refill_$rm; done
for bl in 16 24 40; do one "RTG_ILP=2 RTG_BOX_LEAVE=$bl" ilp2_box_leave_$bl; done
for sm in 8 24; do one "RTG_ILP=2 RTG_SPHERE_MIN=$sm" ilp2_sphere_min_$sm; done
cat $O/ab.txt
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 -k "book1 or lean or c2 or c3 or variant or cost_ordered or shard or tile or reuse or degenerate or multi or albedo or bounce" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
