set -x
O=gpurun_out/r02a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tools/ubench/issue_rate > $O/issue_rate.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
tools/profile_kernel.sh r02a_book1 book1 "render_lean_pool<true, false" 48000000 > $O/profile.log 2>&1
# experiment: per-sample colour scratch written with plain (L2 write-back) stores instead of non-temporal ones
make -C rtiow-rust_amd/csrc -B EXTRA=-DRT_NT_SCRATCH=0 librtiow_gpu.so > $O/make_nt0.log 2>&1
python bench.py --no-cpu-baseline > $O/bench_nt0.json 2>> $O/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02a_nt0_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc.py $O/pmc_nt0_write.json book1 "render_lean_pool<true, false" 48000000 gpurun_out/r02a_nt0_write > /dev/null
rm -rf gpurun_out/r02a_nt0_write
tail -3 $O/pytest.log; cat $O/issue_rate.txt; cat $O/bench.json; cat $O/r02a_book1/roofline.json 2>/dev/null; cat $O/pmc_nt0_write.json | head -20
