O=gpurun_out/r02d; mkdir -p $O
one() { env $1 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', round(d['value'],1), round(d['roofline']['kernel_ms_avg'],3))" >> $O/ab.txt; }
for v in base x3_0 best_g slots128 slots160; do one RTIOW_GPU_LIB=rtiow-rust_amd/csrc/variants/$v.so $v; done
one A=1 current; one A=1 current
for rm in 40 44 48 52; do one RTG_REFILL_MIN=$rm refill_min_$rm; done
for rm in 40 48; do one "RTG_REFILL_MIN=$rm RTIOW_GPU_LIB=rtiow-rust_amd/csrc/variants/best_g.so" best_g_refill_$rm; done
for bl in 24 40; do one RTG_BOX_LEAVE=$bl box_leave_$bl; done
tools/profile_kernel.sh r02d_book1 book1 "render_lean_pool<true, false" 48000000 > $O/profile.log 2>&1
cat $O/ab.txt; cat gpurun_out/r02d_book1/roofline.json
