O=gpurun_out/r02l; mkdir -p $O
for v in "" rtiow-rust_amd/csrc/variants/soa.so; do env ${v:+RTIOW_GPU_LIB=$v} timeout 300 python tools/time_scenes.py book2 800 800 100 cornell 300 300 100 cornell_smoke 300 300 100 volume 300 300 100 book2_bvh 800 800 100 simple_light 300 300 20 2>&1 | grep -v "^\[" | sed "s#^#[$v] #" >> $O/t.txt; done
cat $O/t.txt
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 -k "not book1 and not lean and not c2 and not c3" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
