O=gpurun_out/r02f; mkdir -p $O
for c in "cornell_smoke" "cornell_smoke stats" "fuzzb:0" "fuzzb:0 stats" "fuzzb:1" "fuzzb:2"; do
  echo "== $c" >> $O/try.txt
  timeout 60 python tools/try_case.py $c >> $O/try.txt 2>&1; echo "rc=$?" >> $O/try.txt
done
cat $O/try.txt
