cd $GRAFT_REPO_ROOT
for o in "" "farneback.strict_variant=2" "farneback.strict_variant=2,farneback.lds_pad=40000" "farneback.strict_variant=2,farneback.lds_pad=26000" "farneback.lds_pad=40000"; do
  echo "== opts '$o'"; BENCH_CTX_OPTIONS="$o" timeout 300 python bench.py --steps 60 --warmup 10 --repeats 5 --pairs 3 --no-cpu-baseline --no-extra-legs | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['value_one_pair_in_flight'], l['roofline']['avg_launch_us'])"
done
