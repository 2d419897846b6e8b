cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
( time python bench.py ) > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err
tail -5 gpurun_out/r2e/bench.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r2e/bench.json").read().strip().splitlines()[-1])
print(json.dumps(l, indent=1)[:6000])
PY
