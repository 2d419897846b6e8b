#!/bin/bash
# VALU instruction count of one pyramid mean-shift filter call at 3840x2160 (BASELINE config 4); run on the GPU box.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_segment; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ctx = ofxcv.Context(0)
fr = np.ascontiguousarray(synth.inpaint_frame(3840, 2160, n_holes=0)[..., :3])
d = torch.from_numpy(fr).cuda()
for _ in range(3):
    ctx.pyr_mean_shift_filtering(d)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/p -o p -- python $OUT/one.py > $OUT/p.log 2>&1
python - <<'PY'
import csv, glob, os, json, collections, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_segment"
agg = collections.defaultdict(float)
calls = collections.defaultdict(int)
for f in glob.glob(out + "/p/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].find("ms_") >= 0 and r["Counter_Name"] == "SQ_INSTS_VALU":
            n = re.search(r"(ms_\w+)", r["Kernel_Name"]).group(1)
            agg[n] += float(r["Counter_Value"])
            calls[n] += 1
res = {"filter_calls": 3, "valu_wave_instructions_per_filter_call": sum(agg.values()) / 3,
       "per_kernel_per_filter_call": {k: v / 3 for k, v in agg.items()}, "launches_per_filter_call": {k: v / 3 for k, v in calls.items()},
       "method": "rocprofv3 --pmc SQ_INSTS_VALU over 3 calls of ofxcv_pyr_mean_shift_filtering(3840x2160, sp 10, sr 20, maxLevel 2), tools/pmc_segment.sh"}
json.dump(res, open(out + "/segment_valu.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
