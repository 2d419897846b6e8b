cd /root/repo
timeout 900 python -m pytest tests/test_inpaint_gpu.py -x -q 2>&1 | tail -3
timeout 300 python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
import inspect
print([n for n in dir(synth) if 'inpaint' in n or 'hole' in n or 'mask' in n])
PY
