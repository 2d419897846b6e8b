import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
def gray(w,h,seed):
    c = ofxcv.Context(0)
    a,b = synth.flow_pair(w,h,seed=seed)
    ga = c.to_byte_grayscale(torch.from_numpy(a).cuda()); gb = c.to_byte_grayscale(torch.from_numpy(b).cuda())
    torch.cuda.synchronize(); c.close()
    return ga, gb
for (w,h,n,kw) in [(333,257,1,{}),(333,257,2,{}),(640,480,1,{}),(640,480,2,{}),(640,480,1,dict(levels=1)),(640,480,1,dict(levels=1,iterations=2)),(320,240,1,dict(levels=0,iterations=2)),(320,240,1,dict(levels=0,iterations=3)),(320,240,1,dict(levels=0))]:
    prs=[gray(w,h,7+i) for i in range(n)]
    outs=[]
    for p in (0,1):
        c=ofxcv.Context(0); c.set_option("farneback.persist",p)
        for _ in range(2):
            fl=c.calc_optical_flow_farneback_batch([a for a,_ in prs],[b for _,b in prs],**kw)
        torch.cuda.synchronize()
        outs.append([f.cpu().numpy() for f in fl]); ab=c.get_option("farneback.persist_aborts"); c.close()
    for z in range(n):
        d=np.abs(outs[0][z]-outs[1][z]); bad=np.argwhere(d.max(axis=2)>0)
        print(w,h,n,kw,"pair",z,"aborts",ab,"maxdiff %.3g"%d.max(),"nbad",len(bad),"first bad (y,x)",bad[0] if len(bad) else None, "rows with diffs:", (np.unique(bad[:,0])[:6] if len(bad) else None), flush=True)
