import torch, sys
sys.path.insert(0,'.')
from gcpnet_amd import ops
x=torch.randn(100000,896,device='cuda'); W=torch.randn(256,1049,device='cuda')*0.03
import time
for (xx, od, idim, tr) in [(x,256,896,False),(torch.randn(100000,256,device='cuda'),896,256,True)]:
    o=ops.wg_linear(xx,W,od,idim,col0=0,trans=tr)
    print('supported', o is not None)
    if o is not None:
        ref = xx@ (W[:, :896].t() if not tr else W[:, :896])
        print('err', float((o-ref).abs().max()), float(ref.abs().max()))
        torch.cuda.synchronize(); t=time.time()
        for _ in range(10): ops.wg_linear(xx,W,od,idim,col0=0,trans=tr)
        torch.cuda.synchronize(); print('wg ms', (time.time()-t)*100)
        t=time.time()
        for _ in range(10): xx@ (W[:, :896].t() if not tr else W[:, :896])
        torch.cuda.synchronize(); print('blas ms', (time.time()-t)*100)
