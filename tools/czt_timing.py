import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from grafx_amd import ops
for P,rows in [(135071,256),(483999,128),(483999,1024)]:
    z=torch.randn(rows,P,device='cuda')
    ops.odd_alias(z); torch.cuda.synchronize()
    t=time.time()
    for _ in range(3): y=ops.odd_alias(z)
    torch.cuda.synchronize(); dt=(time.time()-t)/3
    t=time.time()
    for _ in range(3): w=torch.fft.irfft(torch.fft.rfft(z))
    torch.cuda.synchronize(); dt2=(time.time()-t)/3
    print(P,rows,f"native {dt*1e3:.2f} ms  torch.fft fp32 {dt2*1e3:.2f} ms  err {((y-w).abs().max()/w.abs().max()).item():.2e}")
