import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from oracle import ref_layers as R
from pytorch_geometric_signed_directed_amd import graphs
n,e,h=100000,2000000,64
ei=torch.from_numpy(graphs.dsbm_for_edges(n,e,seed=0)[0])
g=torch.Generator().manual_seed(0)
xr=torch.randn(n,h,generator=g,requires_grad=True); xi=torch.randn(n,h,generator=g,requires_grad=True)
w=torch.randn(2,h,h,requires_grad=True); b=torch.zeros(h,requires_grad=True)
op=R.magnet_operator(ei,None,n,0.25,"sym",2.0)
for th in (8,16,32,64,128):
    torch.set_num_threads(th)
    def step():
        o=R.magnet_conv(xr,xi,op,w,b,True); (o[0].sum()+o[1].sum()).backward()
    step(); t=time.perf_counter(); step(); print(th, time.perf_counter()-t, flush=True)
