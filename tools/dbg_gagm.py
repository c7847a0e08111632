import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import torch, numpy as np, cases
from oracle import gmodule as og
from oracle.sinkhorn_spec import sinkhorn as osk
from ttdg_mgm_amd import ops
dev=torch.device('cuda:0')
for name,sizes,seed in cases.GAGM_CASES:
    A,W,U0=cases.gagm_inputs(sizes,seed)
    off,blocks=0,[]
    for n in sizes:
        blocks.append(A[off:off+n,off:off+n].reshape(-1)); off+=n
    tr={}
    Uo=og.gagm(A,W,U0.clone(),sizes,trace=tr)
    U,info,V0=ops.gagm_solve(torch.cat(blocks).to(dev),W.to(dev),U0.to(dev),ops.graphs(sizes),list(sizes))
    U1=ops.gagm_solve.last_U1.cpu()
    V0o=tr['V0']
    U1o=og._project_sinkhorn(V0o,list(sizes),32,0.1,20)
    if len(sizes)==2: U1o[:sizes[0]]=torch.eye(sizes[0],32)
    print(name,'V0 err',float((V0.cpu()-V0o).abs().max()),'U1 err',float((U1-U1o).abs().max()),'iters',info.cpu().tolist()[:7],tr['iters'],'U equal',bool(torch.equal(U.cpu(),Uo)))
