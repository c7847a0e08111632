import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import torch, numpy as np, cases
from oracle import gmodule as og
from ttdg_mgm_amd import ops
dev=torch.device('cuda:0')
for name,sizes,seed in cases.PLANTED_CASES:
    params,nodes,labels,U,_=cases.mgm_inputs(name)
    otr={}; og.mgm3_unsup_forward(params,nodes,labels,U,trace=otr)
    off,blocks=0,[]
    for n in sizes:
        blocks.append(otr['A'][off:off+n,off:off+n].reshape(-1)); off+=n
    Ug,info,V0=ops.gagm_solve(torch.cat(blocks).to(dev),otr['Wds'].to(dev),otr['U0'].to(dev),ops.graphs(sizes),list(sizes))
    d=(Ug.cpu()!=otr['Ub']).any(1).sum().item()
    print(name,'dev',info.cpu().tolist()[:6],'oracle',otr['iters'],'rows differ',d)
