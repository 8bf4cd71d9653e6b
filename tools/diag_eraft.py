import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from test_gpu_eraft import make_net, DEV
from eemflow_amd.weights import synthetic_voxel_pair
g = np.load('/root/repo/tests/golden/eraft_fwd_128x160.npz')
h,w = g['hw'].tolist()
net,_ = make_net(int(g['seed'])); net.change_imagesize((h,w))
e1,e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(int(g['input_seed']), 1, h, w))
with torch.no_grad(): preds = net(e1,e2,iters=3)[1]
def err(a,b): 
    a=a.cpu().float(); b=torch.as_tensor(b); return float((a-b).abs().max()), float(b.abs().max())
fm = net.stage('fmap')
print('fmap1', err(fm[:1], g['fmap1'])); print('fmap2', err(fm[1:], g['fmap2']))
for k in ('inp','pyr1','pyr3','corr0','net1','delta1','mask1'):
    print(k, err(net.stage(k), g[k]))
for i in range(3): print('pred',i, err(preds[i], g['preds'][i]))
