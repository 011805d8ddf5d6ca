import sys; sys.path.insert(0,'/root/repo/tf-mpc_amd'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, torch, problems
from tfmpc.solvers.lqr import LQR
from oracle import lqr_ref
g=np.load('/root/repo/tests/golden/lqr_cfg3.npz'); T=int(g['T'])
for i in range(3):
    F,f,C,c,x0=(g[f"{k}{i}"] for k in ("F","f","C","c","x0"))
    lqr=LQR(F,f,C,c)
    for wv in (False,True):
        out=lqr.solve_device(x0,T,want_policy=True,want_value=wv); torch.cuda.synchronize()
        st=int(out['status'][0])
        K=out['K'][0].cpu().numpy().reshape(g[f'K{i}'].shape); x=out['states'][0].cpu().numpy().reshape(g[f'states{i}'].shape)
        print(i,wv,'status',st,'K err',np.abs(K-g[f'K{i}']).max()/np.abs(g[f'K{i}']).max(),'x err',np.abs(x-g[f'states{i}']).max()/np.abs(g[f'states{i}']).max(), 'K[T-1] err', np.abs(K[-1]-g[f'K{i}'][-1]).max(), 'K[0] err', np.abs(K[0]-g[f'K{i}'][0]).max())
        if wv: print('  const', out['const'][0].flatten()[:3].cpu().numpy(), g[f'const{i}'].flatten()[:3], 'V err', np.abs(out['V'][0].cpu().numpy().reshape(g[f'V{i}'].shape)-g[f'V{i}']).max()/np.abs(g[f'V{i}']).max())
