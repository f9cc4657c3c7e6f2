import sys,time,os,numpy as np,torch
sys.path[:0]=[os.getcwd(),os.path.join(os.getcwd(),'motion-policy-networks_amd'),os.path.join(os.getcwd(),'tests')]
from oracle import oracle as orc
import seeded_weights
g=dict(np.load('tests/golden/model_golden.npz'))
shapes={str(n):tuple(int(x) for x in s[:np.count_nonzero(s)]) for n,s in zip(g['param_names'],g['param_shapes'])}
sd=seeded_weights.seeded_state_dict(shapes,0)
sdt={k:torch.tensor(v) for k,v in sd.items()}
xyz=np.tile(g['f_xyz'],(11,1,1))[:32]; q=np.tile(g['f_q'],(11,1))[:32]
orc.set_threads(os.cpu_count())
for nt in (8,16,32,64,128,256):
    torch.set_num_threads(nt)
    with torch.no_grad():
        orc.policy_forward_torch(sdt,xyz[:4],torch.tensor(q[:4]))
        t=time.time(); orc.policy_forward_torch(sdt,xyz,torch.tensor(q)); dt=time.time()-t
    print('torch threads',nt,'32 envs',round(dt,2),'s ->',round(32/dt,1),'env/s',flush=True)
t=time.time(); orc.policy_forward(sd,xyz,q); dt=time.time()-t; print('numpy f64 32 envs',round(dt,2),round(32/dt,1))
