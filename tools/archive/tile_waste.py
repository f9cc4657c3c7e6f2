import sys,torch
sys.path[:0]=[".","motion-policy-networks_amd"]
from mpinets_amd.model import MotionPolicyNetwork
from mpinets_amd.rollout import RolloutEngine
from mpinets_amd.scenes import make_problem_batch
dev=torch.device("cuda:0"); torch.manual_seed(0)
m=MotionPolicyNetwork().to(dev).eval()
p=make_problem_batch(8192,seed=1000,device=dev,kinds=("tabletop","cubby","dresser"),M1=40,M2=16,scene_pool=1024,device_clouds=True)
e=RolloutEngine(m,p,rerender_scene=True,scene_seed=17,resample_subset=True,subset_seed=23)
e.step(); e.step(); torch.cuda.synchronize()
c1,c2=m.point_cloud_encoder.last_counts
for name,c,q,gr in (("SA1",c1,32,2),("SA2",c2,8,4)):
    c=c.clamp(1,128).long()
    rows=c.sum().item(); r=((c+gr-1)//gr*gr); rr=r.sum().item()
    tiles=((r.reshape(-1,q).sum(1)+31)//32).sum().item()
    print(name,"mean cnt",c.float().mean().item(),"rows",rows,"rounded",rr,"tile rows",tiles*32,"waste vs distinct",tiles*32/rows)
    for q2 in (q*2,q*4):
        t2=((r.reshape(-1,q2).sum(1)+31)//32).sum().item(); print("   Q",q2,"tile rows",t2*32, t2*32/rows)
