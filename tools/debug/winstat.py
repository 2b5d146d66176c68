import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from _util import load_map_image, bench_start_poses
import f1tenth_gym_amd as amd
img,res,origin=load_map_image("example_map")
E,A=512,2
s=amd.BatchSim(num_envs=E,num_agents=A,map_layout=4); s.set_map_image(img,res,origin)
poses=bench_start_poses(E,A); s.reset(poses)
s.scan_lookup_count(enable=True)
rng=np.random.default_rng(0)
act=np.stack([rng.uniform(-0.2,0.2,E*A),rng.uniform(2,6,E*A)],axis=1)
for t in range(20): s.step(act)
tot,lds=s.scan_lookup_count(enable=False,detail=True)
print("lookups",tot,"lds-served",lds, "frac of post-first lookups %.3f"%(lds/(tot-E*A*1080*20)))
