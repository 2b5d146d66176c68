import sys,time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from _util import load_map_image, bench_start_poses
import f1tenth_gym_amd as amd
img,res,origin=load_map_image("example_map")
E,A=32768,2
for cache in (4096, 16):
    s=amd.BatchSim(num_envs=E,num_agents=A); s.set_map_image(img,res,origin); s.set_noise_rng(12345,0.01,cache_rows=cache)
    poses=bench_start_poses(E,A); d=s.device_array((E*A,3)); d.upload(poses); s.reset_device(d); s.set_auto_reseat(d,0,None)
    act=s.device_array((E*A,2)); act.upload(np.tile(np.array([[0.05,4.0]]),(E*A,1)))
    for t in range(60): s.step_device(act)
    s.sync(); t0=time.perf_counter()
    for t in range(200): s.step_device(act)
    s.sync(); print("cache rows",cache,"ms/step %.4f"%((time.perf_counter()-t0)/200*1e3), "max step_count", int(s.get("step_count")["step_count"].max()))
    s.close()
