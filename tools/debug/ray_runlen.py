"""Long rays of the bench workload, on the CPU (oracle rollout to the steady regime + a vectorised NumPy restatement of
trace_ray that records every sample's table value for the rays that pass 32 samples):
  * how long rays get and how many lanes of a 64-beam task are still marching beyond 32 / 64 / 128 samples,
  * run lengths of EQUAL consecutive table values along a long ray — what a "speculate the next samples assuming the step
    stays the same" scheme could accept per memory round trip (round 4: mean run 1.58 -> not worth building),
  * the d^2 histogram of those samples (the rays creep 1-3 cells per step along jagged walls).
usage: python tools/debug/ray_runlen.py   (about 10 s on 8 cores; output kept in profiles/r04_ray_runlen.txt)"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from _util import oracle_map_dt, bench_start_poses
from oracle import orc
dt, res, origin = oracle_map_dt("example_map")
E, A, T = 512, 2, 320
poses = bench_start_poses(E, A)
rng = np.random.default_rng(1000)
sets = np.stack([np.stack([rng.uniform(-0.2, 0.2, E*A), rng.uniform(2.0, 6.0, E*A)], axis=1) for _ in range(T//20)])
sim = orc.SimOracle(E, A); sim.set_map_dt(dt, res, origin)
sim.set_noise(np.random.default_rng(12345).normal(0., .01, size=(T+2,1080)))
sim.reset(poses)
t0=time.time(); sim.rollout(sets, T, 20, poses, True, 8); print("rollout", time.time()-t0)
st = sim.state.copy()
# vectorised sphere trace of all rays, reference arithmetic (laser_models.py:106-186), identity rotation
theta_dis=2000; fov=4.7; B=1080
sines=np.sin(np.linspace(0,2*np.pi,theta_dis)); cosines=np.cos(np.linspace(0,2*np.pi,theta_dis))
inc = theta_dis*(fov/(B-1))/(2*np.pi)
N=E*A
xs=np.repeat(st[:,0],B); ys=np.repeat(st[:,1],B)
ti = theta_dis*(st[:,4]-fov/2.)/(2*np.pi); ti=np.fmod(ti,theta_dis); ti[ti<0]+=theta_dis
idx=np.empty((N,B),dtype=np.int64)
cur=ti.copy()
for b in range(B):
    idx[:,b]=cur.astype(np.int64)
    cur=cur+inc; cur[cur>=theta_dis]-=theta_dis
idx=idx.reshape(-1); c=cosines[idx]; s=sines[idx]
H,W=dt.shape
def lookup(x,y):
    xt=x-origin[0]; yt=y-origin[1]
    oob=(xt<0)|(xt>=W*res)|(yt<0)|(yt>=H*res)
    cc=(xt/res).astype(np.int64); rr=(yt/res).astype(np.int64)
    cc[oob]=-1; rr[oob]=-1
    return dt[rr,cc]
x=xs.copy(); y=ys.copy()
d=lookup(x,y); total=d.copy(); n=np.ones(x.shape,dtype=np.int64)
active=(d>1e-4)&(total<=30.0)
seqs={}   # ray -> list of d   (recorded lazily once a ray passes 48 samples: we keep all d for active rays from step 48 on)
it=0
hist=[]
while active.any():
    ia=np.nonzero(active)[0]
    x[ia]+=d[ia]*c[ia]; y[ia]+=d[ia]*s[ia]
    dn=lookup(x[ia],y[ia]); d[ia]=dn; total[ia]+=dn; n[ia]+=1
    it+=1
    if it>=32:
        for r,v in zip(ia,dn): seqs.setdefault(r,[]).append(v/res)
    active[ia]=(dn>1e-4)&(total[ia]<=30.0)
print("rays", x.size, "mean lookups", n.mean(), "max", n.max(), "rays>64:", (n>64).sum(), ">128:", (n>128).sum(), ">256:", (n>256).sum())
# per 64-beam task: max lookups, and the tail profile: how many lanes still active at sample k
nt=n.reshape(N,B)
tasks=[nt[:,k:k+64] for k in range(0,B,64)]
tmax=np.concatenate([t.max(axis=1) for t in tasks])
print("tasks", tmax.size, "mean task max", tmax.mean(), "p99", np.percentile(tmax,99), "max", tmax.max())
# in tasks with max>64: number of lanes active beyond sample 32/64
for thr in (32,64,128):
    cnt=np.concatenate([(t>thr).sum(axis=1) for t in tasks]); sel=cnt>0
    print("thr",thr,"tasks with any lane beyond:",sel.sum()," lanes beyond (mean over those):",cnt[sel].mean(), "hist 1/2-4/5-16/>16:", (cnt[sel]==1).sum(), ((cnt[sel]>=2)&(cnt[sel]<=4)).sum(), ((cnt[sel]>=5)&(cnt[sel]<=16)).sum(), (cnt[sel]>16).sum())
# run lengths of equal consecutive d for long rays (from sample 32 on)
runs=[]; tot=0
for r,v in seqs.items():
    if len(v)<32: continue
    v=np.array(v); tot+=len(v)
    ch=np.nonzero(np.diff(v)!=0)[0]
    rl=np.diff(np.concatenate([[-1],ch,[len(v)-1]]))
    runs.extend(rl.tolist())
runs=np.array(runs)
print("long rays (>=64 samples):", sum(1 for v in seqs.values() if len(v)>=32), "samples", tot, "runs", runs.size, "mean run", runs.mean(), "median", np.median(runs), "p90", np.percentile(runs,90))
# sample-weighted: expected accepted prefix if we speculate constant d with up to K lanes
for K in (4,8,16,32,64):
    # a run of length L is consumed in ceil(L/K) rounds... first sample of run is a 'normal' step
    rounds=np.ceil(runs/ K).sum()
    print("K",K,"samples per round", runs.sum()/rounds)
vals=np.concatenate([np.array(v) for v in seqs.values() if len(v)>=32])
u,cn=np.unique(np.round(vals**2).astype(int),return_counts=True)
print("d^2 histogram (cells^2):", dict(zip(u[:12],cn[:12])))
