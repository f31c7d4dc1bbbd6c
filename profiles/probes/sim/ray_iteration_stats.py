"""Longest rays of a scan launch on the benchmark distribution (CPU, NumPy sphere tracer on the oracle map): iteration
counts of the worst rays and how many of their steps stay in the same cell.  A launch cannot end before its longest
ray: ~330 dependent lookups at 4 096 cars (profiles/r03_small_launch.txt)."""
import numpy as np, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from red_gym_amd import workload
B=4096
sc = oracle.Scanner(1080, 2*np.pi); sc.set_map(workload.EXAMPLE_MAP+'.yaml','.png')
noise = oracle.noise_table(12345, 400)
batch = oracle.Batch(sc, B, 1, workload.spawn_poses(B,1), noise=noise)
acts = workload.action_pool(8, B, 1)
m = oracle.load_map(workload.EXAMPLE_MAP+'.yaml','.png')
dt = m['dt']; H,W = dt.shape; res=m['resolution']; ox,oy=m['orig_x'],m['orig_y']
def iters(poses):
    n=len(poses); nb=1080
    ang = poses[:,2:3] + np.linspace(-np.pi, np.pi, nb)[None,:]
    c,s=np.cos(ang),np.sin(ang)
    x=np.repeat(poses[:,0:1],nb,1); y=np.repeat(poses[:,1:2],nb,1)
    def look(x,y):
        ci=np.floor((x-ox)/res).astype(int); ri=np.floor((y-oy)/res).astype(int)
        oob=(ci<0)|(ci>=W)|(ri<0)|(ri>=H)
        d=dt[np.clip(ri,0,H-1),np.clip(ci,0,W-1)]
        return np.where(oob, dt[-1,-1], d), ri*4096+ci
    d,cell=look(x,y); tot=d.copy(); it=np.zeros((n,nb),int); same=np.zeros((n,nb),int); act=(d>1e-4)&(tot<=30)
    while act.any():
        x=np.where(act,x+d*c,x); y=np.where(act,y+d*s,y)
        dn,cn=look(x,y); same+=act&(cn==cell); cell=np.where(act,cn,cell)
        d=np.where(act,dn,d); tot=np.where(act,tot+d,tot); it+=act
        act=act&(d>1e-4)&(tot<=30)
    return it,same
for k in range(81):
    batch.step(acts[k%8], threads=8)
    if k in (0,20,40,60,80):
        st=np.array([e.observe(False)['state'][0] for e in batch.envs])
        it,same=iters(st[:,[0,1,4]])
        flat=it.ravel(); order=np.argsort(flat)[::-1][:8]
        print('step',k,'top rays (iters/same-cell):',' '.join('%d/%d'%(flat[i],same.ravel()[i]) for i in order), '| rays>200: %d  >300: %d  >500: %d'%((flat>200).sum(),(flat>300).sum(),(flat>500).sum()))
