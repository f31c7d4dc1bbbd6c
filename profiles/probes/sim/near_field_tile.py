"""SURVEY H3(b) priced on the CPU: which share of a scan's table lookups falls inside a square tile of cells centred on the
car (what an LDS copy of the near field could serve)?  Benchmark distribution, NumPy sphere tracer on the oracle map."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from red_gym_amd import workload
B=512
sc = oracle.Scanner(1080, 2*np.pi); sc.set_map(workload.EXAMPLE_MAP+'.yaml','.png')
noise = oracle.noise_table(12345, 400)
batch = oracle.Batch(sc, B, 1, workload.spawn_poses(B,1), noise=noise)
acts = workload.action_pool(8, B, 1)
m = oracle.load_map(workload.EXAMPLE_MAP+'.yaml','.png')
dt = m['dt']; H,W = dt.shape; res=m['resolution']; ox,oy=m['orig_x'],m['orig_y']
for k in range(61): batch.step(acts[k%8], threads=8)
st=np.array([e.observe(False)['state'][0] for e in batch.envs]); poses=st[:,[0,1,4]]
n=len(poses); nb=1080
ang = poses[:,2:3] + np.linspace(-np.pi, np.pi, nb)[None,:]
c,s=np.cos(ang),np.sin(ang)
x=np.repeat(poses[:,0:1],nb,1); y=np.repeat(poses[:,1:2],nb,1)
c0=np.floor((poses[:,0:1]-ox)/res).astype(int); r0=np.floor((poses[:,1:2]-oy)/res).astype(int)
def look(x,y):
    ci=np.floor((x-ox)/res).astype(int); ri=np.floor((y-oy)/res).astype(int)
    oob=(ci<0)|(ci>=W)|(ri<0)|(ri>=H)
    d=dt[np.clip(ri,0,H-1),np.clip(ci,0,W-1)]
    return np.where(oob, dt[-1,-1], d), ci, ri
d,ci,ri=look(x,y); tot=d.copy(); act=(d>1e-4)&(tot<=30)
halves=[16,24,32,48,64]
cnt={h:0 for h in halves}; total=0; first_exit={h:np.full((n,nb),-1) for h in halves}; it=np.zeros((n,nb),int)
# beams fully served in-tile: all lookups of the beam within the tile
allin={h:np.ones((n,nb),bool) for h in halves}
while act.any():
    x=np.where(act,x+d*c,x); y=np.where(act,y+d*s,y)
    dn,ci,ri=look(x,y)
    total+=act.sum()
    for h in halves:
        inside=(np.abs(ci-c0)<h)&(np.abs(ri-r0)<h)
        cnt[h]+=(act&inside).sum()
        allin[h]&=~(act&~inside)
    d=np.where(act,dn,d); tot=np.where(act,tot+d,tot); it+=act
    act=act&(d>1e-4)&(tot<=30)
print('lookups per car (excl. first):', total/n)
for h in halves:
    print('tile %dx%d cells (%.1f m): %.1f %% of lookups inside; %.1f %% of beams never leave it; lookups of leaving beams: %.1f %% of all' % (2*h,2*h,2*h*res, 100*cnt[h]/total, 100*allin[h].mean(), 100*(it*(~allin[h])).sum()/total))
