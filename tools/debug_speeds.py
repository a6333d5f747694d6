import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'mrs-gym_amd'), os.path.join(ROOT,'tests')]
import numpy as np, torch, oracle, mrsgym_amd
from util_scenarios import *
E,N=1,64
pos,eul=grid_spawn(E,N); z=np.zeros((E,N,3),np.float32)
sh=mrsgym_amd.SwarmShard(E,N,"cuda:0",want_rpm=True); sh.set_state(pos=pos,ori=eul,vel=z,angvel=z)
sw=oracle.OracleSwarm(E,N); sw.set_state(pos=pos.astype(np.float64),euler=eul,vel=z.astype(np.float64),angvel=z.astype(np.float64))
acts=ActionStream("set_speeds",E,N,pos,seed=11)
a=acts(0); sh.step(torch.from_numpy(a).cuda(),"set_speeds"); sw.step(a,"set_speeds")
g=sh.view(sh.angvel).cpu().numpy()
d=g-sw.angvel
i=np.abs(d).max(-1).argmax()
print("worst agent",i,"gpu",g[0,i],"orc",sw.angvel[0,i],"diff",d[0,i])
print("wrench orc",sw.wrench[0,i])
I=np.array(list(sw.p.inertia))
print("implied torque diff (body approx)", d[0,i]*I/0.01)
f32=np.float32
s=a[0,i]; sq=s*s; t=sq*f32(7.94e-12); zt=((-t[0]+t[1])-t[2])+t[3]
print("numpy zt",repr(zt), "F", (sq*f32(3.16e-10)).astype(np.float64))
print("vel diff", (sh.view(sh.vel).cpu().numpy()-sw.vel)[0,i])
