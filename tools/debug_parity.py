import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import numpy as np
import lqrrt_amd as lqrrt
from systems_np import SYSTEMS, make_oracle_planner
name=sys.argv[1] if len(sys.argv)>1 else 'boat_advanced'
N=int(sys.argv[2]) if len(sys.argv)>2 else 200
wave=int(sys.argv[3]) if len(sys.argv)>3 else 64
s=lqrrt.systems.SYSTEMS[name](0)
cons=lqrrt.Constraints(s.nstates,s.ncontrols,s.goal_buffer,s.is_feasible)
p=lqrrt.Planner(s.dynamics,s.lqr,cons,error_tol=s.error_tol,erf=s.erf,min_time=2,max_time=3,max_nodes=N,goal0=s.goal,sys_time=lambda:0.0,printing=False,wave_size=wave,**s.plan_kwargs)
np.random.seed(1)
p.update_plan(s.x0,s.sample_space,goal_bias=s.goal_bias,xrand_gen=10)
rs=SYSTEMS[name](0)
ref=make_oracle_planner(rs,N,min_time=2,max_time=3)
np.random.seed(1)
ref.update_plan(rs.x0,rs.sample_space,goal_bias=rs.goal_bias,xrand_gen=10,trace=True)
print('stats',p.stats)
print('size',p.tree.size,ref.tree.size,'pid equal',list(p.tree.pID)==list(ref.tree.pID))
st=p.tree.state; d=np.abs(st-ref.tree.state[:len(st)])
bad=np.flatnonzero(d.max(axis=1)>1e-9)
print('bad nodes',len(bad),bad[:20])
el=p._engine.edge_lengths()
for ID in bad[:6]:
    par=p.tree.pID[ID]
    print('node',ID,'parent',par,'parent bad',par in bad,'elen',el[ID],len(ref.tree.x_seq[ID]),'err',d[ID])
    xs=np.array(p.tree.x_seq[ID]); rx=np.array(ref.tree.x_seq[ID])
    k=min(len(xs),len(rx)); e=np.abs(xs[:k]-rx[:k]).max(axis=1)
    print('  per-step err',np.array2string(e,precision=2))
    us=np.array(p.tree.u_seq[ID]); ru=np.array(ref.tree.u_seq[ID])
    print('  u err',np.array2string(np.abs(us[:k]-ru[:k]).max(axis=1),precision=2))
    j=int(np.argmax(e>1e-9))
    print('  first bad step',j,'x dev',xs[j],'x ref',rx[j]); print('  u dev',us[j],'u ref',ru[j])
    if j>0: print('  prev x',rx[j-1])
    else: print('  start x',ref.tree.state[par], 'K dev', p._engine.gains(par,1)[0].ravel()[:6])
