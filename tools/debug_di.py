import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import numpy as np
import lqrrt_amd, coracle
from lqrrt_amd.engine import Engine
s=lqrrt_amd.systems.DoubleIntegrator(n_boxes=3000, seed=0)
N=int(sys.argv[1]) if len(sys.argv)>1 else 200
wave=int(sys.argv[2]) if len(sys.argv)>2 else 64
eng=Engine(s,capacity=N+wave+8,max_wave=wave)
kw=s.plan_kwargs
eng.set_resolution(kw['dt'],kw['FPR'],int(kw['horizon']/kw['dt']),np.abs(s.error_tol),s.goal,np.abs(s.goal_buffer))
space=np.array(s.sample_space,dtype=np.float64)
eng.set_sampler(np.mean(space,axis=1),np.diff(space).flatten(),np.array(s.goal_bias,dtype=np.float64),10)
st=np.random.RandomState(1).get_state(); eng.set_mt19937(st[1],st[2]); eng.tree_reset(s.x0)
stats=eng.extend(wave,node_limit=N)
o=coracle.make(s,N+wave+8,seed=1); o.enable_trace(100000); o.extend(max_nodes=N)
print('attempts',stats.attempts,o.iterations,'cand',stats.candidates,o.candidates,'size',eng.size,o.size)
pe,po=eng.parents(),o.parents(); k=min(len(pe),len(po))
d=np.flatnonzero(pe[:k]!=po[:k]); print('first parent diff',d[:5])
se,so=eng.states(),o.states(); dd=np.abs(se[:k]-so[:k]).max(axis=1); b=np.flatnonzero(dd>0); print('first state diff',b[:5], dd[b[:5]])
print('elen',eng.edge_lengths()[:12],o.edge_lengths()[:12])
Ke,Ko=eng.gains(),o.gains(); print('K diff',np.abs(Ke[:k]-Ko[:k]).max())
# ops
rng=np.random.RandomState(0); x=rng.uniform(0,100,(64,12)); u=rng.uniform(-1,1,(64,6))
eng2=s._engine(0.1)
xn=eng2.dynamics_batch(x,u); xo=np.array([o.dynamics(a,b) for a,b in zip(x,u)]); print('dyn diff',np.abs(xn-xo).max())
ok=eng2.feasible_batch(x,u); oo=np.array([o.feasible(a,b) for a,b in zip(x,u)]); print('feas diff',(ok!=oo).sum())
i0,c0=eng.nn_argmin(x,use_ignore=False); print('nn',[int(v) for v in i0[:8]],[o.nearest(a,S=s.S) for a in x[:8]])
stt=eng.states()
for qi in (3,):
    q=x[qi]; d=q-stt; c=np.sum(np.tensordot(d,s.S,axes=1)*d,axis=1)
    cd=eng.costs_to_go(q)
    print('numpy argmin',int(np.argmin(c)),'dev costs argmin',int(np.argmin(cd)),'max rel diff',np.max(np.abs(cd-c)/c))
    print('c[29],c[139] numpy',c[29],c[139],'dev',cd[29],cd[139])
    cI=np.sum(d*d,axis=1); print('identity costs',cI[29],cI[139])
ctr=np.zeros((200,12)); ctr[:,:3]=0.5*(s.obs[:200,:3]+s.obs[:200,3:]); 
ok=eng2.feasible_batch(ctr,np.zeros((200,6))); oo=np.array([o.feasible(a,np.zeros(6)) for a in ctr])
print('box centres feasible dev',ok.sum(),'oracle',oo.sum())
par=np.array([o.parents()[7]],dtype=np.int32)
print('oracle node7 parent',par, 'dev parent', eng.parents()[7])
