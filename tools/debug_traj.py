import sys, os, hashlib, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import numpy as np
import lqrrt_amd as lqrrt
TRAJ = [("boat_advanced", "200", 64), ("boat_advanced", "200", 1024), ("boat_intermediate", "300", 256),
        ("boat_novice", "300", 256), ("car", "500", 256), ("pendulum", "150", 64), ("car", "2000", 1024),
        ("car", "firstgoal", 128), ("boat_novice", "firstgoal", 128), ("boat_advanced", "3000", 1024), ("boat_advanced","10k",1024)]
for name,tag,wave in TRAJ:
    path=os.path.join(ROOT,'tests/golden/traj_%s_%s.npz'%(name,tag))
    if not os.path.exists(path): continue
    g=np.load(path)
    s=lqrrt.systems.SYSTEMS[name](0)
    cons=lqrrt.Constraints(s.nstates,s.ncontrols,s.goal_buffer,s.is_feasible)
    mt=float(g['min_time'])
    p=lqrrt.Planner(s.dynamics,s.lqr,cons,error_tol=s.error_tol,erf=s.erf,min_time=mt,max_time=mt+1,max_nodes=int(g['max_nodes']),goal0=s.goal,sys_time=lambda:0.0,printing=False,wave_size=wave,**s.plan_kwargs)
    np.random.seed(1)
    t0=time.time()
    ret=p.update_plan(s.x0,s.sample_space,goal_bias=s.goal_bias,xrand_gen=10)
    dt=time.time()-t0
    pid=np.array(p.tree.pID,dtype=np.int32)
    k=min(len(pid),len(g['pID']))
    same=np.array_equal(pid,g['pID'])
    first=-1 if same else int(np.argmax(pid[:k]!=g['pID'][:k])) if (pid[:k]!=g['pID'][:k]).any() else k
    st=p.tree.state; d=np.abs(st[:k]-g['state'][:k])
    print('%-18s %-9s W=%-5d wall %.2fs  attempts %d/%d cand %d/%d nodes %d/%d pid_same %s first_diff %d  state maxerr %.3g (median node err %.3g, nodes>1e-9: %d)  hits %d rounds %d resteers %d waves %d'%(
        name,tag,wave,dt,p.stats['attempts'],int(g['iterations']),p.stats['candidates'],int(g['n_candidates']),len(pid),len(g['pID']),same,first,d.max(),np.median(d.max(axis=1)),int((d.max(axis=1)>1e-9).sum()),p.stats['goal_hits'],p.stats['fix_rounds'],p.stats['resteers'],p.stats['waves']))
    sys.stdout.flush()
