import sys, os, subprocess, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
variant=sys.argv[1] if len(sys.argv)>1 else ''
so='/tmp/liblqrrt_%s.so'%(variant or 'base')
cmd=['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-std=c++17','-ffp-contract=off','-fPIC','-shared',os.path.join(ROOT,'lqrrt_amd/csrc/engine.hip'),'-o',so]+(['-D'+v for v in variant.split(',') if v])
subprocess.check_call(cmd)
import lqrrt_amd._native as nat
nat.LIB_PATH=so
import lqrrt_amd
from lqrrt_amd.engine import Engine
s=lqrrt_amd.systems.BoatAdvanced(0)
eng=Engine(s,capacity=12000,max_wave=1024)
kw=s.plan_kwargs
eng.set_resolution(kw['dt'],kw['FPR'],20,np.abs(s.error_tol),s.goal,np.abs(s.goal_buffer))
space=np.array(s.sample_space,dtype=np.float64)
eng.set_sampler(np.mean(space,axis=1),np.diff(space).flatten(),np.array(s.goal_bias,dtype=np.float64),10)
st=np.random.RandomState(1).get_state(); eng.set_mt19937(st[1],st[2]); eng.tree_reset(s.x0)
eng.extend(1024,until_size=3000)
rng=np.random.RandomState(0)
xs=space[:,0]+(space[:,1]-space[:,0])*rng.random_sample((1024,6))
ids,_=eng.nn_argmin(xs)
for cnt in (1,16,1024):
    eng.profile_enable(True)
    for _ in range(20): ln=eng.steer_batch(ids[:cnt],xs[:cnt])[0]
    pr=eng.profile_read()
    print(variant or 'base','problems',cnt,'steer avg us %.1f'%(1e3*pr['steer_ms']/pr['steer_launches']),'mean len %.1f'%ln.mean())
