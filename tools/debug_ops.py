import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import numpy as np
import lqrrt_amd as lqrrt
for name in ['boat_advanced','car','boat_novice','pendulum']:
    g=np.load(os.path.join(ROOT,'tests/golden/ops_%s.npz'%name))
    s=lqrrt.systems.SYSTEMS[name](0)
    eng=s._engine(float(g['dt']))
    e=eng.erf_batch(g['erf_xg'],g['erf_x']); print(name,'erf max err',np.abs(e-g['erf_e']).max())
    K=eng.gain_batch(g['lqr_x']); print(name,'K max err',np.abs(K-g['lqr_K']).max())
    xn=eng.dynamics_batch(g['dyn_x'],g['dyn_u']); d=np.abs(xn-g['dyn_xnext']); print(name,'dyn max err',d.max(axis=0))
    i=np.unravel_index(np.argmax(d),d.shape); print('   worst',i,g['dyn_x'][i[0]],g['dyn_u'][i[0]],xn[i[0]],g['dyn_xnext'][i[0]])
