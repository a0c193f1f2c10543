import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
sys.argv=['bench']
import bench
boat,eng=bench.build_problem(10000,1024,0)
eng.extend(1024,until_size=9500)
os.environ['LQRRT_TRACE']='1'
st=eng.extend(1024,max_attempts=1024)
print(st.as_dict())
