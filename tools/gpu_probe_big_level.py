import sys, os, time, ctypes as C, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.check_call(["make","-C",ROOT+"/rdis_amd/host"],stdout=subprocess.DEVNULL); subprocess.check_call(["make","-C",ROOT+"/tests/cpp"],stdout=subprocess.DEVNULL)
h=C.CDLL(ROOT+"/tests/cpp/libharness.so")
for (nc,npt) in [(64,20000),(128,100000)]:
    t=time.time(); pp=P.make_synthetic_ba(1,nc,npt,obs_per_pt=4); path="/tmp/big_%d_%d.txt"%(nc,npt); P.save_bal(pp,path); print("generated+saved %d factors %d vars in %.1fs"%(pp.nfac,pp.nvars,time.time()-t))
    out,tr=np.zeros(12),np.zeros((4096,8)); v=lambda a:a.ctypes.data_as(C.c_void_p)
    t=time.time(); rc=h.harness_level_driver(path.encode(),C.c_longlong(0),C.c_longlong(0),25,10,C.c_double(0.2),1,v(out),v(tr),C.c_longlong(4096),None); dt=time.time()-t
    print(rc,"%.6g -> %.6g in %d sweeps, %d launches; optimize %.1f ms (decomposition %.1f ms), total incl load %.1fs; nodes %d, largest separator %d, monotone %d"%(out[1],out[0],out[2],out[9],out[6],out[7],dt,out[3],out[8],out[10]))
    for r in tr[:4]: print("   sweep %d depth %d kind %d: %d comps, %d vars, %d factors -> %.6g (%.2f ms)"%tuple(r))
