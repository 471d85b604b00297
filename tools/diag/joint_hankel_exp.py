import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import spectralkernels_jl_amd as sk
from oracle import sk_oracle as oracle
tol=1e-8; prm=(1.0,1.3,0.8); f=sk.SDF.matern(*prm,2)
rng=np.random.default_rng(11)
lags=np.sqrt(rng.uniform(0,1.2,100000)**2+rng.uniform(0,1.2,100000)**2)
kw=dict(dim=2,tol=tol)
cfgs=[sk.AdaptiveKernelConfig(f,**kw),sk.AdaptiveKernelConfig(f.d(1),**kw),sk.AdaptiveKernelConfig(f.d(2),**kw),sk.AdaptiveKernelConfig(f,derivative=True,**kw)]
pds=[False,True,True,False]
k0=float(oracle.matern_cov(0.0,prm,d=2))
def timed(fn,reps=8):
    fn(); fn()
    t0=time.perf_counter()
    for _ in range(reps): out=fn()
    return (time.perf_counter()-t0)/reps,out
for name,sel in (("K+2 params",[0,1,2]),("all four",[0,1,2,3])):
    cs=[cfgs[i] for i in sel]; pd=[pds[i] for i in sel]
    os.environ.pop("SK_JOINT_HK",None)
    ts,(vs,_)=timed(lambda: sk.kernel_values_joint(cs,lags,k0,param_derivative=pd))
    os.environ["SK_JOINT_HK"]="1"
    tj,(vj,_)=timed(lambda: sk.kernel_values_joint(cs,lags,k0,param_derivative=pd))
    st=cs[0].last_stats()
    os.environ["SK_JOINT_TIMING"]="1"
    print(name,"separate %.3f ms, joint %.3f ms (panels %d, syncs %d), maxdiff %.2e"%(1e3*ts,1e3*tj,st["n_panels"],st["n_syncs"],max(np.abs(vj[t]-vs[t]).max()/max(k0,np.abs(vs[t]).max()) for t in range(len(sel)))))
