import os, sys, numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import spectralkernels_jl_amd as sk
from oracle import sk_oracle as o
prm=(0.7,1.3,1.8)
rs=10.0**np.linspace(-6,0.1,100000)
res={}
for lv in (0,1):
    if lv: os.environ["SK_HK_LEVELS"]="1"
    else: os.environ.pop("SK_HK_LEVELS",None)
    cfg=sk.AdaptiveKernelConfig(sk.SDF.matern(*prm,2),dim=2,tol=1e-8,quadspec=(8192,16))
    k0=sk.compute_k0(cfg)
    ks,es=sk.kernel_values(cfg,rs,k0=k0)
    res[lv]=ks
    print("levels" if lv else "boxes", cfg.last_stats()["n_panels"], np.abs(ks-o.matern_cov(rs,prm,d=2)).max()/k0, flush=True)
print("diff", np.abs(res[0]-res[1]).max()/k0)
