import sys, os, json
ROOT='/root/repo'; sys.path[:0]=[ROOT, ROOT+'/dynamic-obstacle-avoidance-mpc_amd', ROOT+'/tests']
import numpy as np, mpc_gpu
from test_gpu_replay import replay, row_match, TABLES
out={}
for lps in (0,1):
    mpc_gpu.BatchedMpc.default_lanes_per_stage=lps
    for stem,t in TABLES.items():
        interp=bool(t['spec'].get('interpolate_init'))
        tb,rows,scen=replay(mpc_gpu, stem, **(dict(interpolate_init=True, bug_compat_alias=False) if interp else {}))
        out[f"{stem}_{lps}"]=(int(row_match(tb,rows,1e-3).sum()), int(row_match(tb,rows,1e-6).sum()), round(float(tb[:,0].mean()),2), round(float(tb[:,1].mean()),2), round(float(tb[:,4].mean()),1))
        print(stem,lps,out[f"{stem}_{lps}"], 'recorded', t['hit'], t['reached'], t['mean_iters'], flush=True)
