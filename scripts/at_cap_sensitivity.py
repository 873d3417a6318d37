"""Oracle alone (CPU): the status differences a closed-loop fuzz run recorded (scripts/fuzz_closed_loop.py: `failures` / `at_cap`) are re-created on the oracle's own
loop, and the solve of the recorded step is repeated 40 times with the warm start perturbed by 1e-7: the statuses and iteration counts it ends with show whether the
solve sits AT the iteration cap, where a rounding error decides between 0, 2 and 4.   usage: python scripts/at_cap_sensitivity.py <fuzz json> <its seed>   -> profiles/r05_at_cap_sensitivity.json"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT,"tests")]
import numpy as np
from oracle import oracle as orc
from helpers import OracleLoop, random_batch
d=json.load(open(sys.argv[1]))
rng=np.random.default_rng(int(sys.argv[2]))
want={f["seed"]:f for f in d["failures"]+d.get("at_cap",[])}
records=[]
found=0
while found<len(want):
    N=int(rng.choice([3,9,10,15,20,21,31,40,50])); no=int(rng.integers(1,11)); B=int(rng.choice([1,3,8,21,40]))
    if N>31: B=min(B,12)
    alias=bool(rng.random()>0.4); lps=int(rng.choice([0,1])); lanes=int(rng.choice([0,0,16,21,32,64])); K=6
    seed=int(rng.integers(1<<30)); v=rng.uniform(-0.5,1.5,B)
    if seed not in want: continue
    x0,goal,obst=random_batch(B,no,seed=seed); x0[:,3]=v
    f=want[seed]; found+=1
    noise=np.random.default_rng(seed).standard_normal((K,B,no,2))
    cfg=orc.config(N,no,0.1*N)
    b=f["inst"]
    L=OracleLoop(orc,cfg,x0[b],goal[b],obst[b],alias=alias)
    for k in range(f["step"]): L.step(noise[k,b])
    P=orc.predict_params(cfg,L.obst)
    out=[]
    pr=np.random.default_rng(1)
    for t in range(40):
        X=L.X+ (0 if t==0 else 1e-7*pr.standard_normal(L.X.shape)); U=L.U+(0 if t==0 else 1e-7*pr.standard_normal(L.U.shape))
        r=orc.rti_solve(cfg,L.x,P,L.goal,X,U)
        out.append((r["status"],r["iters"]))
    from collections import Counter
    c=Counter(out).most_common()
    print(seed, f.get("why"), c)
    records.append(dict(configuration={k:f[k] for k in ("N","n_obst","B","alias","seed","step","inst")}, recorded=f.get("why") or f"status {f['status_gpu']} vs {f['status_oracle']}",
                        oracle_outcomes_under_1e7_perturbation=[dict(status=int(a),iters=int(b),count=n) for (a,b),n in c], cap=int(cfg.qp_iter_max)))
json.dump(dict(source=os.path.basename(sys.argv[1]), perturbation="warm start X, U + 1e-7 * N(0,1), 39 draws + the unperturbed solve", records=records),
          open(os.path.join(ROOT,"profiles","r05_at_cap_sensitivity.json"),"w"), indent=1)
