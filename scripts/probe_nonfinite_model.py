import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ.setdefault("LANCE_HIP_DOT_FLOW_SKEW","1e18")
import numpy as np
import oracle as orc
import fuzz_dot_flow as fz
f32=np.float32
seed, case = 7001, int(sys.argv[1]) if len(sys.argv)>1 else 89
rng = np.random.default_rng([seed, case]); cfg = fz.draw(rng)
d,m,nlist,n = cfg["d"],cfg["m"],cfg["nlist"],cfg["n"]
x = fz.make(rng,cfg,n,d); q = fz.make(rng,cfg,cfg["nq"],d)
if cfg["dup"]: x[n//2:] = x[rng.integers(0,200,n-n//2)]
if cfg["zero_q"]: q[::17]=0.0; q[5::29]*=-3.0
x=x.astype(np.float16); q=q.astype(np.float16)
cent,_,_,_ = orc.kmeans_train(x[:nlist*64], nlist, max_iters=3, seed=case, metric="dot")
cb,_ = orc.pq_train(x[:2560], m, max_iters=2, seed=case+1)
cf=np.asarray(cent,f32); print("centroid rows non-finite:", np.nonzero(~np.isfinite(cf).all(1))[0], "codebook non-finite words:", int((~np.isfinite(np.asarray(cb,f32))).any(-1).sum()))
print("cent nonfinite kinds: +inf", int(np.isposinf(cf).sum()), "-inf", int(np.isneginf(cf).sum()), "nan", int(np.isnan(cf).sum()))
op, od = orc.find_partitions(q[:4], cent, cfg["nprobes"], "dot")
print("oracle probes q0:", op[0], od[0])
if len(sys.argv)>2:
    from lance_amd.engine import Engine
    eng=Engine()
    gp, gd = eng.find_partitions(q[:4], cent, cfg["nprobes"], "dot")
    gp=gp.cpu().numpy(); gd=gd.cpu().numpy()
    print("device probes q0:", gp[0], gd[0])
    print("bits oracle", [hex(v) for v in od[0].view(np.uint32)], "device", [hex(v) for v in gd[0].view(np.uint32)])
