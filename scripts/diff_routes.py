"""Where do the fused transform and the round-5 three-kernel route disagree at the C2 shape, and which one agrees with the oracle?"""
import os, subprocess, sys
import numpy as np
import torch
sys.path.insert(0, ".")
if os.environ.get("ROUTE_CHILD"):
    import lance_amd
    from lance_amd.testing import sift_like
    eng = lance_amd.default_engine()
    x = sift_like(1_000_000, 128, 1234, device="cuda")
    d = np.load("/tmp/route_model.npz")
    part, codes, _ = eng.ivfpq_encode(x, d["cent"], d["cb"], "l2")
    np.savez(os.environ["ROUTE_CHILD"], part=part.cpu().numpy(), codes=codes.cpu().numpy())
    sys.exit(0)
import lance_amd
from lance_amd.testing import sift_like
import oracle
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
g = torch.Generator(device="cuda").manual_seed(1)
cent = x[torch.randperm(1_000_000, device="cuda", generator=g)[:256]].contiguous()
part, _ = eng.assign(x[:200000], cent, "l2")
res = x[:200000] - cent[part.long()]
cb = torch.stack([res[torch.randperm(200000, device="cuda", generator=g)[:256]][:, i * 8:(i + 1) * 8] for i in range(16)]).contiguous()
np.savez("/tmp/route_model.npz", cent=cent.cpu().numpy(), cb=cb.cpu().numpy())
out = {}
for name, env in (("fused", {}), ("r05", {"LANCE_HIP_NO_XFORM_FUSED": "1"})):
    e = dict(os.environ); e.update(env); e["ROUTE_CHILD"] = f"/tmp/route_{name}.npz"
    subprocess.check_call([sys.executable, __file__], env=e)
    out[name] = np.load(f"/tmp/route_{name}.npz")
pf, pr = out["fused"]["part"], out["r05"]["part"]
cf, cr = out["fused"]["codes"], out["r05"]["codes"]
print("part ids differ:", int((pf != pr).sum()))
rows, ms = np.nonzero(cf != cr)
print("codes differ at", len(rows), "items; rows", rows[:20], "m", ms[:20])
ur = np.unique(rows)[:2000]
xs = x[torch.from_numpy(ur).cuda()].cpu().numpy()
centn, cbn = cent.cpu().numpy(), cb.cpu().numpy()
op, _ = oracle.assign(xs, centn, "l2")
ores = oracle.residual(xs, centn, op)
oc = oracle.pq_encode(ores, cbn, "l2")
print("oracle part == fused:", bool((op == pf[ur].view(np.uint32)).all()))
print("rows where FUSED codes != oracle:", int((cf[ur] != oc).any(axis=1).sum()), " rows where R05 codes != oracle:", int((cr[ur] != oc).any(axis=1).sum()), "of", len(ur))
for r, m in list(zip(rows, ms))[:8]:
    i = int(np.nonzero(ur == r)[0][0])
    sub = ores[i, m * 8:(m + 1) * 8]
    dd = ((cbn[m].astype(np.float64) - sub.astype(np.float64)) ** 2).sum(1)
    o = np.argsort(dd)[:3]
    print("row", r, "m", m, "fused", cf[r, m], "r05", cr[r, m], "oracle", oc[i, m], "f64 best", o, dd[o])
