"""Times lance_hip_ivfpq_encode (the transform: assign + residual + PQ encode) at the north-star shapes.
usage: python scripts/probe_xform.py [c2|c4|c5|all] ; LANCE_HIP_NO_XFORM_FUSED=1 selects the round-5 three-kernel route."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like

eng = lance_amd.default_engine()
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
out = {"fused": os.environ.get("LANCE_HIP_NO_XFORM_FUSED") is None}


def run(name, x, nlist, m, reps=5):
    n, d = x.shape
    g = torch.Generator(device="cuda").manual_seed(1)
    sel = torch.randperm(n, device="cuda", generator=g)[:nlist]
    cent = x[sel].float().contiguous()
    if x.dtype == torch.float16:
        cent = cent.half()
    part, _ = eng.assign(x[:200000], cent, "l2")
    res = (x[:200000].float() - cent.float()[part.long()])
    cb = torch.stack([res[torch.randperm(200000, device="cuda", generator=g)[:256]][:, i * (d // m):(i + 1) * (d // m)] for i in range(m)]).contiguous()
    if x.dtype == torch.float16:
        cb = cb.half()
    for _ in range(2):
        eng.ivfpq_encode(x, cent, cb, "l2")
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        part, codes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    eng.timing(True)
    eng.ivfpq_encode(x, cent, cb, "l2")
    eng.synchronize()
    stages = {k: round(eng.timing_query(k)[0], 4) for k in ("xform_fused", "ma_sweep", "ma_recheck", "encode_fused", "assign")}
    eng.timing(False)
    es = x.element_size()
    out[name] = {"n": n, "d": d, "nlist": nlist, "m": m, "dtype": str(x.dtype), "ms_min": round(min(ts) * 1e3, 4), "ms_all": [round(t * 1e3, 4) for t in ts],
                 "stage_ms": stages, "hbm_frac_of_8TBs": round((n * d * es + n * (m + 8)) / min(ts) / 8e12, 4),
                 "codes_sum": int(codes.long().sum().item()), "part_sum": int(part.long().sum().item())}
    print(name, json.dumps(out[name]), flush=True)


if which in ("c2", "all"):
    run("c2_1Mx128_f32_ivf256_pq16", sift_like(1_000_000, 128, 1234, device="cuda"), 256, 16)
if which in ("c4", "all"):
    run("c4_1Mx128_f16_ivf4096_pq16", sift_like(1_000_000, 128, 77, device="cuda").half(), 4096, 16)
if which in ("c5", "all"):
    x8 = (sift_like(1_000_000, 128, 78, device="cuda") - 100).clamp(-128, 127).to(torch.int8)
    run("c5_1Mx128_i8_ivf4096_pq32", x8, 4096, 32)
    run("c5_1Mx128_i8_ivf65536_pq32", x8, 65536, 32, reps=2)
json.dump(out, open(os.environ.get("OUT", "/dev/stdout"), "w"), indent=1)
