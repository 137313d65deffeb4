"""Would a 16-entry-per-sub-quantiser lower-bound table prune the ADC scan?  LB(row) = sum_m min_{c' in group(code_m)} LUT[m][c']
with groups = high nibble of the code (natural codebook order) or of a permuted code (codewords grouped by similarity)."""
import sys
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(10000, 128, 4321, device="cuda")
idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
cent = idx._ix.centroids; cb = idx._ix.codebook            # [256,128], [16,256,8]
nprobes, keff, NQ = 10, 100, 256
probes, _ = eng.find_partitions(q[:NQ], cent, nprobes)
ids, dd = idx.search_device(q[:NQ], keff, nprobes, 0)
T = dd[:, keff - 1]
part = idx.part_ids.long(); codes = idx.codes.long()        # [n], [n,16]

def bisect_groups(c):                                        # c: [256, sd] -> perm (new position -> old code), 16 balanced groups of 16
    order = torch.arange(256, device=c.device)[None]         # list of index groups
    groups = [torch.arange(256, device=c.device)]
    for _ in range(4):
        nxt = []
        for g in groups:
            v = c[g] - c[g].mean(0)
            _, _, vh = torch.linalg.svd(v, full_matrices=False)
            proj = v @ vh[0]
            o = torch.argsort(proj)
            nxt += [g[o[: len(g) // 2]], g[o[len(g) // 2:]]]
        groups = nxt
    return torch.cat(groups)                                 # position i (group i // 16) holds old code perm[i]

perms = [bisect_groups(cb[m]) for m in range(16)]
inv = [torch.empty(256, dtype=torch.long, device="cuda").scatter_(0, p, torch.arange(256, device="cuda")) for p in perms]
tot = pr_nat = pr_grp = 0
tight_nat = tight_grp = 0.0
for qi in range(NQ):
    for p in probes[qi].tolist():
        rows = torch.nonzero(part == p).reshape(-1)
        r = (q[qi] - cent[p]).reshape(16, 1, 8)
        lut = ((r - cb) ** 2).sum(-1)                        # [16,256]
        c = codes[rows]                                      # [np,16]
        d = lut.gather(1, c.t()).sum(0)
        lb_nat_tab = lut.reshape(16, 16, 16).min(-1).values  # [16 m][16 groups]
        lb_nat = lb_nat_tab.gather(1, (c.t() >> 4)).sum(0)
        lutp = torch.stack([lut[m][perms[m]] for m in range(16)])
        lb_grp_tab = lutp.reshape(16, 16, 16).min(-1).values
        newc = torch.stack([inv[m][c[:, m]] for m in range(16)])          # [16, np]
        lb_grp = lb_grp_tab.gather(1, newc >> 4).sum(0)
        tot += len(rows); pr_nat += (lb_nat > T[qi]).sum().item(); pr_grp += (lb_grp > T[qi]).sum().item()
        tight_nat += (lb_nat / d).sum().item(); tight_grp += (lb_grp / d).sum().item()
print(f"rows {tot}: pruned by natural hi-nibble bound {pr_nat/tot:.3f} (LB/d = {tight_nat/tot:.2f}); by grouped codewords {pr_grp/tot:.3f} (LB/d = {tight_grp/tot:.2f})")
