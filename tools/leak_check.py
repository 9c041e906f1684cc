"""Create / use / destroy handles in a loop; device memory in use must return to where it started."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navierstokes_amd import mpk, synth, dist as D
torch.cuda.init()
def used(): torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
p, c, v = synth.rows("s15", 400_000)
pf, cf, vf = synth.fe_matrix(12)
x = torch.from_numpy(synth.x_sin(0, 400_000)).cuda(); y = torch.empty(400_000, dtype=torch.float64, device="cuda")
xf = torch.from_numpy(synth.x_sin(0, len(pf) - 1)).cuda(); yf = torch.empty(len(pf) - 1, dtype=torch.float64, device="cuda")
def once():
    A = mpk.csrmatrix(400_000, p, c, v); mpk.SpMV_CSR(y, x, A); mpk.SpMkV([y.clone(), y.clone()], x, A)
    yh = np.empty(400_000); mpk.SpMV_CSR(yh, synth.x_sin(0, 400_000), A); A.close() if hasattr(A, "close") else None
    B = mpk.csrmatrix(len(pf) - 1, pf, cf, vf); mpk.SpMV_CSR(yf, xf, B); del A, B
    dc = D.DistCSR(np.array([0, 400_000], np.int64), p, c, v); xe = dc.new_x_ext(); yy = dc.new_y(); dc.spmv(xe, yy); dc.close(); del dc
    # round 4: a one-process multi-rank handle (two ranks on this card, event exchange), sliced copies of both kinds, a relabelled twin
    M = mpk.DistMatrix(2, 400_000, p, c, v); hy = np.empty(400_000); M.spmv(hy, synth.x_sin(0, 400_000)); M.close()
    os.environ["MI355_BCSR_SELL"] = "1"; os.environ["MI355_REORDER"] = "1"
    pp, cc, vv, _ = synth.permute_nodes(pf, cf, vf, block=4, seed=3)
    C = mpk.csrmatrix(len(pf) - 1, pp, cc, vv); mpk.SpMV_CSR(yf, xf, C); C.close()
    del os.environ["MI355_BCSR_SELL"], os.environ["MI355_REORDER"]
    # round 5: value updates (fused refills), a sliced copy rebuilt on request, a row-offset piece (shifted plan), the distributed orthogonalize
    A = mpk.csrmatrix(400_000, p, c, v); A.update_values(torch.from_numpy(v).cuda()); A.set_kernel("ring"); A.set_kernel("sstream"); mpk.SpMV_CSR(y, x, A); A.close()
    E = mpk.csrmatrix(400_000, p, c, v, rowmap=(np.arange(400_000) + 3).astype(np.int32)).set_kernel("sstream")
    yo = torch.empty(400_003, dtype=torch.float64, device="cuda"); mpk.SpMV_CSR(yo, x, E); E.close()
    M = mpk.DistMatrix(2, 400_000, p, c, v); va, vb_, vc = M.vector(synth.x_sin(0, 400_000)), M.vector(synth.x_ones(400_000)), M.vector()
    M.orthogonalize_dev(va, vb_, vc, 1e-8, want_beta=False); M.orthogonalize_dev(va, vb_, vc, 1e-8); M.close()  # (the vectors outlive the handle here: mi_dist_destroy releases their memory)
    for t_ in (va, vb_, vc): t_.close()
    gc.collect()
once(); base = used()
for i in range(30): once()
torch.cuda.empty_cache()
print(f"LEAK base {base:.1f} MiB, after 30 more rounds {used():.1f} MiB")
