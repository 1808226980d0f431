import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, torch.nn.functional as F
from hiddenpose_amd import hip_ops as ops
torch.manual_seed(0)
for (cin, cout, dims) in [(4, 4, (1, 4, 8, 16)), (4, 4, (1, 4, 8, 64)), (8, 4, (1, 8, 8, 16))]:
    B, D, H, W = dims
    grid = lambda t: t.bfloat16().float()
    x = grid(torch.randn(B, cin, D, H, W)); w = grid(torch.randn(cout, cin, 3, 3, 3) * .2); gy = grid(torch.randn(B, cout, D, H, W))
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (F.conv3d(xd, wd, padding=1) * gy.double()).sum().backward()
    res = {}
    for m in ("fp32", "bf16"):
        p = ops.set_dconv_precision(m)
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        (ops._DConv3.apply(xg, wg, None, False) * gy.cuda()).sum().backward()
        ops.set_dconv_precision(p)
        res[m] = wg.grad.cpu().double()
    ref = wd.grad
    print(cin, cout, dims, "fp32 err %.2e  bf16 err %.2e" % ((res["fp32"] - ref).norm() / ref.norm(), (res["bf16"] - ref).norm() / ref.norm()))
    e = (res["bf16"] - ref)
    print(" per tap rel err:", [round(float(e[:, :, t // 9, (t // 3) % 3, t % 3].norm() / ref[:, :, t // 9, (t // 3) % 3, t % 3].norm()), 3) for t in range(27)])
    print(" per (co,ci) rel err:\n", (e.flatten(2).norm(dim=2) / ref.flatten(2).norm(dim=2)).numpy().round(3))
    print(" ratio sample dz=1,dy=1:", (res["bf16"][0, 0, 1, 1] / ref[0, 0, 1, 1]).numpy().round(3))
