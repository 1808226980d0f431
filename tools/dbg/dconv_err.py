import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, torch.nn.functional as F
from hiddenpose_amd import hip_ops as ops
torch.manual_seed(0)
for (cin, cout, dims, bias) in [(1, 4, (1, 8, 16, 32), True), (4, 4, (1, 8, 16, 64), True), (4, 4, (1, 8, 16, 64), False)]:
    B, D, H, W = dims
    x = torch.randn(B, cin, D, H, W); w = torch.randn(cout, cin, 3, 3, 3) * .2; b = torch.randn(cout) if bias else None
    ref = F.conv3d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    y = ops._DConv3.apply(x.cuda(), w.cuda(), None if b is None else b.cuda(), False).cpu().double()
    e = (y - ref)
    print(cin, cout, dims, bias, "rel", float(e.norm() / ref.norm()))
    print(" per channel:", [round(float(e[:, c].norm() / ref[:, c].norm()), 4) for c in range(cout)])
    print(" per z:", [round(float(e[:, :, z].norm() / ref[:, :, z].norm()), 4) for z in range(D)])
    print(" per y (mod 8):", [round(float(e[:, :, :, yy::8].norm() / ref[:, :, :, yy::8].norm()), 4) for yy in range(8)])
    d = (y - ref)[0, :, 0, 0, :4]
    print(" diff sample ch x:", d.numpy().round(3).tolist(), " bias:", None if b is None else b.numpy().round(3).tolist())
