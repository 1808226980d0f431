"""One launch set of the thin-channel 3^3 kernels at a full-resolution U-Net shape, for rocprofv3 --pmc runs:
    rocprofv3 --pmc <counters> --kernel-trace -d gpurun_out/pmc_x -o d --output-format csv -- python3 tools/dconv_pmc_probe.py
    python tools/pmc_summary.py gpurun_out/pmc_x/*/d_counter_collection.csv dconv"""
import sys; sys.path.insert(0, '.')
import torch
from hiddenpose_amd import _lib
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
for (cin, cout) in [(4, 4), (8, 4), (1, 1)]:
    B, D, H, W = 1, 256, 256, 256
    x = torch.randn(B, cin, D, H, W, device='cuda'); g = torch.randn(B, cout, D, H, W, device='cuda'); w = torch.randn(cout, cin, 3, 3, 3, device='cuda')
    y = torch.empty_like(g); gx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(cout, device='cuda')
    ws = torch.empty(int(L.hp_dconv3_backward_data_workspace_bytes(B, cin, D, H, W, 0)) // 4 + 1, device='cuda')
    wsw = torch.empty(int(L.hp_dconv3_backward_weight_workspace_bytes(B, cin, cout, D, H, W)) // 4, device='cuda')
    for _ in range(3):
        L.hp_dconv3_forward(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, cin, cout, D, H, W, 0, st)
        L.hp_dconv3_backward_data(g.data_ptr(), w.data_ptr(), gx.data_ptr(), B, cin, cout, D, H, W, 0, ws.data_ptr(), st)
        L.hp_dconv3_backward_weight(x.data_ptr(), g.data_ptr(), dw.data_ptr(), db.data_ptr(), B, cin, cout, D, H, W, 0, wsw.data_ptr(), st)
    torch.cuda.synchronize()
