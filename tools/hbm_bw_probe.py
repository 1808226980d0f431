import torch, time
x = torch.empty(1<<30, device='cuda')  # 4 GiB
y = torch.empty(1<<30, device='cuda')
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
b = x.numel()*4
print("fill   %.2f TB/s" % (b/t(lambda: x.fill_(1.0))/1e12))
print("copy   %.2f TB/s (r+w)" % (2*b/t(lambda: y.copy_(x))/1e12))
print("sum    %.2f TB/s" % (b/t(lambda: x.sum())/1e12))
z = torch.empty(1<<28, device='cuda')
print("1r->4w %.2f TB/s" % (5*z.numel()*4/t(lambda: torch.add(x.view(4,-1), z, out=y.view(4,-1)))/1e12))
