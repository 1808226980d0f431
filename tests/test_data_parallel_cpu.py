"""World-size-2 `gloo` test of the data-parallel path (hiddenpose_amd/data_parallel.py): bucketed
asynchronous gradient all-reduce overlapped with backward must give every rank the average
gradient, identical parameters after the step, and an exact batch-global Dice term."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                               torch.nn.Linear(32, 5))


def _dice_rank_loss_and_grad(x, t, world, eps=1e-9):
    """The arithmetic of hip_ops._BceDiceGlobal / hp_bce_dice_{partial,finalize,backward_scaled} restated in torch:
    local sums, ONE all-reduce of the three Dice sums, loss = local BCE mean + 1 - Dice_global, analytic gradient with
    the Dice part scaled by the world size (gradients are averaged over ranks afterwards)."""
    s = torch.sigmoid(x)
    n = x.numel()
    acc = torch.stack([torch.nn.functional.binary_cross_entropy_with_logits(x, t, reduction="sum"), (s * t).sum(), s.sum(),
                       t.sum()]).double()
    dist.all_reduce(acc[1:4])
    U = acc[2] + acc[3]
    loss = acc[0] / n + 1.0 - (2.0 * acc[1] + eps) / U
    c1, c0 = world * 2.0 / U, world * (2.0 * acc[1] + eps) / (U * U)
    grad = (s - t) / n - (t * c1 - c0) * s * (1 - s)
    return loss, grad


def _worker(rank, world, port, out, algo="all_reduce", wire=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hiddenpose_amd.data_parallel import GradBucketReducer, all_reduce_dice_terms

    model = _model()
    if rank == 1:  # different initial weights: the reducer must broadcast rank 0's
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    red = GradBucketReducer(model, bucket_mb=0.002, algo=algo, wire_dtype=wire)  # ~2 KB buckets -> several buckets
    red.enable_timing(True)   # instrumentation on: per-bucket exchange times, exposed time (must not change any result)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(100 + rank)
    for step in range(2):
        x = torch.randn(8, 12, generator=g)
        y = torch.randn(8, 5, generator=g)
        red.zero_grad()
        loss = ((model(x) - y) ** 2).mean()
        loss.backward()
        red.finish()
        if step == 0:
            grads0 = [p.grad.clone() for p in model.parameters()]
        opt.step()
    i, p_, t = all_reduce_dice_terms(torch.tensor(1.0 + rank), torch.tensor(2.0 + rank), torch.tensor(3.0 + rank))
    # batch-global Dice (utils/criterion.py:358-368) under data parallelism: rank-local logits / targets
    gd = torch.Generator().manual_seed(500 + rank)
    xl = torch.randn(3, 40, generator=gd, dtype=torch.float64)
    tl = (torch.rand(3, 40, generator=gd) < 0.3).double()
    dl, dg = _dice_rank_loss_and_grad(xl, tl, world)
    dist.all_reduce(dl)
    dl /= world              # what averaging the per-rank losses reports
    tm = red.read_timing()
    out[rank] = {"timing": tm, "dice_loss": dl.item(), "dice_grad": dg / world,   # / world: the reducer's gradient averaging
"nb": len(red.buckets), "grads0": grads0, "params": [p.detach().clone() for p in model.parameters()],
                 "dice": (i.item(), p_.item(), t.item()),
                 "views": all(p.grad.data_ptr() >= red.flat[red._bucket_of[p]].data_ptr() for p in model.parameters())}
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,wire", [("all_reduce", None), ("rs_ag", None), ("a2a", None), ("all_reduce", torch.bfloat16)])
def test_bucketed_allreduce_world2(algo, wire):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, algo, wire), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["nb"] > 1 and r0["views"] and r1["views"]
    tm = r0["timing"]
    assert tm["algo"] == algo and tm["steps_timed"] == 2 and len(tm["bucket_exchange_ms"]) == r0["nb"] == len(tm["bucket_mb"])
    assert all(v is not None and v > 0 for v in tm["bucket_exchange_ms"]) and tm["wire_dtype"] == ("bfloat16" if wire else "float32")
    # expected step-0 gradient: average of the two ranks' local gradients from rank 0's weights
    exp = None
    for rank in range(world):
        model = _model()
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(8, 12, generator=g)
        y = torch.randn(8, 5, generator=g)
        ((model(x) - y) ** 2).mean().backward()
        gs = [p.grad.clone() for p in model.parameters()]
        exp = gs if exp is None else [a + b for a, b in zip(exp, gs)]
    exp = [e / world for e in exp]
    tol = dict(atol=1e-7, rtol=1e-5) if wire is None else dict(atol=2e-3, rtol=2e-2)   # bf16 on the wire: 8 bits
    for a, b, e in zip(r0["grads0"], r1["grads0"], exp):
        assert torch.allclose(a, b, atol=0, rtol=0)
        assert torch.allclose(a, e, **tol)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    assert r0["dice"] == r1["dice"] == (3.0, 5.0, 7.0)
    # the reference's loss on the CONCATENATED batch, by autograd, in one process
    xs, ts = [], []
    for rank in range(world):
        gd = torch.Generator().manual_seed(500 + rank)
        xs.append(torch.randn(3, 40, generator=gd, dtype=torch.float64))
        ts.append((torch.rand(3, 40, generator=gd) < 0.3).double())
    x = torch.cat(xs).requires_grad_(True)
    t = torch.cat(ts)
    p = torch.sigmoid(x)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(x, t) + 1.0 - (2.0 * (p * t).sum() + 1e-9) / (p.sum() + t.sum())
    ref.backward()
    assert abs(r0["dice_loss"] - ref.item()) < 1e-12 and abs(r1["dice_loss"] - ref.item()) < 1e-12
    assert torch.allclose(torch.cat([r0["dice_grad"], r1["dice_grad"]]), x.grad, atol=1e-14, rtol=1e-10)


def test_single_process_reducer_is_transparent():
    from hiddenpose_amd.data_parallel import GradBucketReducer

    m1, m2 = _model(), _model()
    red = GradBucketReducer(m1, bucket_mb=0.001)
    x = torch.randn(4, 12)
    red.zero_grad()
    m1(x).sum().backward()
    red.finish()
    m2(x).sum().backward()
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(a.grad, b.grad)
