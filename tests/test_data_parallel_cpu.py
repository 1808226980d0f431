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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hiddenpose_amd.data_parallel import GradBucketReducer, all_reduce_dice_terms

    model = _model()
    if rank == 1:  # different initial weights: the reducer must broadcast rank 0's
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    red = GradBucketReducer(model, bucket_mb=0.002)  # ~2 KB buckets -> several buckets
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
    out[rank] = {"nb": len(red.buckets), "grads0": grads0, "params": [p.detach().clone() for p in model.parameters()],
                 "dice": (i.item(), p_.item(), t.item()),
                 "views": all(p.grad.data_ptr() >= red.flat[red._bucket_of[p]].data_ptr() for p in model.parameters())}
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["nb"] > 1 and r0["views"] and r1["views"]
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
    for a, b, e in zip(r0["grads0"], r1["grads0"], exp):
        assert torch.allclose(a, b, atol=0, rtol=0)
        assert torch.allclose(a, e, atol=1e-7, rtol=1e-5)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    assert r0["dice"] == r1["dice"] == (3.0, 5.0, 7.0)


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
