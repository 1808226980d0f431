"""Training / evaluation callers of the NlosPose hot path.

Counterpart of utils/train_epoch.py:32-104 (inner loop), train.py:97-229 (seed 410,
Adam 1e-3, MultiStepLR([2,4,13], 0.2) stepped BEFORE each epoch, checkpoint dict) and
test.py:138-238 (eval = model.eval() + soft-argmax decode).  Host I/O of the reference
loop (np.savetxt every step, matplotlib dumps, per-step loss.item() sync) is not part of
the hot path and is left to the caller.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import ranges as R
from .criterion import BCEDiceLoss, L2JointLocationLoss, softmax_integral_tensor
from .optimizer import get_optimizer


def seed_everything(seed: int = 410) -> None:
    """train.py:89-98"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def build_training(cfg, model):
    """Losses, optimiser and schedule exactly as train.py:129-141."""
    criterion = L2JointLocationLoss(output_3d=True)
    # under torch.distributed the batch is the union of every rank's samples: Dice is batch-global in the reference
    import torch.distributed as dist

    voxel_criterion = BCEDiceLoss(global_batch=dist.is_available() and dist.is_initialized()
                                  and dist.get_world_size() > 1)
    optimizer = get_optimizer(cfg, model)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, cfg.TRAIN.LR_STEP, cfg.TRAIN.LR_FACTOR)
    return criterion, voxel_criterion, optimizer, scheduler


def compute_loss(model, criterion, voxel_criterion, meas, vol, target_joints):
    """train_epoch.py:38-44.  target_joints (B,24,3) or (B,72) in heat-map voxels."""
    output, feature = model(meas)
    b = output.shape[0]
    with R.stage("losses"):
        tj = target_joints.reshape(b, -1)
        joint_loss = criterion(output, tj, torch.ones_like(tj))
        voxel_loss = voxel_criterion(feature.reshape(b, -1), vol.reshape(b, -1))
        loss = R.mark_backward(joint_loss + voxel_loss, "losses")
    return loss, joint_loss, voxel_loss, output, feature


def train_step(model, criterion, voxel_criterion, optimizer, meas, vol, target_joints, reducer=None):
    """One iteration of train_epoch.py:32-76: forward, loss, zero_grad, backward, step.
    With a GradBucketReducer the gradient all-reduce overlaps backward."""
    loss, jl, vl, _, _ = compute_loss(model, criterion, voxel_criterion, meas, vol, target_joints)
    if reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad()
    with R.stage("backward"):
        loss.backward()
        if reducer is not None:
            reducer.finish()
    with R.stage("optimizer"):
        optimizer.step()
    return loss.detach(), jl.detach(), vl.detach()


def checkpoint_dict(model, optimizer, lr_scheduler, epoch, global_iter_num=None):
    """train.py:210-220 / train_epoch.py:78-90 checkpoint layout (test.py:133-136 needs all three)."""
    d = {"model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
         "lr_scheduler": lr_scheduler.state_dict(), "epoch": epoch}
    if global_iter_num is not None:
        d["global_iter_num"] = global_iter_num
    return d


def save_checkpoint(state: dict, path: str) -> None:
    """Rank 0 only (every rank holds the same replica), written to a temporary file and renamed so that a reader --
    or a second writer -- never sees a truncated checkpoint; the other ranks wait until the file exists."""
    import os

    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    if not distributed or dist.get_rank() == 0:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        tmp = f"{path}.tmp.{os.getpid()}"
        torch.save(state, tmp)
        os.replace(tmp, path)
    if distributed:
        dist.barrier()


@torch.no_grad()
def predict_joints(model, meas, cfg):
    """test.py:156-157: eval-mode forward + soft-argmax decode -> (B, 72) voxel coordinates."""
    model.eval()
    output, _ = model(meas)
    hm = cfg.DATASET.HEATMAP_SIZE
    return softmax_integral_tensor(output, cfg.DATASET.NUM_JOINTS, True, hm[0], hm[1], hm[2])


def train_epoch(cfg, train_loader, model, criterion, voxel_criterion, optimizer, epoch, output_dir, writer, begin_time,
                save_model_dir, lr_scheduler, reducer=None, max_steps=None):
    """One epoch with the call signature of utils/train_epoch.py:20-104 (positional order as train.py:162 passes it).
    Per batch: forward, L2Joint + BCEDice loss, zero_grad, backward, step; a checkpoint dict every 10000 global
    iterations (:78-90) and the running 100-iteration loss (:92-97).  The reference's per-step `loss.item()` print,
    `np.savetxt` and matplotlib dumps are not reproduced: the loss is accumulated on the device and read back once
    per 100 iterations, so the loop never stalls the GPU.  `writer` may be None.  Returns the mean loss of the epoch."""
    import os
    import time

    model.train()
    device = next(model.parameters()).device
    n_batches = len(train_loader)
    total_step = (cfg.TRAIN.END_EPOCH - cfg.TRAIN.BEGIN_EPOCH) * n_batches
    window = torch.zeros((), dtype=torch.float32, device=device)
    epoch_sum = torch.zeros((), dtype=torch.float32, device=device)
    t_window = epoch_t0 = time.time()
    steps = 0
    for step, (meas, vol, target_joints, _ids) in enumerate(train_loader):
        if max_steps is not None and step >= max_steps:
            break
        global_iter_num = epoch * n_batches + step + 1
        meas = torch.as_tensor(meas).to(device=device, dtype=torch.float32)
        vol = torch.as_tensor(vol).to(device=device, dtype=torch.float32)
        target_joints = torch.as_tensor(target_joints).to(device=device, dtype=torch.float32)  # (B, 24, 3)
        loss, _, _ = train_step(model, criterion, voxel_criterion, optimizer, meas, vol, target_joints, reducer)
        window += loss
        epoch_sum += loss
        steps += 1
        if global_iter_num % 10000 == 0:
            save_checkpoint(checkpoint_dict(model, optimizer, lr_scheduler, epoch, global_iter_num),
                            os.path.join(save_model_dir, f"NlosPose_dict_iter{global_iter_num}.pth"))
        if global_iter_num % 100 == 0:
            mean100 = float(window) / 100.0
            window.zero_()
            now = time.time()
            print(f"global iter is {global_iter_num}, loss is {mean100}, iter100 time is {now - t_window}, "
                  f"{global_iter_num / max(total_step, 1) * 100}%")
            t_window = now
            if writer is not None:
                writer.add_scalar("Train Loss", mean100, global_iter_num)
    if writer is not None:
        writer.add_scalar("batch_time", time.time() - epoch_t0, epoch)
    return float(epoch_sum) / max(steps, 1)
