#!/usr/bin/env python3
"""Training entry point with the reference's command line (train.py:29-229):

    python train.py --data /path/to/pose_v2_noise [--PHASE train|continue_train --model ckpt.pth --log ./log]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --data ...   # data parallel
    (under the launcher write --log=DIR: a bare "--log" is ambiguous for torch.distributed.run's own parser)

seed 410; NlosPose at 128 x 128 x 128; NlosPoseDataset with the per-sample ingest on the GPU; Adam(lr 1e-3);
MultiStepLR([2, 4, 13], 0.2) stepped BEFORE each epoch (train.py:193); L2Joint + BCEDice; a checkpoint dict per
epoch ({model_state_dict, optimizer_state_dict, lr_scheduler, epoch}).  Under torch.distributed each rank takes
its own shard of the dataset and gradients are averaged with bucketed RCCL all-reduce overlapped with backward."""
from __future__ import annotations

import os
import time

import torch
from torch.utils.data import DataLoader

from hiddenpose_amd.cli import build_config, load_checkpoint, parse_args
from hiddenpose_amd.NlosPose import NlosPose
from hiddenpose_amd.nlos_pose_dataloader import NlosPoseDataset, PrefetchingLoader
from hiddenpose_amd.train_epoch import build_training, checkpoint_dict, save_checkpoint, seed_everything, train_epoch


def _collate(batch):
    meas, vol, joints, ids = zip(*batch)
    return torch.stack(meas), torch.stack(vol), torch.stack([torch.as_tensor(j, dtype=torch.float32) for j in joints]), list(ids)


def main(argv=None):
    # must be in the environment before the first HIP call initialises the runtime (dmabuf IPC for RCCL);
    # seed_everything below is the first call that touches HIP (torch.cuda.is_available)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    seed_everything(410)
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.device)))
    args.device = (0 if os.environ.get("HP_SHARE_GPU") else local) if world > 1 else args.device
    cfg = build_config(args)
    torch.cuda.set_device(cfg.DEVICE)
    reducer = None
    if world > 1:
        import torch.distributed as dist

        from hiddenpose_amd.data_parallel import GradBucketReducer

        backend = os.environ.get("HP_DIST_BACKEND", "nccl")  # "gloo" only to rehearse on a box without several GPUs
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", cfg.DEVICE))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    model = NlosPose(cfg).to(torch.device("cuda", cfg.DEVICE))
    if rank == 0:
        print(f"Total number of parameters: {sum(p.numel() for p in model.parameters())}")
    criterion, voxel_criterion, optimizer, lr_scheduler = build_training(cfg, model)
    if world > 1:
        reducer = GradBucketReducer(model)

    data = NlosPoseDataset(cfg, cfg.DATASET.TRAIN_PATH, device=torch.device("cuda", cfg.DEVICE))
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler

        sampler = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=410)
    # Host threads read + expand the .hdr files into pinned memory, a side stream copies and runs the ingest kernels:
    # the training stream never waits for a file (HP_LOADER=simple: the in-process torch DataLoader instead)
    if os.environ.get("HP_LOADER", "prefetch") == "simple":
        loader = DataLoader(data, batch_size=cfg.TRAIN.BATCH_SIZE, shuffle=sampler is None, sampler=sampler, num_workers=0,
                            collate_fn=_collate, drop_last=True)
    else:
        loader = PrefetchingLoader(data, cfg.TRAIN.BATCH_SIZE, sampler=sampler, shuffle=sampler is None, drop_last=True,
                                   depth=2, workers=int(os.environ.get("HP_LOADER_WORKERS", "4")), seed=410)

    stamp = f"{time.gmtime().tm_mon}_{time.gmtime().tm_mday}_{cfg.LOSS.TYPE}_{cfg.MODEL.COORD_REPRESENTATION}"
    save_model_dir = os.path.join(cfg.RESULT.FINAL_OUTPUT_DIR, stamp)
    writer = None
    if rank == 0:
        try:
            from torch.utils.tensorboard import SummaryWriter

            writer = SummaryWriter(os.path.join(cfg.LOG_DIR, stamp))
        except Exception:  # tensorboard is optional
            writer = None

    begin_epoch = cfg.TRAIN.BEGIN_EPOCH
    if cfg.PHASE == "continue_train":
        ck = load_checkpoint(cfg.MODEL.LOCATION, model, optimizer, lr_scheduler, device=f"cuda:{cfg.DEVICE}")
        begin_epoch = ck["epoch"] + 1
    begin_time = time.time()
    for epoch in range(begin_epoch, cfg.TRAIN.END_EPOCH):
        t0 = time.time()
        if sampler is not None:
            sampler.set_epoch(epoch)
        if hasattr(loader, "set_epoch"):
            loader.set_epoch(epoch)
        lr_scheduler.step()  # the reference steps the schedule before the epoch's first optimizer step
        mean_loss = train_epoch(cfg, loader, model, criterion, voxel_criterion, optimizer, epoch, cfg.RESULT.FINAL_OUTPUT_DIR,
                                writer, begin_time, save_model_dir, lr_scheduler, reducer=reducer, max_steps=args.max_steps)
        if rank == 0:
            dt = time.time() - t0
            steps = len(loader) if args.max_steps is None else min(len(loader), args.max_steps)
            print(f"epoch {epoch} used {dt}, mean loss {mean_loss}, left {dt * (cfg.TRAIN.END_EPOCH - epoch - 1) / 3600} hours, "
                  f"{steps * cfg.TRAIN.BATCH_SIZE * world / max(dt, 1e-9):.2f} samples/s (files -> ingest -> train step)")
        save_checkpoint(checkpoint_dict(model, optimizer, lr_scheduler, epoch),
                        os.path.join(save_model_dir, f"NlosPose_final_dict_{epoch}.pth"))
    if rank == 0:
        os.makedirs(save_model_dir, exist_ok=True)
        with open(os.path.join(save_model_dir, "NlosPose.log"), "w", encoding="utf-8") as fh:
            fh.write(str(cfg))
        print("finished training")
    if world > 1:
        torch.distributed.destroy_process_group()
    return save_model_dir


if __name__ == "__main__":
    main()
