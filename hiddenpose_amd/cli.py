"""Shared pieces of the train.py / test.py entry points: the reference's command line (train.py:29-74,
test.py:29-74), its config patching (utils/record.py:42-60, train.py:77-86) and checkpoint loading
(test.py:131-136)."""
from __future__ import annotations

import argparse

import torch

from .config import get_cfg_defaults, update_config_t128_128x128


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="HiddenPose network args")
    p.add_argument("--model", type=str, default="", help="model directory / checkpoint file")
    p.add_argument("--test", type=str, default="", help="test options: test_pose_v2 | test_realdata | test_fk")
    p.add_argument("--log", type=str, default="", help="log directory")
    p.add_argument("--data", type=str, default="", help="data directory (DATASET.TRAIN_PATH / TEST_PATH)")
    p.add_argument("--device", type=int, default=0, help="index of GPU to use")
    p.add_argument("--PHASE", type=str, default="train", help="'eval' or 'continue_train' or 'train' or 'test'")
    # not in the reference: bounded runs for smoke tests, and where a measured .mat lives for test_realdata
    p.add_argument("--max-steps", type=int, default=None, help="stop every epoch after this many iterations")
    p.add_argument("--epochs", type=int, default=None, help="override TRAIN.END_EPOCH")
    p.add_argument("--realdata", type=str, default="data/lct256_human.mat")
    return p.parse_args(argv)


def build_config(args):
    """config_noise defaults -> command line -> the 128 x 128 x 128 shape every reference run uses."""
    cfg = get_cfg_defaults()
    cfg.defrost()
    if args.model:
        cfg.MODEL.LOCATION = args.model
    if args.test:
        cfg.TEST.TYPE = args.test
    if args.log:
        cfg.LOG_DIR = args.log
    if args.data:
        cfg.DATA_DIR = args.data
        cfg.DATASET.TRAIN_PATH = args.data
        cfg.DATASET.TEST_PATH = args.data
    cfg.DEVICE = args.device
    cfg.PHASE = args.PHASE
    if args.epochs is not None:
        cfg.TRAIN.END_EPOCH = args.epochs
    cfg.freeze()
    update_config_t128_128x128(cfg)
    return cfg


def load_checkpoint(path, model, optimizer=None, lr_scheduler=None, device="cuda:0"):
    ck = torch.load(path, map_location=device)
    model.load_state_dict(ck["model_state_dict"], strict=True)
    if optimizer is not None and "optimizer_state_dict" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if lr_scheduler is not None and "lr_scheduler" in ck:
        lr_scheduler.load_state_dict(ck["lr_scheduler"])
    return ck
