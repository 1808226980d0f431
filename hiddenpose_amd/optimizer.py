"""models/optimizer.py:9-24 `get_optimizer`: Adam(lr) for cfg.TRAIN.OPTIMIZER == 'adam'
(the reference's SGD branch reads cfg keys that do not exist, so only Adam is live)."""
from torch import optim


def get_optimizer(cfg, model):
    if cfg.TRAIN.OPTIMIZER == "adam":
        params = list(model.parameters())
        # same update rule and state_dict layout; on the GPU the single fused kernel replaces ~50 multi-tensor launches
        fused = bool(params) and all(p.is_cuda for p in params) and __import__("os").environ.get("HP_ADAM_FUSED", "1") != "0"
        return optim.Adam(params, lr=cfg.TRAIN.LR, fused=fused)
    if cfg.TRAIN.OPTIMIZER == "sgd":
        return optim.SGD(model.parameters(), lr=cfg.TRAIN.LR, momentum=getattr(cfg.TRAIN, "MOMENTUM", 0.9),
                         weight_decay=getattr(cfg.TRAIN, "WD", 0.0), nesterov=getattr(cfg.TRAIN, "NESTEROV", False))
    return None
