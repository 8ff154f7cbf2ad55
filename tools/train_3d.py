#!/usr/bin/env python3
"""Training entry point, CLI of the reference's tools/train_3d.py (``--cfg X.yaml``), one process per
GPU:   torchrun --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_3d.py --cfg X.yaml

Replaces single-process nn.DataParallel (/root/reference/tools/train_3d.py:140) with
DistributedDataParallel over RCCL (gradient all-reduce only); ``TRAIN.BATCH_SIZE`` stays the per-GPU
batch.  Stage flags (TRAIN_BACKBONE / TRAIN_ONLY_ROOTNET / FREEZE_ROOTNET / USE_GT) select trainable
parameters as in tools/train_3d.py:48-75; checkpoints keep the reference's names and keys, and the stage
hand-off (NETWORK.PRETRAINED_BACKBONE [+ _PSEUDOGT] / INIT_ROOTNET / INIT_ALL, tools/train_3d.py:150-180) is
selfpose3d_amd.checkpoints.init_from_config.
The real datasets are not in the image: frames come from SyntheticPanoptic (--frames).
"""
import argparse
import logging
import os

import torch

from _common import init_from_config, load_checkpoint, make_loader, save_checkpoint, setup
from selfpose3d_amd import distributed as D
from selfpose3d_amd.engine import make_writer_dict, train_3d, train_3d_ssv, validate_3d
from selfpose3d_amd.models import get_multi_person_pose_net, is_ssv

logger = logging.getLogger("train_3d")


def select_trainable(model, cfg):
    def req(mod, flag):
        if mod is not None:
            for p in mod.parameters():
                p.requires_grad = flag
    req(model.backbone, bool(cfg.NETWORK.TRAIN_BACKBONE))
    if not cfg.NETWORK.TRAIN_ONLY_2D:
        req(getattr(model, "pose_net", None), not cfg.NETWORK.TRAIN_ONLY_ROOTNET)
        # proposals from ground truth: the root net is never called, so it must not ask for gradients either
        req(getattr(model, "root_net", None), (not cfg.NETWORK.USE_GT) and (not cfg.NETWORK.FREEZE_ROOTNET))
    return [p for p in model.parameters() if p.requires_grad]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--frames", type=int, default=32, help="synthetic frames per epoch (global)")
    ap.add_argument("--max-iters", type=int, default=None)
    args, _ = ap.parse_known_args()
    cfg, rank, world, device, out = setup(args.cfg, "train")
    torch.manual_seed(D.rank_seed(int(cfg.get("SEED", 0)), rank))       # per-rank streams (sampling, augmentation)
    ssv = is_ssv(cfg)
    if bool(cfg.get("WITH_SSV", False)) != ssv:
        raise ValueError(f"{args.cfg}: WITH_SSV = {cfg.get('WITH_SSV', False)} but MODEL = {cfg.MODEL}: the self-supervised "
                         "loop and model go together (reference tools/train_3d.py:121,167-170)")
    train_loader = make_loader(cfg, args.frames, int(cfg.TRAIN.BATCH_SIZE), rank, world, seed=1,
                               shuffle=bool(cfg.TRAIN.SHUFFLE), ssv=ssv)
    test_loader = make_loader(cfg, max(world, args.frames // 4), int(cfg.TEST.BATCH_SIZE), rank, world, seed=2,
                              shuffle=False)
    model = get_multi_person_pose_net(cfg, is_train=True).to(device)
    if device.type == "cuda":
        model.use_channels_last(True)
    params = select_trainable(model, cfg)
    optimizer = torch.optim.Adam(params, lr=float(cfg.TRAIN.LR))
    # stage hand-off (reference tools/train_3d.py:150-180): PRETRAINED_BACKBONE, INIT_ROOTNET, INIT_ALL, then RESUME;
    # every rank reads the files itself, a named file that is missing raises
    for what in init_from_config(model, cfg):
        logger.info(f"=> initialised from NETWORK.{what} = {cfg.NETWORK[what]}")
    start, best, last = (load_checkpoint(model, optimizer, out) if cfg.TRAIN.RESUME else (int(cfg.TRAIN.BEGIN_EPOCH), 0.0, -1))
    ddp = D.wrap_ddp(model, device, find_unused=D.needs_find_unused(cfg))
    sched = torch.optim.lr_scheduler.MultiStepLR(optimizer, list(cfg.TRAIN.LR_STEP), float(cfg.TRAIN.LR_FACTOR),
                                                 last_epoch=last)
    # scalars as the reference logs them (train_loss_* per PRINT_FREQ iterations), rank 0 only
    writer_dict = make_writer_dict(os.path.join(out, str(cfg.LOG_DIR))) if rank == 0 else None
    for epoch in range(start, int(cfg.TRAIN.END_EPOCH)):
        if hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        loop = train_3d_ssv if ssv else train_3d                       # reference tools/train_3d.py:167-170
        stats = loop(cfg, ddp, optimizer, train_loader, epoch, out, writer_dict, device, args.max_iters)
        sched.step()
        prec = None if cfg.NETWORK.TRAIN_ONLY_2D else validate_3d(cfg, ddp, test_loader, epoch, out, with_ssv=ssv,
                                                                  device=device, max_iters=args.max_iters)
        is_best = prec is not None and prec > best
        best = max(best, prec or 0.0)
        if rank == 0:
            logger.info(f"epoch {epoch}: {stats}  precision {prec}")
            save_checkpoint({"epoch": epoch + 1, "state_dict": model.state_dict(), "precision": best,
                             "optimizer": optimizer.state_dict()}, is_best, out)
    if rank == 0:
        torch.save(model.state_dict(), os.path.join(out, "final_state.pth.tar"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
