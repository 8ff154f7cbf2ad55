"""Train / validate loops with the reference's call signatures (/root/reference/lib/core/function.py:
219-349 ``train_3d``, :352-489 ``validate_3d``, :492-508 ``AverageMeter``) for one process per GPU.

"Speed" keeps the reference's definition, views x frames / batch time (function.py:318), and is also
reported per frame.  Debug image dumps and dataset metrics (need the real datasets) are not here.
"""
from __future__ import annotations

import logging
import time

import torch

logger = logging.getLogger(__name__)


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = float(val)
        self.sum += float(val) * n
        self.count += n
        self.avg = self.sum / max(1, self.count)


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def zero_anchor(params):
    """0 * (first element of every trainable parameter): added to a loss it makes EVERY trainable parameter part of
    the autograd graph with an exactly-zero gradient contribution.  DistributedDataParallel's gradient all-reduce is a
    collective: a rank whose loss reached no parameter (no valid proposal in its frames) must still run backward() with
    every bucket, or the other ranks wait forever.  The reference gets the same effect from zero-weighted dummy forwards
    (lib/models/multi_person_posenet_ssv.py:290,429,496,499).  None when nothing is trainable."""
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return None
    return torch.cat([p.reshape(-1)[:1] for p in ps]).sum() * 0.0


def anchor_unreached(term, modules, reached):
    """term + zero anchors of every sub-net in `modules` ({name: module or None}) whose name is not in `reached`: the
    models call this on EVERY return path of a training forward with the set of sub-nets their loss terms really went
    through, so that whatever branch a configuration (or one rank's batch) takes, each trainable parameter is in the
    graph - DistributedDataParallel(find_unused_parameters=False) needs exactly that, the reference's DataParallel did not
    care (attention net on the TRAIN_ONLY_* / INIT_TRAIN_EPOCHS_ROOTNET / SINGLE_AUG paths, pose net without 2D
    targets, root net without 3D targets)."""
    for name, mod in modules.items():
        if mod is None or name in reached:
            continue
        a = zero_anchor(mod.parameters())
        if a is not None:
            term = term + a
    return term


def _to_host_async(t):
    """queue ONE device -> pinned-host copy of a small tensor on the current stream (really asynchronous: a pageable
    destination would make the copy wait for the stream); the caller synchronises before reading"""
    if not t.is_cuda:
        return t
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    return host


def _to_device(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(dev, non_blocking=True)
    if isinstance(x, (list, tuple)):
        return [_to_device(v, dev) for v in x]
    return x


def train_3d(config, model, optimizer, loader, epoch, output_dir=None, writer_dict=None, device=None, max_iters=None):
    net = _unwrap(model)
    device = device or next(net.parameters()).device
    bt, dt, losses, l2d, l3d, lcord = (AverageMeter() for _ in range(6))
    model.train()
    if not config.NETWORK.TRAIN_BACKBONE and net.backbone is not None:
        net.backbone.eval()                                                     # function.py:228-230
    end = time.time()
    for i, (inputs, targets_2d, weights_2d, targets_3d, meta, input_heatmap) in enumerate(loader):
        if max_iters is not None and i >= max_iters:
            break
        dt.update(time.time() - end)
        inputs = _to_device(inputs, device)
        if config.NETWORK.TRAIN_ONLY_2D:
            loss_2d, _ = model(views=inputs, meta=meta, targets_2d=targets_2d, weights_2d=weights_2d)
            loss = loss_2d.mean()
            parts = [loss]
        else:
            _, _, _, loss_2d, loss_3d, loss_cord = model(views=inputs, meta=meta, targets_2d=targets_2d,
                                                         weights_2d=weights_2d, targets_3d=targets_3d[0])
            loss_2d, loss_3d, loss_cord = loss_2d.mean(), loss_3d.mean(), loss_cord.mean()
            loss = loss_2d + loss_3d + loss_cord                                  # function.py:279
            parts = [loss_2d, loss_3d, loss_cord, loss]
        # the meters' values leave the device in ONE asynchronous copy, read after backward + step were queued (the
        # reference's four .item() calls per iteration, function.py:262-281, each stall the stream)
        stacked = torch.stack([p.detach().float() for p in parts])
        host = _to_host_async(stacked)
        optimizer.zero_grad(set_to_none=True)
        if not loss.requires_grad:
            # nothing trainable was reached on THIS rank: backward() must run all the same (see zero_anchor); the models
            # anchor their own skipped sub-nets inside forward, this is the last line of defence
            anchor = zero_anchor(net.parameters())
            if anchor is not None:
                loss = loss + anchor
        if loss.requires_grad:
            loss.backward()
            optimizer.step()
        if stacked.is_cuda:
            torch.cuda.current_stream(stacked.device).synchronize()
        vals = host.tolist()
        if len(vals) == 1:
            l2d.update(vals[0])
        else:
            l2d.update(vals[0]); l3d.update(vals[1]); lcord.update(vals[2])
        losses.update(vals[-1])
        bt.update(time.time() - end)
        end = time.time()
        if i % int(config.PRINT_FREQ) == 0:
            V, B = len(inputs), inputs[0].size(0)
            mem = torch.cuda.memory_allocated(device) / 2 ** 30 if device.type == "cuda" else 0.0
            logger.info(f"Epoch: [{epoch}][{i}/{len(loader)}]\tTime: {bt.val:.3f}s ({bt.avg:.3f}s)\t"
                        f"Speed: {V * B / max(bt.val, 1e-9):.1f} samples/s ({B / max(bt.val, 1e-9):.1f} frames/s)\t"
                        f"Data: {dt.val:.3f}s ({dt.avg:.3f}s)\tLoss: {losses.val:.6f} ({losses.avg:.6f})\t"
                        f"Loss_2d: {l2d.val:.7f} ({l2d.avg:.7f})\tLoss_3d: {l3d.val:.7f} ({l3d.avg:.7f})\t"
                        f"Loss_cord: {lcord.val:.6f} ({lcord.avg:.6f})\tMemory {mem:.1f} GB")
            if writer_dict and writer_dict.get("writer") is not None:
                w, g = writer_dict["writer"], writer_dict["train_global_steps"]
                w.add_scalar("train_loss_3d", l3d.val, g); w.add_scalar("train_loss_cord", lcord.val, g)
                w.add_scalar("train_loss", losses.val, g)
                writer_dict["train_global_steps"] = g + 1
    return {"loss": losses.avg, "loss_2d": l2d.avg, "loss_3d": l3d.avg, "loss_cord": lcord.avg,
            "batch_time": bt.avg}


class ScalarLog:
    """``add_scalar`` sink used when neither tensorboardX (what the reference imports, tools/train_3d.py:32) nor
    torch.utils.tensorboard is importable: one JSON line per scalar in <log_dir>/scalars.jsonl"""

    def __init__(self, log_dir):
        import os
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self._f = open(self.path, "a")

    def add_scalar(self, tag, value, global_step=None):
        import json
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": None if global_step is None else int(global_step)}) + "\n")
        self._f.flush()

    def close(self):
        self._f.close()


def make_writer_dict(log_dir):
    """the reference's writer_dict (tools/train_3d.py:183-187): a SummaryWriter when a TensorBoard package is present,
    else ScalarLog; same scalar names either way (lib/core/function.py:155-174,329-334)"""
    writer = None
    for mod in ("tensorboardX", "torch.utils.tensorboard"):
        try:
            writer = __import__(mod, fromlist=["SummaryWriter"]).SummaryWriter(log_dir=log_dir)
            break
        except Exception:
            continue
    return {"writer": writer if writer is not None else ScalarLog(log_dir), "train_global_steps": 0, "valid_global_steps": 0}


SSV_LOSS_KEYS = ("loss_2d", "loss_root_reg", "loss_root_syn", "loss_pose3d_ssv", "loss_attn_ssv", "loss_pose3d_l1_ssv")


def train_3d_ssv(config, model, optimizer, loader, epoch, output_dir=None, writer_dict=None, device=None, max_iters=None):
    """self-supervised loop (/root/reference/lib/core/function.py:27-217): three view sets per frame, a dict of loss
    terms, total = sum of the terms that require grad (:107)"""
    net = _unwrap(model)
    device = device or next(net.parameters()).device
    bt, dt = AverageMeter(), AverageMeter()
    meters = {k: AverageMeter() for k in SSV_LOSS_KEYS + ("losses",)}
    model.train()
    if not config.NETWORK.TRAIN_BACKBONE and net.backbone is not None:
        net.backbone.eval()                                                     # function.py:43-45
    if config.NETWORK.get("FREEZE_ROOTNET", False) and getattr(net, "root_net", None) is not None:
        net.root_net.eval()                                                     # :46-48
    end = time.time()
    for i, batch in enumerate(loader):
        if max_iters is not None and i >= max_iters:
            break
        dt.update(time.time() - end)
        (in1, t2d1, w2d1, t3d1, meta1, _, in2, t2d2, w2d2, t3d2, meta2, _, in3, t2d3, w2d3, t3d3, meta3, _) = batch
        in1, in2, in3 = (_to_device(x, device) for x in (in1, in2, in3))
        _, _, _, loss_dict = model(views1=in1, meta1=meta1, targets_2d1=t2d1, weights_2d1=w2d1, targets_3d1=t3d1[0],
                                   views2=in2, meta2=meta2, targets_2d2=t2d2, weights_2d2=w2d2, targets_3d2=t3d2[0],
                                   views3=in3, meta3=meta3, targets_2d3=t2d3, weights_2d3=w2d3, targets_3d3=t3d3[0],
                                   epoch=epoch)
        terms = [v.mean() for v in loss_dict.values() if v.requires_grad]        # :107
        optimizer.zero_grad(set_to_none=True)
        if not terms:                                                            # every rank must run backward (zero_anchor)
            anchor = zero_anchor(net.parameters())
            terms = [] if anchor is None else [anchor]
        loss = sum(terms) if terms else None
        # ONE device -> host transfer per iteration for all the meters (the reference reads every term with .item(),
        # function.py:109-131: up to seven stream-serialising syncs per step), issued before backward so that it overlaps it
        keys = list(loss_dict.keys())
        stacked = torch.stack([loss_dict[k].detach().mean().float() for k in keys] +
                              ([loss.detach().float()] if loss is not None else []))
        host = _to_host_async(stacked)
        if loss is not None:
            loss.backward()
            optimizer.step()
        if stacked.is_cuda:
            torch.cuda.current_stream(stacked.device).synchronize()
        vals = host.tolist()
        for k, v in zip(keys, vals):
            meters[k].update(v)
        if loss is not None:
            meters["losses"].update(vals[-1])
        bt.update(time.time() - end)
        end = time.time()
        if i % int(config.PRINT_FREQ) == 0:
            V, B = len(in1), in1[0].size(0)
            logger.info(f"Epoch: [{epoch}][{i}/{len(loader)}]\tTime: {bt.val:.3f}s ({bt.avg:.3f}s)\t"
                        f"Speed: {V * B / max(bt.val, 1e-9):.1f} samples/s\tData: {dt.val:.3f}s\t" +
                        "\t".join(f"{k}: {m.val:.6f} ({m.avg:.6f})" for k, m in meters.items()))
            if writer_dict and writer_dict.get("writer") is not None:            # function.py:155-174
                w, g = writer_dict["writer"], writer_dict["train_global_steps"]
                w.add_scalar("train_loss_2d", meters["loss_2d"].val, g)
                w.add_scalar("train_loss_root", meters["loss_root_syn"].val + meters["loss_root_reg"].val, g)
                w.add_scalar("train_loss_pose3d_ssv", meters["loss_pose3d_ssv"].val, g)
                w.add_scalar("train_loss_attn_ssv", meters["loss_attn_ssv"].val, g)
                w.add_scalar("train_loss", meters["losses"].val, g)
                writer_dict["train_global_steps"] = g + 1
    out = {k: m.avg for k, m in meters.items()}
    out["batch_time"] = bt.avg
    return out


@torch.no_grad()
def validate_3d(config, model, loader, epoch=0, output_dir=None, with_ssv=False, device=None, max_iters=None):
    """-> precision proxy.  The reference scores AP/MPJPE with the dataset's evaluate() (needs the real
    data); with synthetic frames we return the fraction of GT roots matched within 150 mm."""
    net = _unwrap(model)
    device = device or next(net.parameters()).device
    model.eval()
    bt = AverageMeter()
    preds, matched, total = [], 0, 0
    end = time.time()
    for i, (inputs, targets_2d, weights_2d, targets_3d, meta, input_heatmap) in enumerate(loader):
        if max_iters is not None and i >= max_iters:
            break
        inputs = _to_device(inputs, device)
        if with_ssv:                                                            # function.py:370-376
            pred, _, grid_centers = model(views1=inputs, meta1=meta, input_heatmaps1=input_heatmap, inference=True)
        else:
            pred, _, grid_centers, _, _, _ = model(views=inputs, meta=meta)
        preds.append(pred.detach().cpu())
        gc = grid_centers.detach().cpu()
        roots, nper = meta[0]["roots_3d"].float(), meta[0]["num_person"]
        for b in range(gc.shape[0]):
            ok = gc[b, :, 3] >= 0
            for p in range(int(nper[b])):
                total += 1
                if bool(ok.any()) and float((gc[b, ok, :3] - roots[b, p]).norm(dim=-1).min()) < 150.0:
                    matched += 1
        bt.update(time.time() - end)
        end = time.time()
        if i % int(config.PRINT_FREQ) == 0:
            V, B = len(inputs), inputs[0].size(0)
            logger.info(f"Test: [{i}/{len(loader)}]\tTime: {bt.val:.3f}s ({bt.avg:.3f}s)\t"
                        f"Speed: {V * B / max(bt.val, 1e-9):.1f} samples/s ({B / max(bt.val, 1e-9):.1f} frames/s)")
    from . import distributed as D
    matched, total = D.sum_over_ranks(matched, total, device=device if device.type == "cuda" else None)   # score ALL shards
    recall = matched / max(1, total)
    logger.info(f"root recall@150mm on synthetic frames: {recall:.4f} ({int(matched)}/{int(total)})")
    return recall
