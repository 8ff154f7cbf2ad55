#!/usr/bin/env python3
"""Validation entry point, CLI of the reference's tools/validate_3d.py (``--cfg X.yaml [--test-file F]
[--with-ssv]``).  The reference's call passes the wrong arity to validate_3d
(/root/reference/tools/validate_3d.py:96 vs lib/core/function.py:352); fixed here."""
import argparse
import logging
import os

import torch

from _common import make_loader, setup
from selfpose3d_amd.engine import validate_3d
from selfpose3d_amd.models import get_multi_person_pose_net, is_ssv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--test-file", default=None)
    ap.add_argument("--with-ssv", action="store_true")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--max-iters", type=int, default=None)
    ap.add_argument("--random-init", action="store_true",
                    help="validate untrained weights when no checkpoint exists (plumbing checks only)")
    args, _ = ap.parse_known_args()
    cfg, rank, world, device, out = setup(args.cfg, "validate")
    loader = make_loader(cfg, args.frames, int(cfg.TEST.BATCH_SIZE), rank, world, seed=2, shuffle=False)
    model = get_multi_person_pose_net(cfg, is_train=False).to(device)          # dispatch on cfg.MODEL
    if args.with_ssv != is_ssv(cfg):
        raise SystemExit(f"--with-ssv {'given' if args.with_ssv else 'not given'} but MODEL is {cfg.MODEL}: the two "
                         "models have different call signatures (lib/core/function.py:370-390)")
    test_file = args.test_file or os.path.join(out, str(cfg.TEST.MODEL_FILE))
    if os.path.isfile(test_file):
        logging.info(f"=> load models state {test_file}")
        sd = torch.load(test_file, map_location="cpu")
        model.load_state_dict(sd.get("state_dict", sd))
    elif args.random_init:
        logging.warning(f"=> no checkpoint at {test_file}: validating random-init weights (--random-init)")
    else:
        raise ValueError(f"Check the model file for testing! ({test_file})")    # reference tools/validate_3d.py:91-92
    if device.type == "cuda":
        model.use_channels_last(True)
    validate_3d(cfg, model, loader, 0, out, with_ssv=args.with_ssv, device=device, max_iters=args.max_iters)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
