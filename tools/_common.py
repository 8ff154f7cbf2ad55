"""shared by tools/train_3d.py and tools/validate_3d.py: config, synthetic data, model, checkpoints"""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

from selfpose3d_amd import distributed as D
from selfpose3d_amd.checkpoints import init_from_config, load_checkpoint, save_checkpoint  # noqa: F401 (re-exported)
from selfpose3d_amd.config import load_config
from selfpose3d_amd.multi_person_posenet import get_multi_person_pose_net
from selfpose3d_amd.synthetic_dataset import SyntheticPanoptic, SyntheticPanopticSSV


def setup(cfg_file, phase):
    cfg = load_config(cfg_file)
    rank, world, local = D.env_world()
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(local)
    D.init("nccl" if use_cuda else "gloo", device if use_cuda else None)
    out = os.path.join(cfg.OUTPUT_DIR, cfg.DATASET.TRAIN_DATASET, cfg.MODEL,
                       os.path.splitext(os.path.basename(cfg_file))[0])
    if rank == 0:
        os.makedirs(out, exist_ok=True)
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING,
                        format="%(asctime)-15s %(message)s")
    torch.backends.cudnn.benchmark = bool(cfg.CUDNN.BENCHMARK)
    return cfg, rank, world, device, out


def make_loader(cfg, frames, batch_per_gpu, rank, world, seed, shuffle, ssv=False):
    ds = (SyntheticPanopticSSV if ssv else SyntheticPanoptic)(cfg, num_frames=frames, seed=seed)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=shuffle) if world > 1 else None
    return torch.utils.data.DataLoader(ds, batch_size=batch_per_gpu, shuffle=(shuffle and sampler is None),
                                       sampler=sampler, num_workers=int(cfg.get("WORKERS", 0)), pin_memory=False,
                                       persistent_workers=False)
