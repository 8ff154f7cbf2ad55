#!/usr/bin/env python3
"""the bench's train_step leg alone (for rocprofv3: tools/trace_train.sh takes the tail of its kernel trace)"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
a = argparse.Namespace(train_find="search", train_steps=int(os.environ.get("STEPS", 6)), train_warmup=int(os.environ.get("WARMUP", 3)), share_gpu=False, legs_list=[])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
leg = bench.train_step_ssv_leg if os.environ.get("LEG") == "ssv" else bench.train_step_leg       # LEG=ssv: the self-supervised step
print(json.dumps(leg(a, 0, 1, dev)))
