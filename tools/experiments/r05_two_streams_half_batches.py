#!/usr/bin/env python3
"""Experiment (round 5, after the default library became safe on shared GPUs): does the headline step get faster when its batch
of 4 frames runs as TWO concurrent graphs of 2 frames on two streams (tails and small kernels of one overlapping the other)?
Prints one JSON line: ms per 4 frames for (a) one graph of batch 4, (b) two graphs of batch 2 replayed on two streams,
(c) the same two graphs replayed one after the other on one stream."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from selfpose3d_amd.graphs import GraphedRootNet

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
N = int(os.environ.get("STEPS", 300))


def graphed(batch, rank):
    cfg, meta, hms, model, golden = bench.build_workload(batch, rank, dev)
    return GraphedRootNet(model, hms, meta)


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


g4 = graphed(4, 0)
one = timed(lambda: g4.graph.replay())
ga, gb = graphed(2, 0), graphed(2, 1)
ref_a, ref_b = ga.out[0].clone() if isinstance(ga.out, (tuple, list)) else ga.out.clone(), None
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def two_streams():
    with torch.cuda.stream(sa):
        ga.graph.replay()
    with torch.cuda.stream(sb):
        gb.graph.replay()


def one_stream():
    ga.graph.replay(); gb.graph.replay()


ser = timed(one_stream)
torch.cuda.synchronize()
first = (ga.out[0] if isinstance(ga.out, (tuple, list)) else ga.out).clone()
par = timed(two_streams)
torch.cuda.synchronize()
same = torch.equal(first, ga.out[0] if isinstance(ga.out, (tuple, list)) else ga.out)
print(json.dumps({"ms_per_4_frames": {"one_graph_batch4": round(one, 4), "two_graphs_batch2_two_streams": round(par, 4),
                                      "two_graphs_batch2_one_stream": round(ser, 4)}, "steps": N,
                  "two_stream_result_bit_identical_to_one_stream": bool(same)}))
