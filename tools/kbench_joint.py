#!/usr/bin/env python3
"""Micro-bench of the fused joint-training step (BASELINE configs[4]): 300 steps over 8 images, convexity prior and the
path-connected (RealNVP) prior, for rocprofv3 --kernel-trace --stats (tools/profile_round.sh)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
res = bench.joint_step_variant(dev, 256, lambda h, c, l: 2 * h * c + l * (2 * h * h + 2 * h * c) + 2 * h + 2 * c)
print(json.dumps(res, indent=1))
