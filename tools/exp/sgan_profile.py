#!/usr/bin/env python3
"""Which aten ops own the big elementwise copies of the SGAN step (torch.profiler, one step)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import radar_ml_amd as rml  # noqa

sgan = importlib.import_module("radar_ml_amd.sgan")
dev = torch.device("cuda", 0)
n = 256
d = sgan.define_discriminator(device=dev)
tr = sgan.DiscriminatorTrainer(d, amp_dtype="float16", ddp=False)
g = torch.Generator(device=dev).manual_seed(0)
x = [torch.rand((n, 128, 128), device=dev, generator=g) * 2 - 1 for _ in range(3)]
y = torch.randint(0, 3, (n,), device=dev, generator=g)
for _ in range(3):
    tr.train_on_batch_c(x, y)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_on_batch_c(x, y)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=40, max_shapes_column_width=60))
