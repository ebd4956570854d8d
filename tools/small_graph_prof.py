#!/usr/bin/env python3
"""One small config, graph-replayed only (for rocprofv3 --kernel-trace: per-kernel durations at replay pace).
    python tools/small_graph_prof.py cfg1|cfg2"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
r = bench.side_config(sys.argv[1], dev, True)
print("%s graph  %.1f us/step" % (sys.argv[1], r["ms_per_step"] * 1e3), flush=True)
