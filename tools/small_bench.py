#!/usr/bin/env python3
"""bench.py's cfg1 / cfg2 / cfg3 side runs alone (eager and graph-replayed), for quick iteration on the small-batch path."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for name, graphed in (("cfg1", False), ("cfg1", True), ("cfg2", False), ("cfg2", True)) + ((("cfg3", False),) if "--cfg3" in sys.argv else ()):
    r = bench.side_config(name, dev, graphed)
    print("%s graph=%d  %.1f us/step  %.0f samples/s" % (name, graphed, r["ms_per_step"] * 1e3, r["value"]), flush=True)
