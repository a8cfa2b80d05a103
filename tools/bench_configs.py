"""BASELINE.json configs 4 and 5 on one MI355X (the same functions bench.py puts under `configs`):
usage: python tools/bench_configs.py [4|5]   (one JSON line per config; run under rocprofv3 for kernel tables)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    dev = torch.device("cuda:0")
    if which in ("4", "both"):
        print(json.dumps({"config": 4, **bench.config4(dev)}), flush=True)
    if which in ("5", "both"):
        print(json.dumps({"config": 5, **bench.config5(dev)}), flush=True)
