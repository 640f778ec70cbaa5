#!/usr/bin/env python3
"""bench.py's `extra.loss_side` pass on its own (pyramid, RGB warps, gradOutput from the concat buffer's gradient):
    python tools/quick_loss_side.py [pairs] [width] [height]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    height = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    res = bench.loss_side_times(pairs, width, height, torch.device("cuda", 0))
    for k, v in res.items():
        print(k, json.dumps(v))
