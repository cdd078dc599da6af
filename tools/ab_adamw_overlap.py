#!/usr/bin/env python3
"""Same-box A/B of the overlapped optimizer (segmentation._ADAMW_OVERLAP: AdamW per gradient range on a side stream during backward) against
one AdamW launch behind the backward pass.  usage: python tools/ab_adamw_overlap.py 0|1 [bench.py args]   (one arm per process)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd")]
import instageo_amd.segmentation as seg  # noqa: E402

seg._ADAMW_OVERLAP = sys.argv[1] == "1"
import bench  # noqa: E402

sys.argv = ["bench.py", "--steps", "60", "--warmup", "10", "--no-parity-leg", "--no-tile", "--no-yaml-legs", "--no-cpu-baseline", "--no-profile",
            "--detail-file", "/tmp/ab_adamw.json"] + sys.argv[2:]
bench.main()
