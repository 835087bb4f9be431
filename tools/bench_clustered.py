#!/usr/bin/env python3
"""bench.py's `clustered` leg alone (duplicate-heavy index around hot spots); BXMI_OPTS selects the stage (e.g. ivl.bm_hard_ppm)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import torch

import bench

print(json.dumps(bench.bench_clustered(torch, reps=int(os.environ.get("REPS", 5)))))
