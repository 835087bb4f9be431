#!/usr/bin/env python3
"""configs[2] alone (bench.py's bitset leg): two hg19-sized dicts of bitsets, popcount / iand per chromosome and as one group
launch per genome.  For rocprofv3 (tools/prof_bits.sh): prints bench.py's `bitset` object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import torch

import bench

print(json.dumps(bench.bench_bitsets(torch, int(os.environ.get("STEPS", 10)), 2)))
