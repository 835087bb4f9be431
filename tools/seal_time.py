#!/usr/bin/env python3
"""Wall time of bxmi_ivl_seal: the first index of the process (code objects load, nothing pooled) and later ones."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np

from bxmi import _ffi, synth
from bxmi.intervals import IntervalIndex

N = int(os.environ.get("NT", 10_000_000))
(ts, te), _ = synth.cfg2(N, 1)
out = []
dts, dte = _ffi.DeviceArray.from_numpy(ts), _ffi.DeviceArray.from_numpy(te)
for k in range(int(os.environ.get("REPS", 5))):
    ix = IntervalIndex()
    t0 = time.perf_counter()
    ix.append(ts, te)
    t1 = time.perf_counter()
    ix.seal()
    _ffi.call("bxmi_synchronize", None)
    t2 = time.perf_counter()
    out.append(dict(k=k, append_host_ms=round((t1 - t0) * 1e3, 2), seal_ms=round((t2 - t1) * 1e3, 2)))
    if k % 2 == 0:
        ix.close()  # (the next index reuses what this one gives back)
for k in range(3):
    ix = IntervalIndex()
    t0 = time.perf_counter()
    ix.append_dev(dts.ptr, dte.ptr, N)
    ix.seal()
    _ffi.call("bxmi_synchronize", None)
    out.append(dict(k="dev%d" % k, append_dev_plus_seal_ms=round((time.perf_counter() - t0) * 1e3, 2)))
    ix.close()
print(json.dumps(out))
