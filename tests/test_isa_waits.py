"""What round 6 read off the compiled code (HISTORY.md section 11), pinned: the hot kernels issue no FLAT instructions where their
pointers come out of segment tables (as_global, common.hpp), and the two software-pipelined fills wait for COUNTED numbers of
outstanding memory operations -- `vmcnt(10)` / `vmcnt(6)` -- where a store loop of unknown length or a copied register makes the
compiler drain everything.  Compiles the two translation units to assembly (hipcc cross-compiles without a GPU, ~1 minute)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")


def _asm(tmp_path_factory, unit):
    out = tmp_path_factory.mktemp("isa") / (unit + ".s")
    src = os.path.join(ROOT, "bx-python_amd", "csrc", unit + ".hip")
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", str(out), src],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    import scan_isa

    return scan_isa.scan(str(out))


@pytest.fixture(scope="module")
def intervals_isa(tmp_path_factory):
    return _asm(tmp_path_factory, "intervals")


@pytest.fixture(scope="module")
def bitset_isa(tmp_path_factory):
    return _asm(tmp_path_factory, "bitset")


def _kernels(isa, prefix):
    got = {k: v for k, v in isa.items() if k.startswith("bxmi::" + prefix)}
    assert got, prefix
    return got


def test_no_flat_accesses_where_pointers_come_out_of_tables(intervals_isa, bitset_isa):
    # (kernels whose escape arms still walk the sealed index through generic pointers -- the un-permute kernels, the walks -- are
    # not listed: a FLAT access there waits for itself only)
    for prefix in ("bm_tile_sort_kernel", "part_fill_pipe_kernel", "fx_fill_kernel", "ivl_local_count_kernel", "lf_offsets_kernel", "bd_transpose_kernel",
                   "bd_plan_kernel"):
        for name, s in _kernels(intervals_isa, prefix).items():
            assert s["flat"] == 0, (name, s["flat"])
    for name, s in _kernels(bitset_isa, "bits_group_kernel").items():
        assert s["flat"] == 0, (name, s["flat"])


def test_the_tile_sort_starts_on_its_first_load(intervals_isa):
    """16 query loads of 16 bytes per thread, the first LDS atomic behind `vmcnt(15)`: as FLAT loads they were all waited for."""
    for name, s in _kernels(intervals_isa, "bm_tile_sort_kernel<1024, 32").items():
        assert 15 in s["waits"], (name, sorted(s["waits"]))


def test_the_fills_wait_for_counted_operations(intervals_isa):
    (name, s), = _kernels(intervals_isa, "part_fill_pipe_kernel").items()
    # the next batch's count and `hi` land behind exactly eight stores and the two or three younger loads of their own request, in
    # both halves of the loop unrolled by two; the staged pairs (older) behind fourteen and more
    assert sum(v for k, v in s["waits"].items() if 10 <= k <= 12) >= 3 and s["vgpr"] <= 80, (name, s["waits"], s["vgpr"])
    (name, s), = _kernels(intervals_isa, "fx_fill_kernel").items()
    # a pass's records land behind exactly three 16-byte and three 4-byte stores -- in both halves of the loop unrolled by two
    assert s["waits"].get(6, 0) >= 2, (name, s["waits"])
