"""csrc/bedparse.cpp reads text files a user hands in: its two parsers run here under AddressSanitizer + UBSan
(host-only build of the same source with the ROCm clang, tests/san/), on hostile input -- random bytes, NULs, CR, huge and
malformed integers, missing columns, buffers without a terminator -- with every view they return walked to its end.
Plain-BED results are checked against the per-line model of tests/test_host_logic.py.  No GPU involved."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def test_text_parsers_are_clean_under_asan_and_ubsan(tmp_path):
    if not os.path.exists(CLANG):
        pytest.skip("no ROCm clang here")
    asan = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(asan):
        pytest.skip("no shared ASan runtime in this toolchain")
    lib = str(tmp_path / "libbedparse_san.so")
    csrc = os.path.join(ROOT, "bx-python_amd", "csrc")
    subprocess.check_call([CLANG, "-x", "hip", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-shared-libsan", "-I" + os.path.join(ROOT, "include"),
                           "-I" + csrc, os.path.join(csrc, "bedparse.cpp"), os.path.join(ROOT, "tests", "san", "stub.cpp"), "-o", lib])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for seed in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san", "fuzz_parser.py"), lib, str(seed), "1500"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "parser fuzz: 1500 inputs" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
