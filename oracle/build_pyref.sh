#!/bin/bash
# oracle/build_pyref.sh -- build the reference's two hot-path Cython extensions
# OUT OF TREE (under $PYREF, default /tmp/bxref) so that oracle/gen_golden.py can
# import the real bx.bitset / bx.intervals.intersection and emit golden vectors.
# Build-container only: the result never enters the repo and never travels.
# Extension/source lists follow /root/reference/setup.py:67-75.
set -euo pipefail
REFERENCE=${REFERENCE:-/root/reference}
PYREF=${PYREF:-/tmp/bxref}
if [ -f "$PYREF/lib/bx/bitset.cpython-310-x86_64-linux-gnu.so" ] && [ -z "${FORCE:-}" ]; then
  echo "pyref already built at $PYREF"; exit 0
fi
rm -rf "$PYREF" && mkdir -p "$PYREF"
cp -r "$REFERENCE/lib" "$REFERENCE/src" "$PYREF/" && chmod -R u+w "$PYREF"
cd "$PYREF"
cat > build_hot.py <<'PY'
from setuptools import setup, Extension
from Cython.Build import cythonize
exts = [
    Extension("bx.bitset", ["lib/bx/bitset.pyx", "src/binBits.c", "src/kent/bits.c", "src/kent/common.c"],
              include_dirs=["src/kent", "src"]),
    Extension("bx.intervals.intersection", ["lib/bx/intervals/intersection.pyx"]),
]
setup(name="bxhot", package_dir={"": "lib"}, ext_modules=cythonize(exts, language_level=3),
      script_args=["build_ext", "--inplace"])
PY
python3 build_hot.py > build.log 2>&1 || { tail -30 build.log; exit 1; }
echo "pyref built at $PYREF (PYTHONPATH=$PYREF/lib)"
