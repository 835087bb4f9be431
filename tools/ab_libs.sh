#!/bin/bash
# A/B of library builds x option sets on the count pass (tools/count_variants.py in one process per library).
# LIBS="default pf0 pf2" (build_variants/libbxmi_NAME.so; "default" = the in-tree library)  VARIANTS=...  OUT=gpurun_out/NAME  TRACE=1 adds a kernel trace cut into segments
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/${OUT:-ab}
mkdir -p $OUT
export TMPDIR=/tmp
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in ${LIBS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so; else cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so; fi
  echo "=== lib $v"
  if [ -n "${TRACE:-}" ]; then
    (cd /tmp && timeout ${TMO:-300} rocprofv3 --kernel-trace -d $OUT/tr_$v -o t --output-format csv -- python $REPO/tools/count_variants.py > $OUT/variants_$v.json 2> $OUT/variants_$v.err)
    python tools/trace_segments.py $OUT/tr_$v 20 3 > $OUT/segments_$v.txt 2>&1
    rm -rf $OUT/tr_$v
    grep -h "tile_sort\|_search\|unpermute\|segment" $OUT/segments_$v.txt | cut -c1-150
  else
    timeout ${TMO:-300} python tools/count_variants.py > $OUT/variants_$v.json 2> $OUT/variants_$v.err
  fi
  echo "rc=$?"; cut -c1-260 $OUT/variants_$v.json; tail -3 $OUT/variants_$v.err
done
cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so
