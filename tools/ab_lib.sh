cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp bx-python_amd/bxmi/libbxmi.so /tmp/lib_default.so
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so bx-python_amd/bxmi/libbxmi.so; else cp build_variants/libbxmi_$v.so bx-python_amd/bxmi/libbxmi.so; fi
  echo -n "$v  "; REPS=10 python tools/count_only.py 2>&1 | tail -1
done
