# usage: [env ...] bash tools/runvariants.sh <variant> ...   (tools/abl/lib_<variant>.so; "base" = the library in the tree)
export SB_WARM_MATCHES=1
for v in "$@"; do
  if [ "$v" = base ]; then unset SAGE_LIB; else export SAGE_LIB=tools/abl/lib_$v.so; fi
  timeout -k 10 200 python -u tools/stepbench.py > gpurun_out/var_$v.log 2>&1 || exit 1
  echo "$v: $(tail -n 1 gpurun_out/var_$v.log | grep -oE "'score': [0-9.]+, 'misc': [0-9.]+, 'describe': [0-9.]+, 'pairs': [0-9.]+, 'logsig': [0-9.]+, 'fix': [0-9.]+")"
done
