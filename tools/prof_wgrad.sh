#!/bin/bash
# On the GPU box: per-kernel times of the weight-gradient routes.  usage: prof_wgrad.sh <route: pm|gemm> shapes...
cd /tmp && export TMPDIR=/tmp
route=$1; shift
for s in "$@"; do
  ROUTE=$route rocprofv3 --kernel-trace --stats -d /tmp/p_$s -o x --output-format csv -- python /root/repo/tools/prof_wgrad.py $s >/dev/null 2>&1
  echo "== $s ($route)"; python /root/repo/tools/kstats.py /tmp/p_$s | grep -i "pm_\|wgrad\|gemm" 
done
