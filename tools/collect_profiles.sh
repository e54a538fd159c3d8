#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh <tag>'): the rocprofv3 passes behind profiles/.
#   kt        : --kernel-trace --stats of the default bench command
#   pmc_fetch / pmc_write / pmc_l2 : separate counter passes of the C2-only command (no other trace domains)
# Output: gpurun_out/<tag>/{kt,pmc_*}/bench_*.csv ; condense with  python tools/prof_summary.py gpurun_out/<tag>
TAG=${1:-r1x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$O/bench_line.json" 2> "$O/bench.err"
rocprofv3 --kernel-trace --stats -d "$O/kt" -o bench --output-format csv -- python "$R/bench.py" --no-cpu-baseline > "$O/kt.log" 2>&1
C2="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --alexnet-batch 0 --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/pmc_fetch" -o bench --output-format csv -- $C2 > "$O/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/pmc_write" -o bench --output-format csv -- $C2 > "$O/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$O/pmc_l2" -o bench --output-format csv -- $C2 > "$O/pmc_l2.log" 2>&1
# the other configs of the bench line's "extra" object, one kernel-stats pass each (C4: DoReFa ResNet-18, C5: ternary VGG-16)
C4="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --alexnet-batch 0 --c5-batch 0 --train-batch 0"
C5="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --alexnet-batch 0 --c4-batch 0 --train-batch 0"
rocprofv3 --kernel-trace --stats -d "$O/kt_c4" -o bench --output-format csv -- $C4 > "$O/kt_c4.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$O/kt_c5" -o bench --output-format csv -- $C5 > "$O/kt_c5.log" 2>&1
for s in kt_c4 kt_c5; do find "$O/$s" -mindepth 2 -name "*.csv" -exec mv {} "$O/$s/" \; ; rm -f "$O/$s/bench_kernel_trace.csv"; done
# rocprofv3 nests its output under <dir>/<hostname>/: flatten
for s in kt pmc_fetch pmc_write pmc_l2; do find "$O/$s" -mindepth 2 -name "*.csv" -exec mv {} "$O/$s/" \; ; done
# keep the merge small: the per-dispatch traces of the big run are not needed
rm -f "$O/kt/bench_kernel_trace.csv"
ls -la "$O" "$O/kt" | head -30
tail -c 600 "$O/bench_line.json"
