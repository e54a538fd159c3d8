cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/kt -o c1 --output-format csv -- python $R/tools/probes/prof_conv1.py > /tmp/kt.log 2>&1
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs head -3 | tail -2 | python3 -c "
import sys,csv
for row in csv.reader(sys.stdin): print(row[0][25:80], row[1], row[3])"
