#!/bin/bash
# rocprofv3 kernel stats of the default bench command (hipGraph replay).  usage: gpu_prof.sh <tag>
TAG=${1:-prof}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/rocprof.log 2>&1
cd $ROOT
tail -1 $OUT/rocprof.log | cut -c1-150
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith(("void fo1","fo1::")))
print("fo1 kernel time total ms:", tot/1e6)
for r in rows[:28]:
    if r["Name"].startswith(("void fo1","fo1::")):
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
