#!/bin/bash
# tools/pmc_run.sh NAME "COUNTER COUNTER ..." [bench args]  -- ON THE GPU BOX: one rocprofv3 --pmc pass (with --kernel-trace only) over
# `python3 bench.py --no-cpu --steps 20 --warmup 5 [bench args]`; prints the counters summed per kernel over the whole run.
set -e
NAME=$1; CTRS=$2; shift 2
ROOTDIR=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOTDIR/gpurun_out/pmc_$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o $NAME -- python3 $ROOTDIR/bench.py --no-cpu --steps 20 --warmup 5 "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, re, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = (m.group(1) if m else r["Kernel_Name"][:30]) + ("<stats>" if "ILb1" in r["Kernel_Name"] else "")
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(per.items()):
    print("%-34s" % k, "  ".join("%s=%.4g" % kv for kv in sorted(v.items())))
PY
find $OUT -name "*.csv" -size +8M -delete
