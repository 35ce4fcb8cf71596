#!/bin/bash
# tools/profile_round.sh TAG CONFIG [bench args...]  -- run ON THE GPU BOX (gpurun): rocprofv3 runs of `python3 bench.py --config CONFIG --no-cpu ...`
#   1. --kernel-trace --stats                      -> per-kernel durations
#   2. --pmc FETCH_SIZE      (own pass)            -> HBM read-side bytes per kernel
#   3. --pmc WRITE_SIZE      (own pass)            -> HBM write-side bytes per kernel
#   4. --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES  -> instructions issued per kernel (VALU instructions per secondary ray)
# Counters are collected in their own runs with --kernel-trace only (gpurun refuses --pmc together with the sys / hip / hsa trace domains).
# Raw output under gpurun_out/prof_TAG/; tools/profile_summary.py turns it into the small files that are committed under profiles/.
set -e
TAG=$1; CFG=$2; shift 2
ROOTDIR=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS="--steps 20 --warmup 5"
run() {  # name, rocprof args...
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python3 $ROOTDIR/bench.py --config $CFG --no-cpu $STEPS "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1
    grep -h '^{"metric"' $OUT/$name.log | tail -1 > $OUT/$name.bench.json || true
}
BENCH_ARGS=("$@")
run kt --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run insts --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
python3 $ROOTDIR/tools/profile_summary.py $OUT $TAG $CFG > $OUT/summary.log 2>&1 || true
# keep the merged-back directory small: the raw per-dispatch traces can be tens of MB
find $OUT -name "*.csv" -size +8M -delete
tail -5 $OUT/summary.log
