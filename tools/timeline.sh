#!/bin/bash
# tools/timeline.sh NAME [bench args]  -- ON THE GPU BOX: rocprofv3 --kernel-trace of `bench.py --no-cpu --steps 20 --warmup 5 [args]`, then the dispatches of
# the TIMED rtw_render_passes call in start order: start (us after the call's first kernel), duration, grid, kernel.
set -e
NAME=$1; shift
ROOTDIR=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOTDIR/gpurun_out/tl_$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o $NAME -- python3 $ROOTDIR/bench.py --no-cpu --steps 20 --warmup 5 "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(\w+_kernel)", n); return m.group(1) if m else n[:40]
names = [short(r["Kernel_Name"]) for r in rows]
replay = [i for i, n in enumerate(names) if n == "render_kernel"]
b = replay[0] if replay else len(names)
fills = [i for i, n in enumerate(names[:b]) if "elementwise" in n]
a = fills[-1] + 1 if fills else 0
while b > a and not names[b - 1].startswith("g"): b -= 1
t0 = int(rows[a]["Start_Timestamp"])
for r, n in list(zip(rows, names))[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  +%7.1f us  grid %7s x %4s  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("Queue_Id", "?"), n))
print("span %.1f us" % ((max(int(r["End_Timestamp"]) for r in rows[a:b]) - t0) / 1e3))
PY
find $OUT -name "*.csv" -size +8M -delete
