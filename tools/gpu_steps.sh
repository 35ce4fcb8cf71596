#!/bin/bash
# tools/gpu_steps.sh LOGNAME:SECONDS:COMMAND ...  -- ON THE GPU BOX: runs the steps one after another, each under `timeout -k 10`, its output in
# gpurun_out/LOGNAME.log.  A step that merely FAILS (a test assertion) does not stop the later ones; a step that was killed or timed out (rc 124 / 137 /
# 139 / 134) does: after a hung or faulted GPU step nothing further is started in the same call.
mkdir -p gpurun_out
status=0
for step in "$@"; do
    name=${step%%:*}; rest=${step#*:}; secs=${rest%%:*}; cmd=${rest#*:}
    echo "== $name: $cmd" | tee gpurun_out/$name.log
    timeout -k 10 $secs bash -c "$cmd" >> gpurun_out/$name.log 2>&1
    rc=$?
    echo "== $name rc $rc"; echo "== rc $rc" >> gpurun_out/$name.log
    tail -3 gpurun_out/$name.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ] || [ $rc -eq 134 ]; then echo "== $name was killed: stopping"; exit $rc; fi
    [ $rc -ne 0 ] && status=$rc
done
exit $status
