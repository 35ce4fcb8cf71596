#!/bin/bash
# tools/rehearse_multirank.sh N [bench args...]  -- ON A BOX WITH ONE GPU: runs `bench.py --gpus N` the way the driver launches it (one process per
# rank under torch.distributed.run), with every rank on GPU 0: gloo for the process group, tests/support/loopback_rccl.cpp (named pipes + host
# copies) in place of librccl under rtw_gather_rows (RCCL refuses two ranks on one device).  It checks the code path of the N > 1 bench --
# communicator bootstrap, every rank's share of the K passes, the gather, the comparison of the gathered image with the one-GPU replay --
# not its speed.  N <= 4 (a GPU box allows few processes on its card).
set -e
N=${1:-2}; shift || true
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TMP=$(mktemp -d)
g++ -O1 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $ROOTDIR/tests/support/loopback_rccl.cpp -o $TMP/libloopback_rccl.so -L/opt/rocm/lib -lamdhip64
export RTW_RCCL_LIBRARY=$TMP/libloopback_rccl.so RTW_LOOPBACK_DIR=$TMP RTW_BENCH_REHEARSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) $ROOTDIR/bench.py --gpus $N --no-cpu "$@"
rm -rf $TMP
