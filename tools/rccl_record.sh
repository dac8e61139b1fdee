#!/bin/bash
# The RCCL path on a one-GPU box (gpurun -- 'bash tools/rccl_record.sh r02'): bench.py with one rank under
# torch.distributed.run, all-gather forced; then the same rank set-up by environment variables directly under
# rocprofv3 (no launcher between the profiler and python) for the kernel statistics that show the RCCL kernel.
R=${1:-r02}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
export CARETTA_FORCE_DIST=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 1 --no-cpu-baseline --no-extras > $O/bench_rccl_forced_n1.json 2> $O/bench_rccl_forced_n1.log
tail -c 300 $O/bench_rccl_forced_n1.json
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29552 TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rccl_ktrace -o kt -- python3 bench.py --gpus 1 --no-cpu-baseline --no-extras > $O/rccl_ktrace.log 2>&1
head -8 $O/rccl_ktrace/kt_kernel_stats.csv | cut -c1-160
