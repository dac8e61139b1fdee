#!/bin/bash
# Regenerate the measured files of profiles/<round> on the GPU box:  gpurun -- 'bash tools/refresh_profiles.sh r06'
# (writes under gpurun_out/<round>/; copy what is to be judged into profiles/<round>/).  Every rocprofv3 run has the program
# itself behind `--` (python3 <script>); counters are collected in their own passes (--kernel-trace only).
R=${1:-r06}
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 20 > $O/bench_headline_n1.json 2> $O/bench_headline.log
tail -c 300 $O/bench_headline_n1.json
for w in c2 c4 c5; do python bench.py --workload $w --steps 5 --repeats 3 --no-cpu-baseline --no-extras 2>/dev/null > $O/bench_$w.json; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --repeats 3 > $O/ktrace.log 2>&1
ls $O/ktrace
bash tools/pmc_profile.sh ${R}_pmc --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras
bash tools/pmc_share.sh ${R}_pmc_c3share tools/c3_share_time.py > $O/pmc_c3share.log 2>&1
bash tools/pmc_share.sh ${R}_pmc_c5share > $O/pmc_c5share.log 2>&1
python tools/bench_msa.py 128 300 > $O/msa_128.txt 2>&1
python tools/bench_msa.py 512 300 > $O/msa_512.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/msa_ktrace -o kt -- python3 tools/bench_msa.py 128 300 > $O/msa_ktrace.log 2>&1
python tools/ragged_time.py --gate > $O/ragged.txt 2>&1
python tools/c3_share.py 16 8 4 2 > $O/c3_share.txt 2>&1
C3_STAGES=1 python tools/c3_share.py 8 > $O/c3_stages.txt 2>&1
python tools/long_share_layouts.py 2 4 8 > $O/long_share_layouts.txt 2>&1
python tools/long_share_layouts.py 64,900,77 2 4 8 >> $O/long_share_layouts.txt 2>&1
python tools/multi_gpu_check.py 512 300 2>/dev/null | grep '^{' > $O/multi_gpu_check_1device.json
python tools/dropin_latency.py > $O/dropin_latency.txt 2>&1
python tools/explicit_batch_rate.py > $O/explicit_batch_rate.txt 2>&1
python tools/sw_rows_probe.py > $O/sw_rows_probe.txt 2>&1
if [ -x tools/sstore_rate.bin ]; then ./tools/sstore_rate.bin > $O/sstore_rate.txt 2>&1; fi
# every branch of an N-rank bench line on this one GPU (RCCL process group with one rank, the N-ranks-against-one-GPU gates)
CARETTA_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 10 --repeats 3 > $O/bench_forced_dist.json 2> $O/bench_forced_dist.log
# HBM counters of the batched explicit-matrix kernels (separate --pmc passes, --kernel-trace only)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/explicit_$C -- python3 tools/explicit_batch_rate.py 8128 300 > $O/explicit_$C.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/explicit_ktrace -o kt -- python3 tools/explicit_batch_rate.py 8128 300 > $O/explicit_ktrace.log 2>&1
python tools/pmc_explicit_summary.py $O > $O/explicit_batch_pmc.json
# the staged sweeps of a tree level in one workgroup (binaries built by `bash tools/step_probe.sh build` before the snapshot), short lists
# on staged scores against the split by function
if [ -x tools/step_probe_lib.bin ]; then bash tools/step_probe.sh run > $O/step_probe_run.txt 2>&1; fi
python tools/staged_vs_trio.py > $O/staged_vs_trio.txt 2>&1
STAMPS_DETAIL=1 python tools/stamps.py run c3share > $O/stamps_c3share.txt 2>&1
python tools/stamps.py run c5share tree128 > $O/stamps.txt 2>&1
tail -n 2 $O/msa_128.txt $O/msa_512.txt $O/explicit_batch_rate.txt
find $O -name "*stats*csv" | head
