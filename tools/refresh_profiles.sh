#!/bin/bash
# Regenerate the measured files of profiles/<round> on the GPU box:  gpurun -- 'bash tools/refresh_profiles.sh r04'
# (writes under gpurun_out/<round>/; copy what is to be judged into profiles/<round>/).
R=${1:-r04}
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python bench.py --steps 20 > $O/bench_headline_n1.json 2> $O/bench_headline.log
tail -c 400 $O/bench_headline_n1.json
for w in c2 c4 c5; do python bench.py --workload $w --steps 5 --no-cpu-baseline --no-extras 2>/dev/null > $O/bench_$w.json; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 > $O/ktrace.log 2>&1
ls $O/ktrace
bash tools/pmc_profile.sh ${R}_pmc --steps 3 --warmup 1 --no-cpu-baseline --no-extras
bash tools/pmc_share.sh ${R}_pmc_c5share > $O/pmc_c5share.log 2>&1
bash tools/pmc_share.sh ${R}_pmc_c3share tools/c3_share_time.py > $O/pmc_c3share.log 2>&1
CARETTA_TRIO=0 bash tools/pmc_share.sh ${R}_pmc_c3duo tools/c3_share_time.py > $O/pmc_c3duo.log 2>&1
python tools/bench_msa.py 128 300 > $O/msa_128.txt 2>&1
python tools/bench_msa.py 512 300 > $O/msa_512.txt 2>&1
python tools/config5_share_time.py > $O/config5_share.txt 2>&1
python tools/calibrate_wide.py c5share p120x900 p120x600 p105x1500 > $O/calibrate_wide.txt 2>&1
python tools/stamps.py run c5share c2 one300 tree128 > $O/stamps.txt 2>&1
CARETTA_STAGED=0 python tools/stamps.py run tree128 one300 > $O/stamps_fused.txt 2>&1
python tools/calibrate_staged.py c2 c2half one300 p64x300 p120x450 p120x600 p120x750 p28x750 p28x1000 p120x900 p28x1500 p105x1500 p6x2000 > $O/calibrate_staged.txt 2>&1
(for a in "128 100" "256 150" "128 300" "512 300" "64 600" "32 900" "16 1300"; do python tools/bench_msa.py $a | tail -1; CARETTA_STAGED=0 python tools/bench_msa.py $a | tail -1 | sed "s/^/   CARETTA_STAGED=0: /"; done) > $O/msa_sizes.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/msa_ktrace -o kt -- python3 tools/bench_msa.py 128 300 > $O/msa_ktrace.log 2>&1
# (the microbenchmarks are built here: their binaries are not part of the tree)
for t in valu_latency dpp_latency wave_placement; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/$t.hip -o tools/$t.bin 2>/dev/null; done
./tools/valu_latency.bin > $O/valu_latency.txt 2>&1
./tools/dpp_latency.bin > $O/dpp_latency.txt 2>&1
(./tools/wave_placement.bin 1016 128 17; ./tools/wave_placement.bin 1016 64 10; ./tools/wave_placement.bin 2032 64 10; ./tools/wave_placement.bin 1016 320 36) > $O/wave_placement.txt 2>&1
python tools/c3_share.py 16 8 4 > $O/c3_share.txt 2>&1
C3_LIMIT=1 python tools/c3_share.py 14 12 10 9 8 7 6 5 > $O/c3_share_limit.txt 2>&1
(C3_LIMIT=1 python tools/c3_share.py --family=128,360,14 8 16; C3_LIMIT=1 python tools/c3_share.py --family=128,450,11 8 16; C3_LIMIT=1 python tools/c3_share.py --family=128,600,12 8 16) > $O/c3_share_lengths.txt 2>&1
(export C3_TRIO=1; for f in 32,150 46,150 32,220 46,220 32,100 40,100 28,128 56,150; do python tools/c3_share.py --family=$f,20241 1; done) > $O/trio_sizes.txt 2>&1
(export C3_FEW=1; for f in 12,300 16,300 20,300 23,300 12,150 16,150 23,150 16,220 23,220; do python tools/c3_share.py --family=$f,20241 1; done) > $O/trio_few.txt 2>&1
python tools/c5_share_layouts.py > $O/c5_share_layouts.txt 2>&1
STAMPS_DETAIL=1 python tools/stamps.py run c3share > $O/stamps_c3share.txt 2>&1
STAMPS_DETAIL=1 CARETTA_TRIO=0 python tools/stamps.py run c3share > $O/stamps_c3share_duo.txt 2>&1
CARETTA_TRIO=0 CARETTA_MID=0 python tools/stamps.py run c3share c3quarter > $O/stamps_c3share_single.txt 2>&1
python tools/multi_gpu_check.py 512 300 2>/dev/null | grep '^{' > $O/multi_gpu_check_1device.json
python tools/dropin_latency.py > $O/dropin_latency.txt 2>&1
python tools/nj_device_time.py > $O/nj_device_time.txt 2>&1
python tools/explicit_batch_rate.py > $O/explicit_batch_rate.txt 2>&1
# HBM counters of the batched explicit-matrix row sweep (separate --pmc passes, --kernel-trace only)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/explicit_$C -- python3 tools/explicit_batch_rate.py 8128 300 > $O/explicit_$C.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/explicit_ktrace -o kt -- python3 tools/explicit_batch_rate.py 8128 300 > $O/explicit_ktrace.log 2>&1
python tools/pmc_explicit_summary.py $O > $O/explicit_batch_pmc.json
rocprofv3 --kernel-trace --output-format csv -d $O/msa_trace -o kt -- python3 tools/bench_msa.py 128 300 > $O/msa_trace.log 2>&1
python tools/trace_gaps.py $O/msa_trace > $O/msa_gaps.txt 2>&1
tail -n 2 $O/msa_128.txt $O/msa_512.txt $O/config5_share.txt $O/explicit_batch_rate.txt
