# Copy what `bash tools/refresh_profiles.sh r05` left under gpurun_out/ into profiles/r05 (the files that are judged).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r05; P=profiles/r05
strip() { grep -v "amdgpu.ids" "$1" > "$2"; }
for f in bench_headline_n1.json bench_c2.json bench_c4.json bench_c5.json; do tail -n 1 $O/$f > $P/$f; done
cp $O/ktrace/kt_kernel_stats.csv $P/rocprofv3_kernel_stats.csv
cp $O/msa_ktrace/kt_kernel_stats.csv $P/rocprofv3_msa_kernel_stats.csv
cp $O/explicit_ktrace/kt_kernel_stats.csv $P/rocprofv3_explicit_batch_kernel_stats.csv
cp gpurun_out/r05_pmc/summary.json $P/pmc_summary.json
cp gpurun_out/r05_pmc_c3share/summary.json $P/pmc_c3share.json
cp gpurun_out/r05_pmc_c5share/summary.json $P/pmc_c5share.json
cp $O/explicit_batch_pmc.json $P/explicit_batch_pmc.json
cp $O/multi_gpu_check_1device.json $P/multi_gpu_check_1device.json
for f in msa_128.txt msa_512.txt ragged.txt c3_share.txt c3_stages.txt long_share_layouts.txt dropin_latency.txt explicit_batch_rate.txt stamps.txt stamps_c3share.txt; do strip $O/$f $P/$f; done
if [ -f $O/staged_vs_trio.txt ]; then strip $O/staged_vs_trio.txt $P/staged_vs_trio.txt; fi
python tools/pmc_traffic.py gpurun_out/r05_pmc/summary.json profiles/pmc_traffic.json headline 1
git status --short | head -30
