# Copy what `bash tools/refresh_profiles.sh <round>` left under gpurun_out/ into profiles/<round> (the files that are judged):
#   bash tools/copy_profiles.sh r06
# Everything is copied into a temporary directory first and moved into place only when every REQUIRED input was there, so a
# missing file never leaves profiles/<round> mixed between two runs (files marked optional below are skipped when absent).
set -e
cd "$(dirname "$0")/.."
R=${1:-r06}
O=gpurun_out/$R; FINAL=profiles/$R
P=$(mktemp -d gpurun_out/.copy_profiles.XXXXXX)
trap 'rm -rf "$P"' EXIT
opt() { if [ -f "$1" ]; then cp "$1" "$2"; else echo "(optional input missing: $1)"; fi; }
strip() { grep -v "amdgpu.ids" "$1" > "$2"; }
for f in bench_headline_n1.json bench_c2.json bench_c4.json bench_c5.json; do tail -n 1 $O/$f > $P/$f; done
cp $O/ktrace/kt_kernel_stats.csv $P/rocprofv3_kernel_stats.csv
cp $O/msa_ktrace/kt_kernel_stats.csv $P/rocprofv3_msa_kernel_stats.csv
cp $O/explicit_ktrace/kt_kernel_stats.csv $P/rocprofv3_explicit_batch_kernel_stats.csv
cp gpurun_out/${R}_pmc/summary.json $P/pmc_summary.json
opt gpurun_out/${R}_pmc_c3share/summary.json $P/pmc_c3share.json
opt gpurun_out/${R}_pmc_c5share/summary.json $P/pmc_c5share.json
opt $O/explicit_batch_pmc.json $P/explicit_batch_pmc.json
cp $O/multi_gpu_check_1device.json $P/multi_gpu_check_1device.json
for f in msa_128.txt msa_512.txt ragged.txt c3_share.txt c3_stages.txt long_share_layouts.txt dropin_latency.txt explicit_batch_rate.txt stamps.txt stamps_c3share.txt sw_rows_probe.txt; do strip $O/$f $P/$f; done
opt $O/sstore_rate.txt $P/sstore_rate.txt
if [ -f $O/bench_forced_dist.json ]; then tail -n 1 $O/bench_forced_dist.json > $P/bench_forced_dist.json; fi
if [ -f $O/staged_vs_trio.txt ]; then strip $O/staged_vs_trio.txt $P/staged_vs_trio.txt; fi
mkdir -p $FINAL
cp $P/* $FINAL/
python tools/pmc_traffic.py gpurun_out/${R}_pmc/summary.json profiles/pmc_traffic.json headline 1
git status --short | head -30
