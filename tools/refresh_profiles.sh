set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
python bench.py --steps 20 > gpurun_out/r01b/bench_headline_n1.json 2> gpurun_out/r01b/bench_headline.log
tail -c 600 gpurun_out/r01b/bench_headline_n1.json
for w in c2 c4 c5; do python bench.py --workload $w --steps 5 --no-cpu-baseline 2>/dev/null > gpurun_out/r01b/bench_$w.json; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/ktrace -o kt -- python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/r01b/ktrace.log 2>&1
ls gpurun_out/r01b/ktrace
bash tools/pmc_profile.sh r01b_pmc
python tools/bench_msa.py 128 300 > gpurun_out/r01b/msa_128.txt 2>&1
python tools/bench_msa.py 512 300 > gpurun_out/r01b/msa_512.txt 2>&1
python tools/host_overheads.py 128 300 > gpurun_out/r01b/host_128.txt 2>&1
tail -n 2 gpurun_out/r01b/msa_128.txt gpurun_out/r01b/msa_512.txt
