O=gpurun_out/r05f; mkdir -p $O
python tools/progressive_breakdown.py 128 300 > $O/progressive_breakdown.txt 2>&1
python tools/bench_msa.py 128 300 >> $O/progressive_breakdown.txt 2>&1
grep -v amdgpu $O/progressive_breakdown.txt
