O=gpurun_out/r05g; mkdir -p $O
C3_OCC=1 python tools/c3_share.py 6 5 4 3 2 > $O/c3_occupancy_split.txt 2>&1
grep -v amdgpu $O/c3_occupancy_split.txt
