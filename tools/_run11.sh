O=gpurun_out/r05j; mkdir -p $O
STAMPS_PLACEMENT=1 STAMPS_DETAIL=1 python tools/stamps.py run c3share > $O/stamps_c3share_placement.txt 2>&1
STAMPS_PLACEMENT=1 CARETTA_TRIO_WAVES=4 python tools/stamps.py run c3share >> $O/stamps_c3share_placement.txt 2>&1
grep -v amdgpu $O/stamps_c3share_placement.txt
