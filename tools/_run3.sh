O=gpurun_out/r05c; mkdir -p $O
python tools/long_share_layouts.py 2 4 8 > $O/long_share_c5.txt 2>&1
python tools/long_share_layouts.py 64,900,77 2 4 8 > $O/long_share_900.txt 2>&1
python tools/long_share_layouts.py 64,600,78 2 4 > $O/long_share_600.txt 2>&1
for kb in 0 33 41 54; do echo "CARETTA_MID_LDS_KB=$kb"; CARETTA_MID_LDS_KB=$kb python tools/c3_share_time.py 2>/dev/null | tail -1; done > $O/c3_lds_cap.txt 2>&1
cat $O/*.txt | grep -v amdgpu
