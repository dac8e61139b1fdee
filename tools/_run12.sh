O=gpurun_out/r05k; mkdir -p $O
for i in 1 2; do
for lib in main stamps; do for w in 3 4; do
  if [ $lib = main ]; then unset CARETTA_HIP_LIB; else export CARETTA_HIP_LIB=$PWD/caretta_amd/csrc/libcaretta_hip_stamps.so; fi
  echo "[$lib waves=$w] $(CARETTA_TRIO_WAVES=$w python tools/c3_share_time.py 2>/dev/null | tail -1)"
done; done; done > $O/trio_lib_compare.txt 2>&1
cat $O/trio_lib_compare.txt
