O=gpurun_out/r05l; mkdir -p $O
bash tools/trio_compare.sh 3 main ageprio1 ageprio2 > $O/trio_ageprio.txt 2>&1
cat $O/trio_ageprio.txt
