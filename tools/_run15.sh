O=gpurun_out/r05m; mkdir -p $O
timeout 2300 python tests/fuzz_parity.py 2100 6002 > $O/fuzz_parity_b.txt 2>&1; echo "rc $?" >> $O/fuzz_parity_b.txt
tail -3 $O/fuzz_parity_b.txt
