O=gpurun_out/r05i; mkdir -p $O
timeout 900 python tests/fuzz_parity.py 600 6001 > $O/fuzz_parity.txt 2>&1; echo "rc $?" >> $O/fuzz_parity.txt
tail -5 $O/fuzz_parity.txt
