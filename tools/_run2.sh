set -x
O=gpurun_out/r05b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/trio_compare.sh 4 main dynprio stealprio > $O/trio_compare.txt 2>&1
python tools/stamps.py run c3share > $O/stamps_c3share.txt 2>&1
python tools/ragged_time.py --gate > $O/ragged.txt 2>&1
python tools/explicit_batch_rate.py > $O/explicit_batch_rate.txt 2>&1
python tools/multi_gpu_check.py 128 300 > $O/multi_gpu_check_1device.json 2>&1
python bench.py > $O/bench_headline_n1.json 2> $O/bench_headline_n1.err
tail -c 600 $O/bench_headline_n1.err
cat $O/trio_compare.txt $O/explicit_batch_rate.txt
