O=gpurun_out/r05e; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "explicit or dtw_align_batch or smith_waterman_batch or plugin" > $O/pytest_sel.log 2>&1; echo "rc $?" >> $O/pytest_sel.log
tail -5 $O/pytest_sel.log
python tools/explicit_batch_rate.py > $O/explicit_batch_rate.txt 2>&1
grep -v amdgpu $O/explicit_batch_rate.txt
