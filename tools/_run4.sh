O=gpurun_out/r05d; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "flexible or size_classes or path_selection or midsize or streamed" > $O/pytest_sel.log 2>&1; echo "rc $?" >> $O/pytest_sel.log
tail -15 $O/pytest_sel.log
