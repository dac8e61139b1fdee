set -x
mkdir -p gpurun_out/r05a
python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/pytest.log
tail -5 gpurun_out/r05a/pytest.log
C3_STAGES=1 python tools/c3_share.py 8 4 > gpurun_out/r05a/c3_stages.txt 2>&1
python tools/c3_share.py 2 > gpurun_out/r05a/c3_share_2.txt 2>&1
python tools/c5_share_layouts.py > gpurun_out/r05a/c5_layouts.txt 2>&1
python tools/c5_share_layouts.py 41,800,7,4 >> gpurun_out/r05a/c5_layouts.txt 2>&1
python tools/c5_share_layouts.py 46,600,8,5 >> gpurun_out/r05a/c5_layouts.txt 2>&1
python tools/ragged_time.py --gate > gpurun_out/r05a/ragged.txt 2>&1
tail -3 gpurun_out/r05a/*.txt
