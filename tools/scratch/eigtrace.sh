mkdir -p gpurun_out/eigtr; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fast.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/eigtr/tests.txt
for s in "2000000 200" "250000 200" "1000000 100" "200000 50" "250000 200" "200000 50"; do TRACE_PIN=1 python tools/host_trace.py $s | grep "^step\|stages"; done > gpurun_out/eigtr/marks.txt 2>&1
cat gpurun_out/eigtr/tests.txt gpurun_out/eigtr/marks.txt
