cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or bf16x3 or weight" 2>&1 | tail -3 ) > gpurun_out/r03ao_wgrad_tests.log 2>&1
cat gpurun_out/r03ao_wgrad_tests.log
( echo "== TNR_MMA=bf16x3 (+ counted tile coordinates, barrier in front of the last pair, next tile prefetched behind it)"; TNR_MMA=bf16x3 timeout 200 python tools/microbench_wgrad.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03ao_microbench_wgrad.txt 2>&1
grep -E " 16 | 64 |fit" gpurun_out/r03ao_microbench_wgrad.txt
