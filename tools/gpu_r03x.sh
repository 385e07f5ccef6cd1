# round 3: bf16x3 weight gradient, pipelined one-workgroup form (TNR_WG_X3_OCC=3, default) against the two-workgroup form (=2)
cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or bf16x3 or weight" 2>&1 | tail -8 ) > gpurun_out/r03x_wgrad_tests.log 2>&1
cat gpurun_out/r03x_wgrad_tests.log
( for o in 3 2; do echo "== TNR_MMA=bf16x3 TNR_WG_X3_OCC=$o"; TNR_WG_X3_OCC=$o TNR_MMA=bf16x3 timeout 200 python tools/microbench_wgrad.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r03x_microbench_wgrad.txt 2>&1
grep -v "^   [0-9 ]* [0-9]*  *[48] " gpurun_out/r03x_microbench_wgrad.txt | grep -v "   16 "
