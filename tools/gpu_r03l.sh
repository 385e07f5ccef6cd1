# round 3: the weight gradient with the pre-split pixel-major LDS image + transposing reads (TNR_MMA_BF16X3)
cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or bf16x3 or weight" 2>&1 | tail -8 ) > gpurun_out/r03l_wgrad_tests.log 2>&1
cat gpurun_out/r03l_wgrad_tests.log
( for m in bf16x3 f32; do echo "== TNR_MMA=$m"; TNR_MMA=$m timeout 200 python tools/microbench_wgrad.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r03l_microbench_wgrad.txt 2>&1
cat gpurun_out/r03l_microbench_wgrad.txt
