cd /root/repo
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_i2i.py tests/test_gpu_kernels.py tests/test_gpu_nets.py -x -q -k "i2i or conv4x4_s1 or patchgan or resnet_generator" ) > gpurun_out/r02d_i2i_tests.log 2>&1
tail -5 gpurun_out/r02d_i2i_tests.log
bash tools/gpu_prof_i2i.sh pix2pix 16
timeout 300 python tools/bench_i2i.py --model cyclegan --batch 8 --steps 4 2>/dev/null | tail -1 | cut -c1-330
