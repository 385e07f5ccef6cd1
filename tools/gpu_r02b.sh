# r02b GPU check: new i2i tests first, then the whole GPU suite, the i2i bench variants, a short headline bench
cd /root/repo
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_i2i.py tests/test_gpu_kernels.py -x -q -k "i2i or ganloss" ) > gpurun_out/r02b_i2i_tests.log 2>&1
tail -5 gpurun_out/r02b_i2i_tests.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r02b_pytest_gpu.log 2>&1
tail -5 gpurun_out/r02b_pytest_gpu.log
timeout 300 python tools/bench_i2i.py --model pix2pix > gpurun_out/r02b_variant_bench_pix2pix.json.log 2> gpurun_out/r02b_variant_bench_pix2pix.err
timeout 300 python tools/bench_i2i.py --model cyclegan --batch 8 > gpurun_out/r02b_variant_bench_cyclegan.json.log 2> gpurun_out/r02b_variant_bench_cyclegan.err
tail -2 gpurun_out/r02b_variant_bench_pix2pix.json.log gpurun_out/r02b_variant_bench_cyclegan.json.log | cut -c1-900
tail -3 gpurun_out/r02b_variant_bench_pix2pix.err gpurun_out/r02b_variant_bench_cyclegan.err
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r02b_bench_q.log 2> gpurun_out/r02b_bench_q.err
tail -1 gpurun_out/r02b_bench_q.log | cut -c1-600
