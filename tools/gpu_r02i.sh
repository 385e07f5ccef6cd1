cd /root/repo
bash tools/gpu_variants.sh noperm
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x 2>&1 | tail -3
