cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 600 python -m pytest tests/test_degrade.py tests/test_metrics.py tests/test_feed.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r03h_degrade_tests.log 2>&1
cat gpurun_out/r03h_degrade_tests.log
