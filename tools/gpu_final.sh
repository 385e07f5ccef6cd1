# last validation of the round: the whole GPU suite on the default (fp32 matrix core) configuration + smoke
TAG=${1:-r02h}
cd /root/repo; mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/${TAG}_smoke.log 2>&1
grep smoke gpurun_out/${TAG}_smoke.log | cut -c1-200
