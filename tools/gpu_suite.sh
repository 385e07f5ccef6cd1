# the whole GPU suite (both matrix-core modes) + smoke; usage: bash tools/gpu_suite.sh <tag> [pytest args]
TAG=${1:-rXX}; shift
cd /root/repo; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q "$@" ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -5 gpurun_out/${TAG}_pytest_gpu.log
grep -n "^E  \|^FAILED\|gate-pinned" gpurun_out/${TAG}_pytest_gpu.log | head -40
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/${TAG}_smoke.log 2>&1
grep "smoke" gpurun_out/${TAG}_smoke.log | cut -c1-200
