#!/bin/bash
# The ONE runner for everything that goes to the GPU box (gpurun -- 'bash tools/gpu.sh <cmd> <tag> ...'); logs land in gpurun_out/<tag>_*.
#   suite <tag> [pytest args]        the -m gpu suite (both arithmetic modes, tests/conftest.py) + smoke
#   prof  <tag> <bf16x3|f32> [bench args]   rocprofv3 kernel stats of 3 bench steps + separate PMC passes (HBM traffic, MFMA busy) of 1
#   prof_i2i <tag> <pix2pix|cyclegan> <resnet|unet>   the PMC passes of one step of tools/bench_i2i.py
#   final <tag>                      suite + default bench line (with cpu_baseline) + prof bf16x3: the validation of a tree
#   power <tag> <name> <cmd...>      the command with rocm-smi sampled every 0.5 s next to it (clock, socket power, temperature)
#   run   <tag> <name> <cmd...>      any command, output to gpurun_out/<tag>_<name>.txt (tail printed)
#   ab    <tag> <name> <v1,v2,..> <cmd...>  the command once per kernel-build variant (tools/build_variant.py; "" = the shipped library)
set -u
CMD=${1:?cmd}; TAG=${2:?tag}; shift 2
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R

suite() {
  export TNR_REQUIRE_HEADLINE=1      # the batch-16 128 -> 512 parity test FAILS instead of skipping when the box lacks host memory for its oracle
  ( time timeout 1800 python -m pytest tests -m gpu -q "$@" ) > $O/${TAG}_pytest_gpu.log 2>&1
  tail -5 $O/${TAG}_pytest_gpu.log
  grep -n "^E  \|^FAILED\|gate-pinned" $O/${TAG}_pytest_gpu.log | head -40
  ( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${TAG}_smoke.log 2>&1
  grep "smoke" $O/${TAG}_smoke.log | cut -c1-200
}

prof() {
  local MODE=${1:-bf16x3}; shift || true
  local ARGS="--mma $MODE --no-cpu-baseline --no-roofline --no-variant $*"
  local SFX=${MODE}$(echo "$*" | tr -d ' -')
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof_$TAG
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py $ARGS --steps 3 --warmup 1 > $O/${TAG}_prof_${SFX}.log 2>&1
  cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_bench_steps3_${SFX}.csv
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    local N=$(echo $C | cut -d' ' -f1)
    rm -rf /tmp/pmc_$N
    timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$N -- python $R/bench.py $ARGS --steps 1 --warmup 1 > $O/${TAG}_pmc_run_$N.log 2>&1
    python $R/tools/pmc_summary.py /tmp/pmc_$N > $O/${TAG}_pmc_${N}_${SFX}.summary.csv
  done
  python $R/tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE_${SFX}.summary.csv $O/${TAG}_pmc_WRITE_SIZE_${SFX}.summary.csv $O/${TAG}_pmc_traffic_${SFX}.json
  python $R/tools/pmc_busy.py $O/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES_${SFX}.summary.csv $O/${TAG}_pmc_mfma_busy_${SFX}.json
  python $R/tools/pmc_stamp.py $O/${TAG}_pmc_traffic_${SFX}.json $O/${TAG}_pmc_mfma_busy_${SFX}.json
  head -14 $O/${TAG}_kernel_stats_bench_steps3_${SFX}.csv | cut -c1-150
  cd $R
}

# PMC passes (HBM traffic, MFMA busy) of one step of tools/bench_i2i.py: records *_pmc_{traffic,mfma_busy}_i2i_<model>_<netg>_bf16x3.json
prof_i2i() {
  local MODEL=${1:-pix2pix} NETG=${2:-resnet}
  local SFX=i2i_${MODEL}_${NETG}_bf16x3
  cd /tmp && export TMPDIR=/tmp
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    local N=$(echo $C | cut -d' ' -f1)
    rm -rf /tmp/pmc_$N
    timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$N -- python $R/tools/bench_i2i.py --model $MODEL --netg $NETG --steps 1 --warmup 1 > $O/${TAG}_pmc_run_${N}_${SFX}.log 2>&1
    python $R/tools/pmc_summary.py /tmp/pmc_$N > $O/${TAG}_pmc_${N}_${SFX}.summary.csv
  done
  python $R/tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE_${SFX}.summary.csv $O/${TAG}_pmc_WRITE_SIZE_${SFX}.summary.csv $O/${TAG}_pmc_traffic_${SFX}.json
  python $R/tools/pmc_busy.py $O/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES_${SFX}.summary.csv $O/${TAG}_pmc_mfma_busy_${SFX}.json
  python $R/tools/pmc_stamp.py $O/${TAG}_pmc_traffic_${SFX}.json $O/${TAG}_pmc_mfma_busy_${SFX}.json
  cd $R
}

case $CMD in
  suite) suite "$@" ;;
  prof) prof "$@" ;;
  prof_i2i) prof_i2i "$@" ;;
  final)
    suite
    ( time timeout 900 python bench.py ) > $O/${TAG}_bench_default.json.log 2> $O/${TAG}_bench_default.err
    tail -1 $O/${TAG}_bench_default.json.log | cut -c1-1500
    prof bf16x3 > $O/${TAG}_prof_summary.txt 2>&1
    tail -4 $O/${TAG}_prof_summary.txt | cut -c1-600 ;;
  power)
    NAME=${1:?name}; shift
    ( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Socket Graphics Package Power|junction" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/${TAG}_${NAME}_smi.txt &
    SMI=$!
    ( time "$@" ) > $O/${TAG}_${NAME}.txt 2>&1
    kill $SMI
    grep -v amdgpu.ids $O/${TAG}_${NAME}.txt | tail -3 | cut -c1-300
    awk 'NR%4==0' $O/${TAG}_${NAME}_smi.txt | cut -c1-260 | tail -${TAIL:-25} ;;
  run)
    NAME=${1:?name}; shift
    ( time "$@" ) > $O/${TAG}_${NAME}.txt 2>&1
    grep -v amdgpu.ids $O/${TAG}_${NAME}.txt | tail -${TAIL:-40} | cut -c1-400 ;;
  ab)
    NAME=${1:?name}; VARS=${2?variants}; shift 2
    IFS=',' read -ra VL <<< "$VARS"
    for v in "${VL[@]}"; do
      if [ -n "$v" ]; then export TNR_HIP_LIB=$R/trainner_amd/lib/variants/lib$v.so; else unset TNR_HIP_LIB; fi
      echo "=== variant: ${v:-shipped}" >> $O/${TAG}_${NAME}.txt
      ( "$@" 2>&1 | grep -v amdgpu.ids ) >> $O/${TAG}_${NAME}.txt 2>&1
    done
    unset TNR_HIP_LIB
    tail -${TAIL:-60} $O/${TAG}_${NAME}.txt | cut -c1-300 ;;
  *) echo "unknown command $CMD"; exit 2 ;;
esac
