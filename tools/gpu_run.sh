#!/bin/bash
# One parameterised GPU-box script (replaces round 3's one-shot tools/r03_gpu*.sh): runs the steps named on the command line
# from the repo root, every log under gpurun_out/<tag>_*.  usage: bash tools/gpu_run.sh <tag> <step> [<step> ...]
#   steps:  t4 (tests/test_gpu_r4.py)  tall (all -m gpu tests)  time4 (isolated F(4x4) / F(2x2) / direct launch)
#           abl4 (ablation ladder of the F(4x4) kernel: make -C bmcnet-esr_amd/csrc libbmc_hip_w4ablN.so first)
#           bench (default bench line)  benchq (bench without side runs)  stats (rocprofv3 --kernel-trace --stats of benchq)  stats1 (the same with BMC_WGRAD_STREAM=0: every kernel alone on one stream)
#           pmc (the three PMC passes of tools/pmc_summary.py)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
TAG=$1; shift
cd $R
export TMPDIR=/tmp
for step in "$@"; do
  case $step in
    t4)    timeout 900 python -m pytest tests/test_gpu_r4.py -x -q -m gpu -s > $O/${TAG}_t4.log 2>&1; tail -5 $O/${TAG}_t4.log ;;
    tall)  timeout 3000 python -m pytest tests -x -q -m gpu > $O/${TAG}_tall.log 2>&1; tail -5 $O/${TAG}_tall.log ;;
    time4) timeout 300 python tools/time_wino4.py > $O/${TAG}_time4.log 2>&1; cat $O/${TAG}_time4.log ;;
    abl4)  for n in ${W4_ABLS:-0 2 4 8 16 32 64 128 6 22 54 118 246 254}; do
             lib=$R/bmcnet-esr_amd/csrc/libbmc_hip_w4abl$n.so; [ $n = 0 ] && lib=$R/bmcnet-esr_amd/csrc/libbmc_hip.so
             [ -f $lib ] && echo "abl $n: $(W4_ONLY=1 KB_ITERS=200 BMC_HIP_LIB=$lib timeout 120 python tools/time_wino4.py 2>&1 | grep 'F(4x4)' | tr '\n' ' ' | sed 's/algorithmic[^=]*=//g')"
           done > $O/${TAG}_abl4.log 2>&1; cat $O/${TAG}_abl4.log ;;
    abl4prof) cd /tmp; for n in ${W4_ABLS:-0 254 255}; do
             lib=$R/bmcnet-esr_amd/csrc/libbmc_hip_w4abl$n.so; [ $n = 0 ] && lib=$R/bmcnet-esr_amd/csrc/libbmc_hip.so
             for shp in ${W4_SHAPES:-8,180,240,128}; do
               IFS=, read b h w cin <<< "$shp"
               W4_ONLY=1 KB_ITERS=50 KB_CIN=$cin BMC_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w4p_${n}_$shp -o p -- python3 $R/tools/time_wino4.py $b $h $w > /dev/null 2>&1
               echo "abl $n shape $shp: $(find /tmp/w4p_${n}_$shp -name '*kernel_stats.csv' -exec grep wino4_conv {} \; | sed 's/.*ConvK)",//' | cut -d, -f1-3)"
             done
           done > $O/${TAG}_abl4prof.log 2>&1; cat $O/${TAG}_abl4prof.log; cd $R ;;
    abl4pmc) cd /tmp; for n in ${W4_ABLS:-0 254}; do
             lib=$R/bmcnet-esr_amd/csrc/libbmc_hip_w4abl$n.so; [ $n = 0 ] && lib=$R/bmcnet-esr_amd/csrc/libbmc_hip.so
             W4_ONLY=1 KB_ITERS=30 BMC_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/w4pmc_$n -o p -- python3 $R/tools/time_wino4.py ${W4_SHAPE:-8 180 240} > /dev/null 2>&1
             echo "abl $n: $(python3 $R/tools/pmc_agg.py $(find /tmp/w4pmc_$n -name '*.db' | head -1) wino4_conv)"
           done > $O/${TAG}_abl4pmc.log 2>&1; cat $O/${TAG}_abl4pmc.log; cd $R ;;
    stamp4) for n in ${W4_ABLS:-0 254}; do echo "== abl $n"; BMC_HIP_LIB=$R/bmcnet-esr_amd/csrc/libbmc_hip_w4stamp$n.so timeout 120 python tools/w4_stamps.py ${W4_SHAPE:-8 180 240} 2>&1 | grep -v amdgpu.ids; done > $O/${TAG}_stamp4.log 2>&1; cat $O/${TAG}_stamp4.log ;;
    c3trace) cd /tmp; for g in "" "--graph"; do
             rocprofv3 --kernel-trace --output-format csv -d /tmp/c3t$g -o t -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none --height 31 --width 56 --math ${C3_MATH:-bf16} $g > $O/${TAG}_c3trace$g.log 2>&1
             echo "config3 ${C3_MATH:-bf16} $g: $(grep -o '"ms_per_step": [0-9.]*' $O/${TAG}_c3trace$g.log | head -1)"
             python3 $R/tools/gpu_timeline.py $(find /tmp/c3t$g -name '*kernel_trace.csv' | head -1) 3
           done > $O/${TAG}_c3trace.log 2>&1; cat $O/${TAG}_c3trace.log; cd $R ;;
    bench) timeout 1500 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; head -c 600 $O/${TAG}_bench.json ;;
    benchq) timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none > $O/${TAG}_benchq.json 2> $O/${TAG}_benchq.err; head -c 700 $O/${TAG}_benchq.json ;;
    stats) cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16x6 --also none > $O/${TAG}_stats.log 2>&1
           find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \; ; head -12 $O/${TAG}_kernel_stats.csv; cd $R ;;
    stats1) cd /tmp; export BMC_WGRAD_STREAM=0; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$TAG -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16x6 --also none > $O/${TAG}_stats1.log 2>&1
           unset BMC_WGRAD_STREAM; find /tmp/prof1_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_onestream_kernel_stats.csv \; ; head -6 $O/${TAG}_onestream_kernel_stats.csv; cd $R ;;
    pmc)   cd /tmp
           for p in "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
             rocprofv3 --kernel-trace --pmc ${p#*:} -d /tmp/pmcw_$TAG/pmc_fp32_${p%%:*} -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-bf16x6 --also none > $O/${TAG}_pmc_${p%%:*}.log 2>&1
           done
           cd $R; python3 tools/pmc_summary.py /tmp/pmcw_$TAG ${PMC_TAG:-$TAG} > $O/${TAG}_pmc_summary.log 2>&1; cp /tmp/pmcw_$TAG/${PMC_TAG:-$TAG}_pmc_summary.json $O/ 2>/dev/null ;;
    *) echo "unknown step $step" ;;
  esac
done
