cd $GRAFT_REPO_ROOT; L=$PWD/bmcnet-esr_amd/csrc; O=$PWD/gpurun_out; mkdir -p $O
{
for r in 1 2; do for s in hip hip_c1pepf0; do echo "== $r $s"; BMC_HIP_LIB=$L/libbmc_$s.so timeout 300 python tools/kbench.py conv1x256 conv1x256res conv1 apply 2>&1 | grep -v amdgpu.ids; done; done
timeout 900 python -m pytest tests/test_gpu_r2.py tests/test_gpu_parity.py -x -q -m gpu -k "bie or conv1 or golden or attn or chain" 2>&1 | tail -4
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none > $O/r06h_benchq.json 2> $O/r06h_benchq.err; head -c 420 $O/r06h_benchq.json; echo
BMC_HIP_LIB=$L/libbmc_hip_c1pepf0.so timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none > $O/r06h_benchq0.json 2> $O/r06h_benchq0.err; head -c 420 $O/r06h_benchq0.json; echo
timeout 600 python -m pytest tests/test_gpu_r6.py -x -q -m gpu -k launcher 2>&1 | tail -4
} > $O/r06h.log 2>&1
tail -60 $O/r06h.log
