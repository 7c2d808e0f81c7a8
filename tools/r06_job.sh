cd $GRAFT_REPO_ROOT; L=$PWD/bmcnet-esr_amd/csrc; O=$PWD/gpurun_out; mkdir -p $O
{
for r in 1 2 3; do for s in hip hip_pgstagger0; do echo "== $r $s: $(BMC_HIP_LIB=$L/libbmc_$s.so timeout 200 python tools/time_pgemm1.py 2>&1 | tail -1)"; done; done
timeout 1200 python -m pytest tests/test_gpu_r2.py tests/test_gpu_parity.py tests/test_gpu_r3.py -x -q -m gpu -k "bie or pgemm or golden or attn or gram or wgrad" 2>&1 | tail -3
for s in hip hip_pgstagger0 hip hip_pgstagger0; do BMC_HIP_LIB=$L/libbmc_$s.so timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none > $O/r06v_$s.json 2> $O/r06v_$s.err; echo "$s: $(grep -o '"ms_per_step": [0-9.]*' $O/r06v_$s.json | head -1)"; done
} > $O/r06v.log 2>&1
tail -40 $O/r06v.log
