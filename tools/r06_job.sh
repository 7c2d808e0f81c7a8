cd $GRAFT_REPO_ROOT; L=$PWD/bmcnet-esr_amd/csrc; O=$PWD/gpurun_out; mkdir -p $O
export BMC_PMC_COMMIT=a4c0cc4
{
timeout 600 python -m pytest tests/test_gpu_r3.py tests/test_gpu_r5.py -x -q -m gpu -k "wgrad or weight_grad or winograd" 2>&1 | tail -3
for r in 1 2; do for s in hip hip_wwil0; do for b in 8 16; do echo "== wgrad $r $s B=$b: $(TW_B=$b TW_ONLY=winograd BMC_HIP_LIB=$L/libbmc_$s.so timeout 200 python tools/time_wgrad.py 2>&1 | tail -2 | tr '\n' ' ')"; done; done; done
for s in hip hip_wwil0 hip; do BMC_HIP_LIB=$L/libbmc_$s.so timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16x6 --also none > $O/r06n_$s.json 2> $O/r06n_$s.err; echo "$s: $(grep -o '"ms_per_step": [0-9.]*' $O/r06n_$s.json | head -1)"; done
bash tools/gpu_run.sh r06 bench
} > $O/r06n.log 2>&1
tail -30 $O/r06n.log | cut -c1-700
