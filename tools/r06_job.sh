cd $GRAFT_REPO_ROOT; L=$PWD/bmcnet-esr_amd/csrc; O=$PWD/gpurun_out; mkdir -p $O
{
timeout 600 python -m pytest tests/test_gpu_r4.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do for s in hip hip_w4epi0; do echo "== $r $s: $(W4_ONLY=1 KB_ITERS=300 BMC_HIP_LIB=$L/libbmc_$s.so timeout 200 python tools/time_wino4.py 2>&1 | grep 'F(4x4)' | sed 's/algorithmic.*executed//' | tr '\n' ' ')"; done; done
timeout 1500 python -m pytest tests/test_gpu_r6.py -x -q -m gpu 2>&1 | tail -30
} > $O/r06f.log 2>&1
tail -50 $O/r06f.log
