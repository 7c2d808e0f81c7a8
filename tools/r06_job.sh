cd $GRAFT_REPO_ROOT; L=$PWD/bmcnet-esr_amd/csrc; O=$PWD/gpurun_out; mkdir -p $O
{
for r in 1 2; do for s in hip hip_w4dprio10 hip_w4dprio11 hip_w4dprio12; do echo "== $r $s: $(W4_ONLY=1 KB_ITERS=300 BMC_HIP_LIB=$L/libbmc_$s.so timeout 200 python tools/time_wino4.py 2>&1 | grep 'F(4x4)' | sed 's/algorithmic.*executed//' | tr '\n' ' ')"; done; done
} > $O/r06k.log 2>&1
tail -40 $O/r06k.log
