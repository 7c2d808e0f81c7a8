cd $GRAFT_REPO_ROOT; L=bmcnet-esr_amd/csrc; O=gpurun_out; mkdir -p $O
{
timeout 600 python -m pytest tests/test_gpu_r4.py -x -q -m gpu 2>&1 | tail -4
for r in 1 2; do for s in hip hip_w4il0; do echo "== $r $s: $(W4_ONLY=1 KB_ITERS=200 BMC_HIP_LIB=$PWD/$L/libbmc_$s.so timeout 200 python tools/time_wino4.py 2>&1 | grep 'F(4x4)' | sed 's/algorithmic.*executed//' | tr '\n' ' ')"; done; done
for s in w4ilclk1 w4ilclk0 w4clkabl1 w4clkabl2 w4clkabl4 w4clkabl6 w4clkabl8 w4clkabl16 w4clkabl32 w4clkabl64 w4clkabl254 w4ilclk1; do echo "== clock $s: $(W4_CLOCK_SECONDS=1.5 BMC_HIP_LIB=$PWD/$L/libbmc_hip_$s.so timeout 200 python tools/w4_clock.py 2>&1 | tail -1)"; done
timeout 1500 python -m pytest tests/test_gpu_r6.py -x -q -m gpu -s 2>&1 | tail -40
} > $O/r06c.log 2>&1
tail -70 $O/r06c.log
