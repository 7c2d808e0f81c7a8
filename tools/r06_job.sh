cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out; mkdir -p $O
export BMC_PMC_COMMIT=817096d
{
bash tools/gpu_run.sh r06 pmc
timeout 900 python bench.py --steps 3 --warmup 2 --no-bf16x6 --also none > $O/r06o_bench.json 2> $O/r06o_bench.err; python3 -c "import json;d=json.load(open('$O/r06o_bench.json'));print(d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['all_host_cores'], d['roofline']['traffic'])"
} > $O/r06o.log 2>&1
tail -30 $O/r06o.log | cut -c1-700
