cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out; mkdir -p $O
{
bash tools/gpu_run.sh r06 bench
python3 -c "import json;d=json.load(open('$O/r06_bench.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['pmc'], d['cpu_baseline']['all_host_cores']['value'])"
} > $O/r06p.log 2>&1
tail -5 $O/r06p.log | cut -c1-600
