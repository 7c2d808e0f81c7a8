cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out; mkdir -p $O
{
bash tools/gpu_run.sh r06 bench
python3 -c "import json;d=json.load(open('$O/r06_bench.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['all_host_cores']['value'])"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r06t_bench20.json 2> $O/r06t_bench20.err; python3 -c "import json;d=json.load(open('$O/r06t_bench20.json'));print('steps20', d['ms_per_step'], d['value'])"
} > $O/r06t.log 2>&1
tail -4 $O/r06t.log | cut -c1-300
