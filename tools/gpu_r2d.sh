R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2d; mkdir -p $O
cd $R
timeout 200 python3 bench.py --no-cpu-baseline --no-extras --trace $O/trace_f32.json > $O/t_f32.json 2> $O/t_f32.err
python3 tools/stream_timeline.py $O/trace_f32.json 90
timeout 200 python3 bench.py --no-cpu-baseline --no-extras --depth 3 --trace $O/trace_d3.json > $O/t_d3.json 2> $O/t_d3.err
python3 tools/stream_timeline.py $O/trace_d3.json
timeout 200 python3 bench.py --no-cpu-baseline --no-extras --dtype bf16 --trace $O/trace_bf16.json > $O/t_bf16.json 2> $O/t_bf16.err
python3 tools/stream_timeline.py $O/trace_bf16.json
grep -h -o '"value": [0-9.]*' $O/t_f32.json $O/t_d3.json $O/t_bf16.json
