# round 6, batch 2: nocopy probe, f32 cell phases in the job, kernel trace of the driver-form line (committed summary), full GPU suite timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O; cd $R
timeout 300 python3 tools/r06/nocopy_probe.py > $O/nocopy_probe.json 2> $O/nocopy_probe.err
timeout 300 python3 tools/r06/cell_phases.py > $O/cell_phases.json 2> $O/cell_phases.err
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_full.txt 2>&1
cat $O/nocopy_probe.json $O/cell_phases.json; tail -3 $O/nocopy_probe.err $O/cell_phases.err; cat $O/pytest_full.txt
