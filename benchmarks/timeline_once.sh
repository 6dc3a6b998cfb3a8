cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x -o p -- python bench.py --plain --steps 20 > $O/prof_x.json 2> $O/prof_x.err
python benchmarks/step_timeline.py $(find $O/prof_x -name '*kernel_trace.csv' | head -1) > $O/timeline_x.txt
rm -rf $O/prof_x
cat $O/timeline_x.txt
