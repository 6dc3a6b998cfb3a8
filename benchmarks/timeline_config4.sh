cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o p -- python benchmarks/config4_step.py --plain --steps 20 > $O/c4.json 2> $O/c4.err
python benchmarks/step_timeline.py $(find $O/prof_c4 -name '*kernel_trace.csv' | head -1) > $O/timeline_c4.txt
rm -rf $O/prof_c4
tail -1 $O/c4.json
