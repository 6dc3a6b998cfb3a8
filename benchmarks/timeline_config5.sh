cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o p -- python benchmarks/config5_step.py --plain --steps 20 > $O/c5.json 2> $O/c5.err
python benchmarks/step_timeline.py $(find $O/prof_c5 -name '*kernel_trace.csv' | head -1) > $O/timeline_c5.txt
rm -rf $O/prof_c5
tail -1 $O/c5.json
