cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 150 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -8 > gpurun_out/final/pytest.log
timeout 60 python benchmarks/transformer_micro.py --neck 2>&1 | tail -2 > gpurun_out/final/neck_micro.log
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/neck_prof -- python benchmarks/transformer_micro.py --neck --iters 5 > /dev/null 2>&1
timeout 90 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 > gpurun_out/final/smoke.log
cat gpurun_out/final/pytest.log gpurun_out/final/neck_micro.log gpurun_out/final/smoke.log; head -c 600 gpurun_out/final/bench.json
