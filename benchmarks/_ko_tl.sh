export TMPDIR=/tmp
O=$PWD/gpurun_out
cp salience_detr_amd/libsalience_hip.so /tmp/prod.so
for v in prod koattn koproj; do
  if [ $v = prod ]; then cp /tmp/prod.so salience_detr_amd/libsalience_hip.so; else cp benchmarks/libv_$v.so salience_detr_amd/libsalience_hip.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k -o p -- python bench.py --plain --steps 20 > /dev/null 2>&1
  python benchmarks/step_timeline.py $(find $O/prof_k -name '*kernel_trace.csv' | head -1) | grep "fused_attn_proj" | awk '{print $5}' | tr '\n' ' ' | sed "s/^/$v: /"; echo
  rm -rf $O/prof_k
done
cp /tmp/prod.so salience_detr_amd/libsalience_hip.so
