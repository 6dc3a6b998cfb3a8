cp salience_detr_amd/libsalience_hip.so /tmp/prod.so
cp benchmarks/libv_${V:-tkstamps}.so salience_detr_amd/libsalience_hip.so
python bench.py --plain --no-graph --steps 3 --warmup 1 2>&1 | grep -E "attn blk|token_linear blk" | tail -12
cp /tmp/prod.so salience_detr_amd/libsalience_hip.so
