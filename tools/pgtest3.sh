for q in 4 8; do for o in pg_first eng_first; do for p in 0 -1; do
GPU_MAX_HW_QUEUES=$q python tools/pg_probe3.py $o $p on_stream 2>&1 | grep "prio\|Error\|error" | grep -v socket
done; done; done
