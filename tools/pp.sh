export GPU_MAX_HW_QUEUES=16
for ns in 3 4; do echo "== NS=$ns"; PIPE_NS=$ns timeout 400 python3 tools/pipe_probe.py 2>&1 | grep "full pipeline\|forward (fresh output)  "; done
