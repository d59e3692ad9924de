#!/bin/bash
# tools/gpu_ingest.sh — the edit / streaming tests, then the host-cost tools
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "edit or stream or format or operating or accel or api or multidevice or cpp" > gpurun_out/ingest_tests.log 2>&1 || { tail -30 gpurun_out/ingest_tests.log; exit 1; }
tail -2 gpurun_out/ingest_tests.log
python tools/stream_cost.py 2>/dev/null
python tools/edit_cost.py 8 2>/dev/null
python tools/edit_cost.py 32 2>/dev/null
