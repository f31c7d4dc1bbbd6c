#!/bin/bash
set -u
O=gpurun_out/r03h; mkdir -p $O
T="timeout -k 10 300"
for kv in "NONE=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "GPU_MAX_HW_QUEUES=2" "HSA_ENABLE_INTERRUPT=0"; do
  echo "== $kv" >> $O/env_knobs.txt
  env $kv $T python tools/graph_vs_eager.py eager torch nodes 4096 65536 >> $O/env_knobs.txt 2>&1
done
grep -v amdgpu.ids $O/env_knobs.txt
