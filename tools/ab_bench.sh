#!/bin/bash
# A/B two builds of the engine inside ONE GPU session (boxes differ by a few percent, runs on one box by ~0.5 %):
#   gpurun -- 'bash tools/ab_bench.sh base ""'     compares lib/libllama_gguf_hip_base.so with the default library
# Prints tokens/s of bench.py for A, B, A, B.
A=${1:-base}; B=${2:-}
for round in 1 2; do
  for v in "$A" "$B"; do
    LGH_LIB_VARIANT=$v timeout -k 10 200 python bench.py --cpu-seconds 0 --profile-steps 0 --steps 256 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=%-8s %8.2f tok/s  %.4f ms' % ('$v' or 'default', d['value'], d['ms_per_step']))"
  done
done
