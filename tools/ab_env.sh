#!/bin/bash
# A/B of runtime switches (environment variables) of ONE build inside one GPU session: every setting twice, interleaved.
#   gpurun -- 'bash tools/ab_env.sh "LGH_MVQ_EXP=0" "LGH_MVQ_EXP=1" "LGH_MVQ_EXP=2" "LGH_MVQ_EXP=3"'
# Prints tokens/s of bench.py (Llama-3-8B Q4_K_M headline workload, 256 steps) per setting; LGH_AB_MODEL=tinyllama-1.1b etc. changes the model.
MODEL=${LGH_AB_MODEL:-llama-3-8b}
for round in 1 2; do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --model "$MODEL" --cpu-seconds 0 --profile-steps 0 --steps 256 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %8.2f tok/s  %.4f ms' % ('$v', d['value'], d['ms_per_step']))"
  done
done
