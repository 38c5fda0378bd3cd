#!/bin/bash
# A/B of the single-launch decode attention at short contexts: bash tools/ab_attn.sh
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 --profile-steps 0 --prompt $1 --warmup 4 --steps $2 --attn-direct $3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('prompt $1 steps $2 attn_direct $3:', d['value'], 'tokens/s', d['ms_per_step'], 'ms')"; }
for a in 1 255 1 255; do run 8 48 $a; done
for a in 2 255 2 255; do run 64 56 $a; done
