#!/bin/bash
# Where does a pipeline boundary's time go?  rocprofv3 HIP-API + kernel + copy trace of the in-library 2-stage pipeline on ONE
# GPU.  Usage (through gpurun): bash tools/pp_trace.sh <out-tag>
TAG=${1:-pp}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 280 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d "$OUT/trace" -o pp --output-format csv -- \
  python3 "$R/bench.py" --inlib --gpus 1 --stages 2 --steps 24 --warmup 4 --reps 1 --prompt 8 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "exit $?"
cd "$R"
python3 tools/pp_trace_summary.py "$OUT/trace" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt" | head -80
