#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel trace and the two PMC passes (FETCH_SIZE / WRITE_SIZE,
# each in its own run as MI355X_MICROARCH.md prescribes).  Usage (from the repo root, through gpurun):
#   gpurun --timeout 900 -- 'bash tools/gpu_measure.sh r01b'
# Everything lands under gpurun_out/<tag>/ ; tools/summarize_profiles.py turns it into profiles/<tag>_*.
set -o pipefail
TAG=${1:-r01}
STEPS=${2:-128}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$(pwd)

timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
rc=$?; tail -2 "$OUT/pytest_gpu.log"; [ $rc -ne 0 ] && { echo "pytest rc=$rc"; exit $rc; }

timeout -k 10 300 python bench.py --steps "$STEPS" --warmup 16 > "$OUT/bench.json" 2> "$OUT/bench.err"
rc=$?; echo "bench exit $rc"; [ $rc -ne 0 ] && { tail -5 "$OUT/bench.err"; exit $rc; }

cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof" -o ktrace --output-format csv -- \
  python3 "$R/bench.py" --steps "$STEPS" --warmup 16 > "$R/$OUT/prof_bench.json" 2> "$R/$OUT/prof.log"
rc=$?; echo "prof exit $rc"; [ $rc -ne 0 ] && { tail -5 "$R/$OUT/prof.log"; exit $rc; }

timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$R/$OUT/pmc_fetch" -o f --output-format csv -- \
  python3 "$R/bench.py" --steps 8 --warmup 2 --prompt 8 --cpu-seconds 0 --profile-steps 0 > /dev/null 2> "$R/$OUT/pmc_fetch.log"
rc=$?; echo "pmc fetch exit $rc"; [ $rc -ne 0 ] && { tail -5 "$R/$OUT/pmc_fetch.log"; exit $rc; }

timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$R/$OUT/pmc_write" -o w --output-format csv -- \
  python3 "$R/bench.py" --steps 8 --warmup 2 --prompt 8 --cpu-seconds 0 --profile-steps 0 > /dev/null 2> "$R/$OUT/pmc_write.log"
rc=$?; echo "pmc write exit $rc"; [ $rc -ne 0 ] && { tail -5 "$R/$OUT/pmc_write.log"; exit $rc; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof_prefill" -o pf --output-format csv -- \
  python3 "$R/tools/prefill_bench.py" > "$R/$OUT/prefill_bench.log" 2>&1
rc=$?; echo "prefill prof exit $rc"; grep "tokens/s" "$R/$OUT/prefill_bench.log"
cd "$R"
# the trace itself is large; keep the stats and drop the per-dispatch rows beyond what the summary needs
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
ls -R "$OUT" | head -40
cat "$OUT/bench.json"
