#!/bin/bash
# Collects the rocprofv3 evidence for one build (run ON the MI355X box, from the repo root):
#   bash profiles/collect.sh <tag> [bench.py scene arguments]     e.g. r1_resident      or      r1_8k --scene StressN:20
# Three separate passes (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE), each with the program directly
# after "--" (no shell hop), then profiles/summarise.py reduces them to the small files that are committed.
set -u
TAG=${1:-r1}
shift
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 100 --warmup 10 --no-cpu --no-strong-n1 "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
# SKIP_PMC=1: kernel trace + plain bench only (the counter passes of the 64 000-cube scene did not finish within 40 minutes:
# under --pmc the several-bodies-per-sub-group launch is serialised dispatch by dispatch and its steps take seconds)
if [ -z "${SKIP_PMC:-}" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-strong-n1 "$@" > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-strong-n1 "$@" > "$OUT/bench_write.json" 2> "$OUT/write.err"
fi
python3 bench.py --steps ${BENCH_STEPS:-200} --warmup 20 --no-strong-n1 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 profiles/summarise.py "$OUT" "$TAG"
