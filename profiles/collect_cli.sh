#!/bin/bash
# rocprofv3 evidence for one scene through the headless CLI, starting from a saved steady-state snapshot so that the counter
# passes (which serialise dispatches) only run a handful of steps.  Run ON the MI355X box from the repo root:
#   bash profiles/collect_cli.sh <tag> <scene> <presim steps> <profiled steps> [extra --pmc groups...]
# e.g. bash profiles/collect_cli.sh r3_64k StressN:40 470 5 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
# AVBD_CLI_EXTRA="--mode fast" adds flags to every run of the CLI (round 6: the fast sweep).
# The program stands directly behind "--" (no shell hop).  Output: gpurun_out/prof_<tag>/{trace,fetch,write,pmcN}.
set -u
TAG=$1; SCENE=$2; PRESIM=$3; STEPS=$4; shift 4
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
CLI=$ROOT/avbd-demo3d_amd/avbd_demo3d
SNAP=/tmp/avbd_${TAG}.snap
EXTRA=${AVBD_CLI_EXTRA:-}
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
$CLI --nogfx --scene "$SCENE" --steps "$PRESIM" --batch $EXTRA --save-state "$SNAP" > "$OUT/presim.json" 2>&1
$CLI --nogfx --load-state "$SNAP" --steps 50 --batch $EXTRA > "$OUT/plain.json" 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CLI --nogfx --load-state "$SNAP" --steps 50 --batch $EXTRA > "$OUT/trace.json" 2> "$OUT/trace.err"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CLI --nogfx --load-state "$SNAP" --steps "$STEPS" --batch $EXTRA > "$OUT/fetch.json" 2> "$OUT/fetch.err"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CLI --nogfx --load-state "$SNAP" --steps "$STEPS" --batch $EXTRA > "$OUT/write.json" 2> "$OUT/write.err"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc$i" -- $CLI --nogfx --load-state "$SNAP" --steps "$STEPS" --batch $EXTRA > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err"
done
cd "$ROOT"
python3 profiles/summarise_cli.py "$OUT" "$TAG" "$STEPS"
