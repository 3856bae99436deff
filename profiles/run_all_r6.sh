#!/bin/bash
# the round-6 profiles (run on the MI355X box from the repo root); see profiles/README.md
#   bash profiles/run_all_r6.sh                 all seven: the exact mode's four, the fast sweep's three
#   bash profiles/run_all_r6.sh "r6_fast_64k StressN:40 470 5 fast"   one (tag scene presim steps [fast])
G1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
G3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
if [ $# -eq 0 ]; then set -- "r6_stress1000 Stress1000 420 20" "r6_8k StressN:20 410 20" "r6_64k StressN:40 470 5" "r6_64k_welds StressN:40+welds:10000 470 5" \
                             "r6_fast_1000 Stress1000 420 20 fast" "r6_fast_8k StressN:20 410 20 fast" "r6_fast_64k StressN:40 470 5 fast"; fi
for spec in "$@"; do
  read -r tag scene presim steps mode <<< "$spec"
  if [ "$mode" = "fast" ]; then export AVBD_CLI_EXTRA="--mode fast"; else unset AVBD_CLI_EXTRA; fi
  bash profiles/collect_cli.sh "$tag" "$scene" "$presim" "$steps" "$G1" "$G3" > "gpurun_out/prof_$tag.log" 2>&1
done
