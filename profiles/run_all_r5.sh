#!/bin/bash
# the four round-5 profiles (the last one is BASELINE configs[4]: 64 000 cubes + 10 000 weld joints) (run on the MI355X box from the repo root); see profiles/README.md
#   bash profiles/run_all_r5.sh                       all four
#   bash profiles/run_all_r5.sh "r5_64k StressN:40 470 5"   one (tag scene presim steps)
G1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
G2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES"
G3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
if [ $# -eq 0 ]; then set -- "r5_stress1000 Stress1000 420 20" "r5_8k StressN:20 410 20" "r5_64k StressN:40 470 5" "r5_64k_welds StressN:40+welds:10000 470 5"; fi
for spec in "$@"; do
  read -r tag scene presim steps <<< "$spec"
  bash profiles/collect_cli.sh "$tag" "$scene" "$presim" "$steps" "$G1" "$G2" "$G3" > "gpurun_out/prof_$tag.log" 2>&1
done
