G1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
G2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES"
G3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
bash profiles/collect_cli.sh r3_stress1000 Stress1000 420 20 "$G1" "$G2" "$G3" > gpurun_out/prof_s.log 2>&1
bash profiles/collect_cli.sh r3_8k StressN:20 410 20 "$G1" "$G2" "$G3" > gpurun_out/prof_8k.log 2>&1
bash profiles/collect_cli.sh r3_22k StressN:28 300 10 "$G1" "$G2" "$G3" > gpurun_out/prof_22k.log 2>&1
bash profiles/collect_cli.sh r3_64k StressN:40 470 5 "$G1" "$G2" "$G3" > gpurun_out/prof_64k.log 2>&1
ls gpurun_out/prof_r3_*/ | head -40
