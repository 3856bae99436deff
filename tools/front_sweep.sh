for s in Stress1000 StressN:12 StressN:14; do
 for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
  echo "$s front=$1 topo=$2: $(AVBD_FUSED_FRONT=$1 AVBD_FUSED_TOPOLOGY=$2 python tools/time_scene.py $s 300 | tail -1 | cut -c1-260)"
 done
done
