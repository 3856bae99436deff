export TMPDIR=/tmp
OUT=gpurun_out/prof_sharded1
rm -rf $OUT; mkdir -p $OUT
AVBD_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu --force-dist > $OUT/bench.json 2> $OUT/err.txt
tail -2 $OUT/bench.json | cut -c1-400
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -15 $f | cut -c1-200
