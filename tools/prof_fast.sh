cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for sc in "Stress1000 400 100" "StressN:20 410 50" "StressN:40 470 20"; do
  set -- $sc
  tag=$(echo $1 | tr ':' '_')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -o kt -- python3 tools/fast_prof.py $1 $2 $3 sweep=1 > gpurun_out/kt_$tag.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kt_$tag/**/kt_kernel_stats.csv", recursive=True)
for p in f:
    rows=list(csv.DictReader(open(p)))
    print("$tag")
    for r in rows[:14]:
        print("  %-60s calls %6s avg %10.1f us total%% %s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
