# rocprofv3 kernel stats of bench.py --workload <w>, by class and one replayed step.   usage: prof_workload.sh <workload> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-c3}
T=${2:-x}
O=$R/gpurun_out/prof_${W}_$T
( while sleep 60; do echo "profiling $W ..."; done ) &      # a silent command is taken to be hung after 180 s
HB=$!
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload $W --no-cpu-baseline --no-fp32 --no-rooflines > $R/gpurun_out/r${ROUND:-3}_${T}_bench_$W.json 2> $R/gpurun_out/r${ROUND:-3}_${T}_bench_$W.err || { kill $HB; tail -n 5 $R/gpurun_out/r${ROUND:-3}_${T}_bench_$W.err; exit 1; }
kill $HB
f=$(ls -t $O/*/*kernel_stats.csv | head -n 1)
cp $f $R/gpurun_out/r${ROUND:-3}_${T}_bench_${W}_kernel_stats.csv
t=$(ls -t $O/*/*kernel_trace.csv | head -n 1)
python3 $R/benchmarks/step_census.py $t > $R/gpurun_out/r${ROUND:-3}_${T}_bench_${W}_one_step.txt && head -n 40 $R/gpurun_out/r${ROUND:-3}_${T}_bench_${W}_one_step.txt
rm -f $t
