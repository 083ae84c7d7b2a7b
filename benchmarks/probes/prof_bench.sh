# rocprofv3 --kernel-trace --stats of the default bench line (configs[1], bf16, 10 steps after 3), the by-class table, and the same
# run unprofiled: profiles/r<round>_<tag>_bench_c2*.   usage: prof_bench.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-x}
O=$R/gpurun_out/prof_bench_$T
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-fp32 --no-rooflines > $R/gpurun_out/r${ROUND:-3}_${T}_bench_c2.json 2> $R/gpurun_out/r${ROUND:-3}_${T}_bench_c2.err || { tail -n 5 $R/gpurun_out/r${ROUND:-3}_${T}_bench_c2.err; exit 1; }
f=$(ls -t $O/*/*kernel_stats.csv | head -n 1)
cp $f $R/gpurun_out/r${ROUND:-3}_${T}_bench_c2_kernel_stats.csv
python3 $R/benchmarks/prof_summary.py $f 13 > $R/gpurun_out/r${ROUND:-3}_${T}_bench_c2_by_class.txt
cat $R/gpurun_out/r${ROUND:-3}_${T}_bench_c2_by_class.txt
cd $R && timeout -k 10 400 python3 bench.py > gpurun_out/r${ROUND:-3}_${T}_bench_c2_unprofiled.json 2> gpurun_out/r${ROUND:-3}_${T}_bench_c2_unprofiled.err
tail -n 1 gpurun_out/r${ROUND:-3}_${T}_bench_c2_unprofiled.json | cut -c1-600
t=$(ls -t $O/*/*kernel_trace.csv | head -n 1)
python3 $R/benchmarks/step_census.py $t > $R/gpurun_out/r${ROUND:-3}_${T}_bench_c2_one_step.txt && head -n 12 $R/gpurun_out/r${ROUND:-3}_${T}_bench_c2_one_step.txt
