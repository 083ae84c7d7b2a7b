# rocprofv3 kernel stats of the attention launches (forward and backward) of one c4 train step (Dh 256 generator, Dh 64 discriminator)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-x}
( while sleep 50; do echo "profiling c4 attention ..."; done ) &
HB=$!
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_attn_c4_$T -- python3 $R/benchmarks/attn_mix.py c4 both 5 > $R/gpurun_out/prof_attn_c4_$T.log 2>&1 || { kill $HB; tail -n 5 $R/gpurun_out/prof_attn_c4_$T.log; exit 1; }
kill $HB
f=$(ls -t $R/gpurun_out/prof_attn_c4_$T/*/*kernel_stats.csv | head -n 1)
grep attn $f | cut -d, -f1-4 | cut -c1-130 | tee $R/gpurun_out/r3_attn_c4_$T.txt
rm -f $R/gpurun_out/prof_attn_c4_$T/*/*kernel_trace.csv
