# rocprofv3 kernel stats of the attention forward launches of one train step: key-split kernel for every shape vs the round-1 rule
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-c2}
for v in 1 0; do
  export BRANCHGAN_ATTN_FWD_SPLIT_ALL=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_attn_fwd$v -- python3 $R/benchmarks/attn_mix.py $WL fwd 5 > $R/gpurun_out/prof_attn_fwd$v.log 2>&1 || exit 1
  f=$(ls -t $R/gpurun_out/prof_attn_fwd$v/*/*kernel_stats.csv | head -n 1)
  echo "== SPLIT_ALL=$v"; grep attn_fwd $f | cut -d, -f1-4 | cut -c1-120
done
