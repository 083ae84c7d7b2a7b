# HBM traffic records (profiles/r2_pmc_*.json): one rocprofv3 --pmc pass per counter and program, nothing else traced.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
run() {  # name, program args...
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/${name}_$c -- python3 "$@" > $O/${name}_$c.log 2>&1 || { echo "pass $name $c failed"; tail -n 5 $O/${name}_$c.log; exit 1; }
    echo "pass $name $c done"
  done
}
run attn_bwd $R/benchmarks/attn_mix.py c2 bwd 2 && \
run attn_fwd $R/benchmarks/attn_mix.py c2 fwd 2 && \
run sampling $R/benchmarks/sampling_once.py 8192 50304 3 || exit 1
P=$R/profiles
# attn_mix: 1 recording step + `reps` replays; the eager steps before the recording launch the same kernels: count launches from the log
python3 $R/benchmarks/pmc_traffic.py $O/attn_bwd_FETCH_SIZE $O/attn_bwd_WRITE_SIZE $P/r2_pmc_attn_bwd_c2.json -1 "bg_branch_attn_bwd (one-launch kernel; bf16, configs[1] launch mix)" branch_attn_bwd
python3 $R/benchmarks/pmc_traffic.py $O/attn_fwd_FETCH_SIZE $O/attn_fwd_WRITE_SIZE $P/r2_pmc_attn_fwd_c2.json -1 "bg_branch_attn_fwd (fwd + fwd_split kernels; bf16, configs[1] launch mix)" branch_attn_fwd
python3 $R/benchmarks/pmc_traffic.py $O/sampling_FETCH_SIZE $O/sampling_WRITE_SIZE $P/r2_pmc_topp.json 3 "bg_topp_select (topp_reset + topp_list + topp_regs_todo kernels), 8192 x 50304 fp32, p 0.9, n 4" topp_
python3 $R/benchmarks/pmc_traffic.py $O/sampling_FETCH_SIZE $O/sampling_WRITE_SIZE $P/r2_pmc_topn.json 6 "bg_topn_select (TOP_K and SAMPLE_NO_REPLACEMENT launches together), 8192 x 50304 fp32, n 4" topn_
