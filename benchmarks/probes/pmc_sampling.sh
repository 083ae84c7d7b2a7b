# HBM traffic of the sampling kernels (profiles/r<round>_pmc_topp.json, _topn.json): one rocprofv3 --pmc pass per counter, nothing
# else traced.   usage: pmc_sampling.sh [round]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RD=${1:-3}
O=$R/gpurun_out/pmc_r$RD
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/sampling_$c -- python3 $R/benchmarks/sampling_once.py 8192 50304 3 > $O/sampling_$c.log 2>&1 || { echo "pass $c failed"; tail -n 5 $O/sampling_$c.log; exit 1; }
  echo "pass $c done"
done
P=$R/gpurun_out
python3 $R/benchmarks/pmc_traffic.py $O/sampling_FETCH_SIZE $O/sampling_WRITE_SIZE $P/r${RD}_pmc_topp.json 3 "bg_topp_select (topp_reset + topp_list + topp_regs_todo kernels), 8192 x 50304 fp32, p 0.9, n 4" topp_ && \
python3 $R/benchmarks/pmc_traffic.py $O/sampling_FETCH_SIZE $O/sampling_WRITE_SIZE $P/r${RD}_pmc_topn.json 6 "bg_topn_select (TOP_K and SAMPLE_NO_REPLACEMENT launches together), 8192 x 50304 fp32, n 4" topn_
