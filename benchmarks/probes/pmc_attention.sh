# HBM traffic of the attention launches of one train step (profiles/r<round>_pmc_attn_{bwd,fwd}_<workload>.json): one rocprofv3
# --pmc pass per counter and direction over benchmarks/attn_mix.py, nothing else traced.   usage: pmc_attention.sh <workload> [round]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-c2}
RD=${2:-3}
O=$R/gpurun_out/pmc_attn_r${RD}_$W
mkdir -p $O
( while sleep 50; do echo "pmc passes over $W ..."; done ) &
HB=$!
for kind in bwd fwd; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 280 rocprofv3 --pmc $c --output-format csv -d $O/${kind}_$c -- python3 $R/benchmarks/attn_mix.py $W $kind 2 > $O/${kind}_$c.log 2>&1 || { kill $HB; echo "pass $kind $c failed"; tail -n 5 $O/${kind}_$c.log; exit 1; }
    echo "pass $kind $c done"
  done
done
kill $HB
python3 $R/benchmarks/pmc_traffic.py $O/bwd_FETCH_SIZE $O/bwd_WRITE_SIZE $R/gpurun_out/r${RD}_pmc_attn_bwd_$W.json -1 "bg_branch_attn_bwd (every kernel of the backward launches; bf16, $W launch mix; per kernel dispatch)" branch_attn_bwd > /dev/null && \
python3 $R/benchmarks/pmc_traffic.py $O/fwd_FETCH_SIZE $O/fwd_WRITE_SIZE $R/gpurun_out/r${RD}_pmc_attn_fwd_$W.json -1 "bg_branch_attn_fwd (split + merge kernels; bf16, $W launch mix; per kernel dispatch)" branch_attn_fwd > /dev/null && \
python3 - <<PY
import json
for k in ("bwd", "fwd"):
    d = json.load(open("$R/gpurun_out/r${RD}_pmc_attn_%s_$W.json" % k))
    print(k, d["dispatches"], d["dispatches_by_kernel"], round(d["traffic_bytes_per_launch"] / 1e6, 2), "MB per dispatch")
PY
