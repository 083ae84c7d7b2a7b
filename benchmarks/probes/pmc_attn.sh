cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_attn_sq -- python3 $R/benchmarks/attn_mix.py c2 bwd 2 > $R/gpurun_out/pmc_attn_sq.log 2>&1
