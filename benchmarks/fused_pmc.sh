# per-iteration latency and SQ counters of the S-free MPM kernels by graph size (256 graphs: one per CU;
# 100 iterations, no early stop), one counter pass per size
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fused_pmc
rm -rf $O; mkdir -p $O
timeout -k 10 200 python3 benchmarks/fused_sizes.py 256 100 16 32 44 64 88 115 128 > $O/sizes.log 2>&1
for r in 32 64 128; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_$r -o p -- python3 benchmarks/fused_sizes.py 256 100 $r > $O/pmc_$r.log 2>&1
  python3 benchmarks/pmc_summary.py $O/pmc_$r $O/fused_pmc_$r.csv mpm_fused
done
rm -f $O/*/*.db $O/*/*/*.db
cat $O/sizes.log
