# latency and SQ counters of lsap_kernel by graph size (one wavefront per graph): time per launch, then
# one counter pass per size (8 SQ counters fit one pass)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lsap_pmc
rm -rf $O; mkdir -p $O
timeout -k 10 120 python3 benchmarks/lsap_sizes.py 256 16 32 > $O/sizes_256.log 2>&1
timeout -k 10 120 python3 benchmarks/lsap_sizes.py 64 64 80 128 > $O/sizes_64.log 2>&1
for cfg in "256 32" "64 80" "64 128"; do
  set -- $cfg
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_$2 -o p -- python3 benchmarks/lsap_sizes.py $1 $2 > $O/pmc_$2.log 2>&1
  python3 benchmarks/pmc_summary.py $O/pmc_$2 $O/lsap_pmc_$2.csv lsap_kernel
done
rm -f $O/*/*.db $O/*/*/*.db
cat $O/sizes_256.log $O/sizes_64.log
