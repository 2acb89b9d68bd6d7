set -u
REPO=$(pwd); mkdir -p $REPO/gpurun_out/summ; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAVES -d /tmp/pmc_sq -o run -- python $REPO/bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --concurrent 1 > /tmp/pmc_sq.log 2>&1
db=$(find /tmp/pmc_sq -name '*.db' | head -1)
python $REPO/tools/rocpd_pmc.py $db | grep -E "k_lr_|kernel" | cut -c1-200
