#!/bin/bash
# tools/profile_round.sh TAG   (run ON the GPU box, from the repo root: gpurun -- 'bash tools/profile_round.sh r3a')
# rocprofv3 --kernel-trace --stats of the headline bench leg, then four separate --pmc passes of the same command (one counter set per pass,
# --kernel-trace only: MI355X_MICROARCH.md, HBM / rocprofv3 section), then tools/pmc_summary.py + tools/pmc_traffic.py.
# Everything lands under gpurun_out/TAG_* (the only directory that travels back); copy what is to be judged into profiles/.
set -u
TAG="${1:-prof}"
R="${GRAFT_REPO_ROOT:-$PWD}"
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out"
B="python $R/bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline --no-s64 --no-simil --no-post-pass --no-scenes"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o s --output-format csv -- $B > $O/${TAG}_bench_under_rocprofv3.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmc_f -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/${TAG}_pmc_w -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/${TAG}_pmc_s -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/${TAG}_pmc_i -o p --output-format csv -- $B > /dev/null 2>&1
cd $R
STATS=$(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1)
cp "$STATS" $O/${TAG}_kernel_stats.csv
python tools/pmc_summary.py --out $O/${TAG}_summary.json --stats "$STATS" --pmc $(find $O/${TAG}_pmc_f $O/${TAG}_pmc_w $O/${TAG}_pmc_s $O/${TAG}_pmc_i -name "*counter_collection.csv")
python tools/pmc_traffic.py --summary $O/${TAG}_summary.json && cp profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
rm -rf $O/${TAG}_stats $O/${TAG}_pmc_f $O/${TAG}_pmc_w $O/${TAG}_pmc_s $O/${TAG}_pmc_i
tail -1 $O/${TAG}_bench_under_rocprofv3.json | cut -c1-400
