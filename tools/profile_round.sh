#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box through gpurun): kernel trace + statistics, SQ / MFMA counters,
# HBM traffic (FETCH_SIZE and WRITE_SIZE in separate passes), summarised to text / JSON; the raw databases are dropped.
#   bash tools/profile_round.sh r03_512 ; bash tools/profile_round.sh r03_256 --size 256 ; ... --size 1024 ; ... --multistyle 4
TAG=${1:-r02}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# one stream (kernel durations are serial, comparable with the in-bench HIP events), 8 frames per launch
shift
CMD="python $R/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 --pipeline 1 --profile-steps 1 $*"
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $O/pmc1 -o p -- $CMD > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $O/pmc2 -o p -- $CMD > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc3 -o p -- $CMD > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc4 -o p -- $CMD > $O/pmc4.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/prof_summary.py $(db trace) > $O/kernel_trace.txt 2> $O/kernel_trace.err
python tools/pmc_summary.py $(db pmc1) $(db pmc2) $(db pmc3) $(db pmc4) > $O/pmc.txt 2> $O/pmc.err
python tools/pmc_traffic_json.py $(db pmc3) $(db pmc4) "$CMD" > $O/hbm_traffic.json 2> $O/traffic.err
rm -rf $O/trace $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
python bench.py $* > $O/bench.json 2> $O/bench.err
ls -la $O; head -12 $O/kernel_trace.txt; tail -c 600 $O/bench.json
