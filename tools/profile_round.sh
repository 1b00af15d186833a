cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r1g
mkdir -p $O
CMD="python $R/bench.py --no-cpu-baseline --steps 16 --warmup 2"
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $O/pmc1 -o p -- $CMD > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $O/pmc2 -o p -- $CMD > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc3 -o p -- $CMD > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc4 -o p -- $CMD > $O/pmc4.log 2>&1
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
ls -la $O $O/*/ | head -40; tail -1 $O/bench.json | cut -c1-300
