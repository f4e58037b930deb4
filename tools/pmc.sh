#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag> <bench args...>   -- separate rocprofv3 --pmc passes of `python bench.py <args>`,
# each with --kernel-trace only (never combined with other trace domains), plus one --kernel-trace --stats run;
# writes gpurun_out/<tag>_{sq1,sq2,sq3,grbm,fetch,write,stats}
tag=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$name -- python $R/bench.py "${BARGS[@]}" --steps 2 --warmup 0 --no-cpu-baseline --no-also > $R/gpurun_out/${tag}_$name.log 2>&1; }
BARGS=("$@")
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD
run sq3 SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python $R/bench.py "${BARGS[@]}" --steps 3 --warmup 1 --no-cpu-baseline --no-also > $R/gpurun_out/${tag}_stats.log 2>&1
