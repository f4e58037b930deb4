#!/bin/bash
# usage: tools/pmc_book2.sh <tag>   -- SQ counters for the full-feature pool kernel on book-2 800x800x100 (GPU box)
tag=$1; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$name -- python $R/bench.py --workload book2 --spp 100 --steps 2 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${tag}_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD
run sq3 SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python $R/bench.py --workload book2 --spp 100 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1
