#!/bin/bash
# usage (GPU box): tools/pmc_vmem.sh <tag> <bench args...>  -- the vector-memory side of a render kernel: separate rocprofv3 --pmc
# passes (with --kernel-trace only) for instruction counts / issue cycles, TA / TCP stalls, instruction fetch and the TLB;
# writes gpurun_out/<tag>_{vm1,vm2,vm3,vm4,sq1}
tag=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
BARGS=("$@")
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$name -- python $R/bench.py "${BARGS[@]}" --steps 2 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${tag}_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU
run vm1 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM
run vm2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES
run vm3 TA_TA_BUSY TA_BUFFER_WRITE_WAVEFRONTS TA_BUFFER_READ_WAVEFRONTS TA_BUFFER_TOTAL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WRITE_WAVEFRONTS TA_FLAT_READ_WAVEFRONTS
run vm4 TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ_LATENCY TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCP_TA_DATA_STALL_CYCLES
run vm5 SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQC_TC_STALL SQC_DCACHE_MISSES SQC_DCACHE_HITS
