#!/bin/bash
# usage (on the GPU box): tools/r3_base.sh <tag>: SQ counters + kernel stats of the back half at 4096 and 512 channels (config 4)
tag=$1
R=$GRAFT_REPO_ROOT
C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"
bash tools/pmc_any.sh ${tag}_c4all_sq "$C" tools/shape_bench.py 64000000 524288 524289 2048 4096 16 > gpurun_out/${tag}_c4all_sq.txt 2>&1 || exit 1
bash tools/pmc_any.sh ${tag}_c4share_sq "$C" tools/shape_bench.py 64000000 524288 524289 2048 512 16 > gpurun_out/${tag}_c4share_sq.txt 2>&1 || exit 1
bash tools/pmc_any.sh ${tag}_c2_sq "$C" tools/shape_bench.py 8000000 131072 131073 256 640 32 > gpurun_out/${tag}_c2_sq.txt 2>&1 || exit 1
cd $R && bash tools/big_shapes.sh gpurun_out/${tag}_shapes.txt
