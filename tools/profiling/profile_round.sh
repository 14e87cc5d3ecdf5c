#!/bin/bash
# Profile artefacts of a round (TAG=r04 by default) (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats of the three
# workloads, SQ counter table and HBM traffic (separate --pmc passes, kernel-trace only) of the layer bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r04}; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --workload train --steps 10 --warmup 4 > $O/bench_train_lite.json 2> $O/bench_train_lite.err
python bench.py --workload train --model configPCF_10cm --steps 10 --warmup 4 > $O/bench_train_10cm.json 2> $O/bench_train_10cm.err
python bench.py --workload train --model configPCF_5cm --steps 10 --warmup 4 > $O/bench_train_5cm.json 2> $O/bench_train_5cm.err
python bench.py --workload train --model configPCF_2cm_PTF2 --steps 10 --warmup 4 > $O/bench_train_2cm.json 2> $O/bench_train_2cm.err
python bench.py --workload subsample --steps 10 --warmup 3 > $O/bench_subsample.json 2> $O/bench_subsample.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_layer -o layer -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_layer.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -o train -- python3 $R/bench.py --workload train --no-graph --steps 10 --warmup 4 > $O/kt_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train10 -o train10 -- python3 $R/bench.py --workload train --model configPCF_10cm --no-graph --steps 10 --warmup 4 > $O/kt_train10.log 2>&1
CMD="python3 $R/bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 2"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc1 -o p1 -- $CMD > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA -d $O/pmc2 -o p2 -- $CMD > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -o p3 -- $CMD > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc4 -o p4 -- $CMD > $O/pmc4.log 2>&1
cd $R
python3 tools/profiling/pmc_table.py $O/pmc1 $O/pmc2 > $O/sq_table.txt 2>&1
python3 tools/profiling/pmc.py $O/pmc3 > $O/fetch.txt 2>&1
python3 tools/profiling/pmc.py $O/pmc4 > $O/write.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
find $O -name "*counter_collection.csv" -size +8M -delete
ls $O; head -5 $O/sq_table.txt
