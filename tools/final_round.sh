#!/bin/bash
# Round-end artefacts in ONE gpurun call: bench lines of every configuration, one rocprofv3 kernel trace + step timeline + per-step kernel
# table, the kernel micro-benchmarks, then the whole -m gpu suite.  usage: tools/final_round.sh <tag>
tag=${1:-r06}
mkdir -p gpurun_out; export TMPDIR=/tmp
b() { name=$1; shift; timeout 300 python bench.py --steps 50 --warmup 10 "$@" > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err; echo "bench $name rc=$? $(cut -c1-200 gpurun_out/${tag}_bench_${name}.json)"; }
b std
b large --tile large
b tsrn --arch tsrn --no-cpu-baseline
b tbsrn --arch tbsrn --no-cpu-baseline
b tpg --arch tatt_tpg
b tssim --tssim --no-cpu-baseline
b dp_selftest --dp-selftest --no-cpu-baseline
tools/gpu_quick.sh ${tag}_final none "prof:" > /dev/null 2>&1
head -3 gpurun_out/${tag}_final_timeline.txt
for cfg in "tbsrn:--arch tbsrn" "tpg:--arch tatt_tpg" "large:--tile large"; do     # kernel traces of the other configurations
  tools/gpu_quick.sh ${tag}_${cfg%%:*} none "prof:${cfg#*:}" > /dev/null 2>&1; head -1 gpurun_out/${tag}_${cfg%%:*}_timeline.txt
done
bash tools/sattn_prof.sh pmc > gpurun_out/${tag}_sattn_ab.txt 2>&1; grep "^gen" gpurun_out/${tag}_sattn_ab.txt
[ -x tools/ubench/coissue ] && timeout 60 tools/ubench/coissue > gpurun_out/${tag}_valu_mfma_coissue.txt 2>&1
timeout 300 python tools/bench_kernels.py > gpurun_out/${tag}_kernel_microbench.txt 2>&1; tail -3 gpurun_out/${tag}_kernel_microbench.txt
timeout 200 python tools/step_stamps.py > gpurun_out/${tag}_step_stamps.txt 2>&1; tail -2 gpurun_out/${tag}_step_stamps.txt
timeout 100 python tools/stn_head_bench.py > gpurun_out/${tag}_stn_head_bench.txt 2>&1
timeout 1150 python -m pytest tests -m gpu -q --timeout=600 --durations=15 2>&1 | tail -30 > gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log
