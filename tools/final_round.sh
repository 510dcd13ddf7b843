#!/bin/bash
# Round-end artefacts in ONE gpurun call: bench lines of every configuration, one rocprofv3 kernel trace + step timeline, the kernel
# micro-benchmarks, PMC passes of the kernels touched this round, then the whole -m gpu suite.  usage: tools/final_round.sh <tag>
tag=${1:-r03}
mkdir -p gpurun_out; export TMPDIR=/tmp
b() { name=$1; shift; timeout 300 python bench.py --steps 30 --warmup 10 "$@" > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err; echo "bench $name rc=$? $(cut -c1-220 gpurun_out/${tag}_bench_${name}.json)"; }
b std
b large --tile large
b tsrn --arch tsrn --no-cpu-baseline
b tbsrn --arch tbsrn --no-cpu-baseline
b tpg --arch tatt_tpg
b tssim --tssim --no-cpu-baseline
b dp_selftest --dp-selftest --no-cpu-baseline
tools/gpu_quick.sh ${tag}_final none "prof:" > /dev/null 2>&1
head -3 gpurun_out/${tag}_final_timeline.txt
timeout 200 python tools/bench_kernels.py > gpurun_out/${tag}_kernel_microbench.txt 2>&1; tail -3 gpurun_out/${tag}_kernel_microbench.txt
for k in conv9_fwd_mfma_64_4_hr gru_wgrad_sb_G128; do
  tools/pmc_collect.sh ${tag} $k > /dev/null 2>&1; echo "pmc $k: $(wc -l < gpurun_out/${tag}_pmc_${k}.txt) lines"
done
timeout 560 python -m pytest tests -m gpu -q --timeout=500 2>&1 | tail -15 > gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log
