tag=r03
b() { name=$1; shift; timeout 200 python bench.py --steps 30 --warmup 10 "$@" > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err; echo "bench $name rc=$? $(cut -c1-200 gpurun_out/${tag}_bench_${name}.json)"; }
b std
b large --tile large --no-cpu-baseline
b tsrn --arch tsrn --no-cpu-baseline
b tpg --arch tatt_tpg
b tssim --tssim --no-cpu-baseline
b dp_selftest --dp-selftest --no-cpu-baseline
tools/gpu_quick.sh ${tag}_final none "prof:" > /dev/null 2>&1
head -3 gpurun_out/${tag}_final_timeline.txt
