#!/bin/bash
# Round-2 evidence, one GPU box: bench lines, rocprofv3 kernel stats and PMC traffic passes of the same commands.
# usage (on the GPU box, repo root): tools/collect_profiles.sh <outdir under gpurun_out>
set -u
out=$1; mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
b="python3 $root/bench.py"
$b --steps 2000 --warmup 100 > "$root/$out/bench_config2.json" 2> "$root/$out/bench_config2.err"
$b --steps 20 --warmup 5 --no-cpu-baseline > "$root/$out/bench_config2_driver_form.json" 2>> "$root/$out/bench_config2.err"
$b --workload config4 --steps 1000 --warmup 50 > "$root/$out/bench_config4.json" 2> "$root/$out/bench_config4.err"
$b --workload config5 --steps 300 --warmup 20 > "$root/$out/bench_config5.json" 2> "$root/$out/bench_config5.err"
$b --workload hbm --steps 100 --warmup 10 > "$root/$out/bench_hbm.json" 2> "$root/$out/bench_hbm.err"
# kernel trace + stats of the default bench command (short: the summary is per-kernel averages)
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace_config2" -o r02 -- python3 "$root/bench.py" --steps 512 --warmup 64 --no-cpu-baseline --repeats 2 > "$root/$out/trace_config2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace_config5" -o r02 -- python3 "$root/bench.py" --workload config5 --steps 64 --warmup 8 --repeats 2 > "$root/$out/trace_config5.log" 2>&1
# HBM-side traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), counters only with --kernel-trace
for wl in config2 config4 config5; do
  steps=64; [ $wl = config5 ] && steps=24
  extra=""; [ $wl != config2 ] && extra="--workload $wl"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$root/$out/pmc_${wl}_fetch" -o pmc -- python3 "$root/bench.py" $extra --steps $steps --warmup 8 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_${wl}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$root/$out/pmc_${wl}_write" -o pmc -- python3 "$root/bench.py" $extra --steps $steps --warmup 8 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_${wl}_write.log" 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d "$root/$out/pmc_config2_sq" -o pmc -- python3 "$root/bench.py" --steps 64 --warmup 8 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_config2_sq.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d "$root/$out/pmc_config5_sq" -o pmc -- python3 "$root/bench.py" --workload config5 --steps 24 --warmup 8 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_config5_sq.log" 2>&1
ls "$root/$out"
