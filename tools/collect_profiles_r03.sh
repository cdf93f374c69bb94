#!/bin/bash
# Round-3 evidence, one GPU box: bench lines, rocprofv3 kernel stats, PMC traffic and SQ passes of the same commands, and the
# two-rank rehearsal of the peer-store halo exchange.
# usage (on the GPU box, repo root): tools/collect_profiles_r03.sh <outdir under gpurun_out>
set -u
out=$1; mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
b="python3 $root/bench.py"
$b --steps 2001 --warmup 101 > "$root/$out/bench_config2.json" 2> "$root/$out/bench_config2.err"
$b --steps 20 --warmup 5 --no-cpu-baseline > "$root/$out/bench_config2_driver_form.json" 2>> "$root/$out/bench_config2.err"
$b --workload config4 --steps 1000 --warmup 50 > "$root/$out/bench_config4.json" 2> "$root/$out/bench_config4.err"
$b --workload config5 --steps 300 --warmup 20 > "$root/$out/bench_config5.json" 2> "$root/$out/bench_config5.err"
APS_NTT=0 $b --workload config5 --steps 300 --warmup 20 > "$root/$out/bench_config5_sweep.json" 2>> "$root/$out/bench_config5.err"
$b --workload config5 --f64 --steps 300 --warmup 20 > "$root/$out/bench_config5_f64.json" 2>> "$root/$out/bench_config5.err"
APS_NTT=0 $b --workload config5 --f64 --steps 100 --warmup 10 > "$root/$out/bench_config5_f64_sweep.json" 2>> "$root/$out/bench_config5.err"
$b --workload hbm --steps 100 --warmup 10 > "$root/$out/bench_hbm.json" 2> "$root/$out/bench_hbm.err"
APS_BENCH_DEVICE=0 $b --gpus 2 --steps 200 --warmup 20 > "$root/$out/bench_config3_two_ranks_one_gpu.json" 2> "$root/$out/bench_two_ranks.err"
APS_BENCH_DEVICE=0 $b --gpus 2 --workload config5 --steps 60 --warmup 10 > "$root/$out/bench_config5_two_ranks_one_gpu.json" 2>> "$root/$out/bench_two_ranks.err"
echo "bench done"
# every tile_loop launch of this command takes 513 steps (warm-up, timed repeats, the timed-loop passes)
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace_config2" -o r03 -- python3 "$root/bench.py" --steps 513 --warmup 513 --no-cpu-baseline --repeats 2 > "$root/$out/trace_config2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace_config5" -o r03 -- python3 "$root/bench.py" --workload config5 --steps 64 --warmup 8 --no-cpu-baseline --repeats 2 > "$root/$out/trace_config5.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace_hbm" -o r03 -- python3 "$root/bench.py" --workload hbm --steps 32 --warmup 8 --no-cpu-baseline --repeats 1 > "$root/$out/trace_hbm.log" 2>&1
echo "trace done"
for wl in config2 config5 hbm; do
  args="--steps 513 --warmup 513"; [ $wl = config5 ] && args="--workload config5 --steps 24 --warmup 8"; [ $wl = hbm ] && args="--workload hbm --steps 16 --warmup 4"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$root/$out/pmc_${wl}_fetch" -o pmc -- python3 "$root/bench.py" $args --no-cpu-baseline --repeats 1 > "$root/$out/pmc_${wl}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$root/$out/pmc_${wl}_write" -o pmc -- python3 "$root/bench.py" $args --no-cpu-baseline --repeats 1 > "$root/$out/pmc_${wl}_write.log" 2>&1
done
echo "traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d "$root/$out/pmc_config2_sq" -o pmc -- python3 "$root/bench.py" --steps 513 --warmup 513 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_config2_sq.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d "$root/$out/pmc_config5_sq" -o pmc -- python3 "$root/bench.py" --workload config5 --steps 24 --warmup 8 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_config5_sq.log" 2>&1
echo "sq done"
cd "$root"
for wl in config2 config5 hbm; do
  python3 tools/summarize_pmc_db.py "$out/pmc_${wl}_summary.json" $(find "$out" -path "*pmc_${wl}_*" -name "*_results.db" | sort) > "$out/pmc_${wl}_summary.log" 2>&1
done
find "$out" -name "*.db" -delete
ls "$root/$out"
