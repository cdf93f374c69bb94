#!/bin/bash
# Round-2 evidence for the resident loop (tile_loop), one GPU box: bench lines, rocprofv3 kernel stats, PMC traffic and SQ passes.
# usage (on the GPU box, repo root): tools/collect_profiles_loop.sh <outdir under gpurun_out>
set -u
out=$1; mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
b="python3 $root/bench.py"
$b --steps 2001 --warmup 101 > "$root/$out/bench_config2.json" 2> "$root/$out/bench_config2.err"
$b --steps 20 --warmup 5 --no-cpu-baseline > "$root/$out/bench_config2_driver_form.json" 2>> "$root/$out/bench_config2.err"
echo "bench done"
# every tile_loop launch of this command takes 513 steps (warm-up, timed repeats, the timed-loop passes)
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace_config2" -o r02 -- python3 "$root/bench.py" --steps 513 --warmup 513 --no-cpu-baseline --repeats 2 > "$root/$out/trace_config2.log" 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$root/$out/pmc_config2_fetch" -o pmc -- python3 "$root/bench.py" --steps 513 --warmup 513 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_config2_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$root/$out/pmc_config2_write" -o pmc -- python3 "$root/bench.py" --steps 513 --warmup 513 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_config2_write.log" 2>&1
echo "traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d "$root/$out/pmc_config2_sq" -o pmc -- python3 "$root/bench.py" --steps 513 --warmup 513 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_config2_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --kernel-trace -d "$root/$out/pmc_config2_sq2" -o pmc -- python3 "$root/bench.py" --steps 513 --warmup 513 --no-cpu-baseline --repeats 1 > "$root/$out/pmc_config2_sq2.log" 2>&1
echo "sq done"
cd "$root" && python3 tools/summarize_pmc_db.py "$out/pmc_summary.json" $(find "$out" -name "*_results.db" | sort) > "$out/pmc_summary.log" 2>&1
find "$out" -name "*.db" -delete
ls "$root/$out"
