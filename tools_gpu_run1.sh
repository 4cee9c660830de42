#!/bin/bash
# first GPU validation: parity tests, short encode bench, kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > gpurun_out/gpuinfo.txt 2>&1
nproc >> gpurun_out/gpuinfo.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --blocks 32 --steps 5 --warmup 2 --encode-only --no-cpu > gpurun_out/bench_enc_zipf.json 2> gpurun_out/bench_enc_zipf.err
timeout 600 python bench.py --blocks 32 --steps 5 --warmup 2 --encode-only --no-cpu --dist uniform > gpurun_out/bench_enc_uniform.json 2> gpurun_out/bench_enc_uniform.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof1" -- python3 "$GRAFT_REPO_ROOT/bench.py" --blocks 32 --steps 5 --warmup 2 --encode-only --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof1.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof1 -name "*stats*" | head; 
for f in $(find gpurun_out/prof1 -name "*kernel_stats.csv"); do head -12 $f; done
cat gpurun_out/pytest_gpu.log | tail -30
cat gpurun_out/bench_enc_zipf.json gpurun_out/bench_enc_uniform.json
tail -5 gpurun_out/bench_enc_zipf.err
