#!/bin/bash
# generic GPU run: parity tests (all, no -x), smoke, then optional extra commands passed as args
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
for cmd in "$@"; do echo "=== $cmd"; timeout 400 bash -c "$cmd"; done
