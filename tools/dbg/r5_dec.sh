#!/bin/bash
# decode-side check on the GPU box (through gpurun): the decoder's parity tests, then bench.py twice, then the profile summary
# of the decode kernels.   usage: tools/dbg/r5_dec.sh [tag]
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r5dec}
timeout 900 python -m pytest tests/test_gpu_decode_spec.py tests/test_gpu_codec.py tests/test_gpu_fuzz.py tests/test_gpu_roundtrip_random.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
  timeout 300 python bench.py --no-cpu --no-host-path --no-pipeline --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f dec %.3f value %.1f frac_dec %s" % (d["encode_ms"], d["decode_ms"], d["value"], d.get("roofline_decode")))'
done
bash tools/gpu_prof.sh $tag --no-pipeline --steps 5 --warmup 2 2>&1 | grep -A2 "^sfd_scan<false\|^sfd_wstage<\|bytes per input" | grep -v "^--"
