#!/bin/bash
# decode parity tests, then bench.py with sf_decode_emit 1 / 0 alternating, then kernel stats.  usage: tools/dbg/r5_dec2.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_decode_spec.py tests/test_gpu_codec.py tests/test_gpu_fuzz.py tests/test_gpu_roundtrip_random.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -12
for v in 1 0 1 0; do
  echo "emit=$v $(timeout 300 python bench.py --no-cpu --no-host-path --no-pipeline --steps 5 --warmup 2 --opt sf_decode_emit=$v 2>&1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f dec %.3f value %.1f" % (d["encode_ms"], d["decode_ms"], d["value"]))' 2>&1 | tail -1)"
done
bash tools/gpu_kstats.sh --no-pipeline --steps 5 --warmup 2 2>&1 | grep -i "sfd_"
