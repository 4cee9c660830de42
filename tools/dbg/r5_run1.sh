cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_decode_spec.py tests/test_gpu_codec.py tests/test_gpu_fuzz.py tests/test_gpu_roundtrip_random.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5a/tests.log
for v in 0 1 0 1; do
  echo "scan=$v $(timeout 300 python bench.py --no-cpu --no-host-path --no-pipeline --steps 5 --warmup 2 --opt sf_decode_scan=$v 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f dec %s value %.1f" % (d["encode_ms"], d["decode_ms"], d["value"]))')" >> gpurun_out/r5a/ab.log
done
cat gpurun_out/r5a/tests.log gpurun_out/r5a/ab.log
