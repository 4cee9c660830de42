cd $GRAFT_REPO_ROOT
export SHAFA_BENCH_ORACLE_CHECK=0
run() { python bench.py --no-cpu --no-host-path --no-pipeline --encode-only --steps 8 "$@" 2>&1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("enc %.3f ms frac %.3f ratio %.3f" % (d["encode_ms"], d["roofline_encode"]["frac"], d["config"]["compressed_ratio"]))' 2>&1 | tail -1; }
for W in "--zipf-s 2.0" "--dist zipf --zipf-s 2.0" "--zipf-s 1.6" "--zipf-s 2.0 --blocks 32"; do
  echo "[$W] wide+redo: $(run $W)"
  echo "[$W] 256 lanes: $(run $W --opt sf_encode_lanes=256)"
done
