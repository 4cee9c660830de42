cd ${GRAFT_REPO_ROOT:-/root/repo}
export SHAFA_BENCH_ORACLE_CHECK=0
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for i in 1 2; do for L in base C128 C512; do cp _ab/$L.so shafa-cd_amd/libshafa_hip.so
echo "$L $(python bench.py --no-cpu --no-host-path --pipeline-only --pipeline-kind runs --steps 8 2>&1 | tail -1 | python3 -c "
import json,sys
p=json.loads(sys.stdin.read())['pipeline']
print('  '.join('%s %.3f ms %.3f' % (k[:2], v['ms'], v['frac']) for k,v in p.items() if isinstance(v, dict) and k[:2] in ('K1','K2')))")"
done; done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
