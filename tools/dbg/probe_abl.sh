cd $GRAFT_REPO_ROOT
cp shafa-cd_amd/libshafa_hip.so /tmp/orig.so
for L in _ab/a.so _ab/nooct.so; do cp $L shafa-cd_amd/libshafa_hip.so; echo "== $L"; python tools/dbg/lut_conflict_probe.py 5 2>&1 | tail -3; done
cp /tmp/orig.so shafa-cd_amd/libshafa_hip.so
