#!/bin/bash
# Where a CLI process's fixed cost goes (through gpurun): SHAFA_TRACE of -m c on a 256 MiB file at -b m, and the wall clock of
# -m f / -m c / default run / -m d on it.   usage: tools/dbg/r5_startup.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LD_LIBRARY_PATH=$PWD/shafa-cd_amd:$LD_LIBRARY_PATH
D=/dev/shm/shafa_start; rm -rf $D; mkdir -p $D
python3 - <<'PY'
import sys; sys.path.insert(0,'tests')
import pkgload
synth = pkgload.load_submodule("synth")
synth.gen_bytes(77, 256 << 20, synth.zipf_mod256_table(1.2)).tofile("/dev/shm/shafa_start/z")
PY
O=shafa-cd_amd/bin/shafa
t() { s=$(date +%s%N); "$@" > /dev/null 2>$D/err; e=$(date +%s%N); echo "$(( (e-s)/1000000 )) ms  : $*"; }
for b in m M; do
  echo "== -b $b"
  t $O $D/z -m f -b $b
  t $O $D/z.freq -m t
  t $O $D/z -m c
  t env SHAFA_FTC=2 $O $D/z -b $b
  cp $D/z $D/z.orig; rm $D/z
  t $O $D/z.shaf -m d
  cmp $D/z $D/z.orig && echo "round trip ok"
  rm -f $D/z.shaf $D/z.cod $D/z.freq
done
$O $D/z -m f -b m > /dev/null; $O $D/z.freq -m t > /dev/null
SHAFA_TRACE=1 $O $D/z -m c 2>&1 >/dev/null | head -40
rm -rf $D
