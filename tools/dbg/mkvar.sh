#!/bin/bash
# Build variants of libshafa_hip.so that differ in the flags one source file is compiled with (HERE, before gpurun; the
# .so files travel with the snapshot):  tools/dbg/mkvar.sh <file.hip> NAME:"-DFLAG ..." ...   ->  _ab/NAME.so (+ _ab/base.so)
R=/root/repo
C=$R/shafa-cd_amd/csrc
src=$1; shift
mkdir -p $R/_ab
for v in "$@"; do
  name=${v%%:*}; fl=${v#*:}
  touch $C/$src
  make -C $C --no-print-directory FLAGS_EXTRA="$fl" 2>&1 | grep -E "error|warning: unused|Error"
  cp $R/shafa-cd_amd/libshafa_hip.so $R/_ab/$name.so
done
touch $C/$src
make -C $C --no-print-directory 2>&1 | grep -E "error|Error"
cp $R/shafa-cd_amd/libshafa_hip.so $R/_ab/base.so
md5sum $R/_ab/*.so | cut -c1-8,33-
