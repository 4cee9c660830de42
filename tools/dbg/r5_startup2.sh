#!/bin/bash
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
$O $D/z -m f -b m > /dev/null; $O $D/z.freq -m t > /dev/null
for i in 1 2; do
s=$(date +%s%N); SHAFA_TRACE=1 $O $D/z -m c 2>$D/tr >/dev/null; e=$(date +%s%N); echo "total $(( (e-s)/1000000 )) ms"; tail -4 $D/tr
done
which strace ltrace perf 2>/dev/null
s=$(date +%s%N); AMD_LOG_LEVEL=0 HIP_VISIBLE_DEVICES=0 $O $D/z -m c >/dev/null 2>&1; e=$(date +%s%N); echo "HIP_VISIBLE_DEVICES=0 total $(( (e-s)/1000000 )) ms"
s=$(date +%s%N); python3 -c "
import ctypes,time
t=time.time(); l=ctypes.CDLL('libamdhip64.so'); print('dlopen hip', round((time.time()-t)*1e3),'ms')
t=time.time(); n=ctypes.c_int(); l.hipGetDeviceCount(ctypes.byref(n)); print('hipGetDeviceCount', round((time.time()-t)*1e3),'ms', n.value)
t=time.time(); l.hipSetDevice(0); l.hipFree(0); print('context', round((time.time()-t)*1e3),'ms')
p=ctypes.c_void_p(); t=time.time(); l.hipHostMalloc(ctypes.byref(p), 64<<20, 0); print('hipHostMalloc 64M', round((time.time()-t)*1e3),'ms')
t=time.time(); l.hipHostMalloc(ctypes.byref(p), 64<<20, 0); print('hipHostMalloc 64M again', round((time.time()-t)*1e3),'ms')
t=time.time(); l.hipMalloc(ctypes.byref(p), 64<<20); print('hipMalloc 64M', round((time.time()-t)*1e3),'ms')
t=time.time(); l2=ctypes.CDLL('$PWD/shafa-cd_amd/libshafa_hip.so'); print('dlopen shafa_hip', round((time.time()-t)*1e3),'ms')
"
rm -rf $D
