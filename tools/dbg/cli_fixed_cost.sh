cd ${GRAFT_REPO_ROOT:-/root/repo}
export LD_LIBRARY_PATH=$PWD/shafa-cd_amd:$LD_LIBRARY_PATH
D=/dev/shm/shafa_small; rm -rf $D; mkdir -p $D
head -c 3000000 /dev/urandom | tr -c 'a-f' 'a' > $D/s
O=shafa-cd_amd/bin/shafa
$O $D/s -m f -b M > /dev/null; $O $D/s.freq -m t > /dev/null
for i in 1 2 3; do s=$(date +%s%N); SHAFA_TRACE=1 $O $D/s -m c 2>$D/tr >/dev/null; e=$(date +%s%N); echo "small -m c total $(( (e-s)/1000000 )) ms; $(grep -c . $D/tr) trace lines; last: $(tail -1 $D/tr)"; done
s=$(date +%s%N); $O $D/s.freq -m t > /dev/null; e=$(date +%s%N); echo "-m t (no GPU) total $(( (e-s)/1000000 )) ms"
s=$(date +%s%N); /bin/true; e=$(date +%s%N); echo "/bin/true $(( (e-s)/1000000 )) ms"
for i in 1 2; do s=$(date +%s%N); SHAFA_TRACE=1 $O $D/s -m f -b M 2>$D/tr > /dev/null; e=$(date +%s%N); echo "small -m f total $(( (e-s)/1000000 )) ms; last: $(tail -1 $D/tr)"; done
ldd $O | head -20
rm -rf $D
