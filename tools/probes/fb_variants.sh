# the fused boundary kernel alone in library variants build/ab/lib_<name>.so: FB_LIBS="fd1 fd2" bash tools/probes/fb_variants.sh
LIB=image-caption-emotion-indonesia_amd/libcapnet_hip.so
cp $LIB build/ab/lib__orig.so
for v in $FB_LIBS; do cp build/ab/lib_$v.so $LIB; echo "== $v"; timeout -k 10 120 python tools/probes/fused_block_bench.py 2>&1 | grep stage | cut -c1-110; done
cp build/ab/lib__orig.so $LIB
