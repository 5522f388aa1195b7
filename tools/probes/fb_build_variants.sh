# Library variants that differ in fused_block.hip's compile-time switches, for tools/probes/fb_variants.sh (run HERE, hipcc
# cross-compiles):   FB_DEFS="dbg1:-DCAPNET_FB_DBG=1 dbg3:-DCAPNET_FB_DBG=3 dbg4:-DCAPNET_FB_DBG=4 fd2:-DCAPNET_FB_FD=2" bash tools/probes/fb_build_variants.sh
# -> build/ab/lib_<name>.so (+ lib_base.so = the shipped library). CAPNET_FB_DBG: 1 no MFMAs, 2 no weight DMA, 3 neither MFMAs
# nor fragment reads, 4 = 3 without the DMA; CAPNET_FB_FD: batches of fragments read ahead; CAPNET_FB_GB8: groups per batch.
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
cd $ROOT/image-caption-emotion-indonesia_amd/csrc
make -j8 > /dev/null
mkdir -p $ROOT/build/ab
cp ../libcapnet_hip.so $ROOT/build/ab/lib_base.so
FLAGS=$(grep '^CXXFLAGS' Makefile | sed 's/.*:= //; s/$(ARCH)/gfx950/')
OBJS=$(grep -A1 '^SRCS' Makefile | sed 's/SRCS *:= //; s/\\//' | tr ' ' '\n' | grep -v '^$' | grep -v fused_block | sed "s|^|$ROOT/build/obj/|; s|$|.o|")
for v in $FB_DEFS; do
  name=${v%%:*}; defs=$(echo ${v#*:} | tr ',' ' ')
  /opt/rocm/bin/hipcc $FLAGS $defs -x hip -c fused_block.hip -o $ROOT/build/ab/fused_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/ab/lib_$name.so $OBJS $ROOT/build/ab/fused_$name.o -ldl
  rm -f $ROOT/build/ab/fused_$name.o
  echo built build/ab/lib_$name.so
done
