# Builds tools/phase_stamps/libvda_hip_stamped.so: the library's own objects with the time-stamped direct convolution in place of
# conv_direct.o (run `make -C monocular-depth-estimation_amd/csrc` first).  Then, on the GPU box:
#   VDA_HIP_LIB=$PWD/tools/phase_stamps/libvda_hip_stamped.so python tools/phase_stamps/conv_phases.py
set -e
cd "$(dirname "$0")"
python make_conv_source.py
C=../../monocular-depth-estimation_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -I$C/csrc -I$C/../include -c conv_direct_stamped.hip -o conv_direct_stamped.o 2>&1 | grep -v "recognized feature" || true
O=$(ls $C/lib/obj/*.o | grep -v conv_direct.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvda_hip_stamped.so $O conv_direct_stamped.o
rm -f conv_direct_stamped.o
