# h8 knock-outs (measurement only; library: python tools/gpu_calls_r06/build_h8ko.py): VPS_H8_KO bits 1 no A loads, 2 no A staging, 4 no B loads, 8 no B staging, 16 no barrier
for ko in 0 1 3 4 12 15 16 31; do
echo "KO $ko"; VPS_HIP_LIB=build/h8ko/libvpship.so VPS_H8P=0 VPS_H8_KO=$ko BENCH_CONV_FILTER="${F:-fpn/tcea 256->256 3x3 @256x512}" timeout 200 python tools/bench_conv.py 4 2>&1 | grep -v amdgpu.ids
done
