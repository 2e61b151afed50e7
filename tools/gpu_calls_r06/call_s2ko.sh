# h8s2 knock-outs (measurement only; library: python tools/gpu_calls_r06/build_s2ko.py): VPS_S2_KO bits 1 no A loads, 2 no A staging, 4 no B loads, 8 no B staging, 16 no barrier, 32 no MFMA
for ko in 0 1 3 12 32 35 47 63; do
echo "KO $ko"; VPS_HIP_LIB=build/s2ko/libvpship.so VPS_S2_KO=$ko BENCH_CONV_REPS=20 BENCH_CONV_FILTER="${F:-fusion conv1 64->64 3x3s2}" timeout 200 python tools/bench_conv.py 4 2>&1 | grep -v "amdgpu.ids\|Warning"
done
