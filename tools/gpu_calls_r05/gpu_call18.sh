mkdir -p gpurun_out
timeout 600 python tools/run_vps_synthetic.py --height 1024 --width 2048 --videos 2 --frames 30 --prec f16x3 --gt-prec f32 --out gpurun_out/vps_near_tied > gpurun_out/r05_vpq_f16x3_vs_f32_1024x2048_near_tied.json 2> gpurun_out/c18.err; tail -1 gpurun_out/r05_vpq_f16x3_vs_f32_1024x2048_near_tied.json | head -c 600; echo
timeout 600 python tools/run_vps_synthetic.py --height 1024 --width 2048 --videos 2 --frames 30 --prec f16x3 --gt-prec f32 --separated --out gpurun_out/vps_separated > gpurun_out/r05_vpq_f16x3_vs_f32_1024x2048_separated.json 2>> gpurun_out/c18.err; tail -1 gpurun_out/r05_vpq_f16x3_vs_f32_1024x2048_separated.json | head -c 600; echo
rm -rf gpurun_out/vps_near_tied gpurun_out/vps_separated
timeout 600 python tools/exp_two_clips.py --frames 40 > gpurun_out/r05_two_clips_final.json 2>> gpurun_out/c18.err; cat gpurun_out/r05_two_clips_final.json
