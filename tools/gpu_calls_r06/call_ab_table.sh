# A/B of one environment switch with the per-layer conv table of each side: usage: call_ab_table.sh VAR A B
mkdir -p gpurun_out
V=$1; A=$2; B=$3
for val in $A $B; do
env $V=$val timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --conv-table gpurun_out/abt_$val.txt > gpurun_out/abt_$val.json 2> gpurun_out/abt_$val.err
done
python tools/compare_conv_tables.py gpurun_out/abt_$A.txt gpurun_out/abt_$B.txt 2>&1 | head -${N:-40}
