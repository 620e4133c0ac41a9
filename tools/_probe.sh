mkdir -p gpurun_out/r04
O=gpurun_out/r04/ab_epi_ahead.txt
python -m fal_net_amd._build --ab ah0 -DFALNET_DMA_EPI_AHEAD=0 > /dev/null 2>&1
python -m fal_net_amd._build --ab ah1 -DFALNET_DMA_EPI_AHEAD=1 > /dev/null 2>&1
python -m fal_net_amd._build --ab ahm > /dev/null 2>&1
for t in ahm ah0 ah1; do
FALNET_LIB=fal_net_amd/libfalnet_hip_$t.so FALNET_AUTOTUNE_CACHE=$PWD/gpurun_out/r04/cache_$t.json python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --no-roofline --no-trajectory 2>&1 | tail -1 | cut -c1-130
done
for i in 1 2 3; do for t in ahm ah0 ah1; do
FALNET_LIB=fal_net_amd/libfalnet_hip_$t.so FALNET_AUTOTUNE_CACHE=$PWD/gpurun_out/r04/cache_$t.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-trajectory 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value'],1), round(d['ms_per_step'],3), d['config']['final_loss'], flush=True)" >> $O
done; done
cat $O
