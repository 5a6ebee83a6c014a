#!/bin/bash
# Run ON THE MI355X BOX (through gpurun): everything profiles/ holds for one build.
#   tools/round_profiles.sh r02_c    -> gpurun_out/profiles_<tag>/
set -u
TAG=${1:-r02_x}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
exec < /dev/null                        # nothing here may wait for a terminal (a rocm-smi prompt once cost a call its whole limit)
export UVA_DEBUG_SWITCHES=1             # the A/B lines below use the library's debug switches (ignored without the opt-in)
bash "$REPO/tools/collect_profiles.sh" "$TAG" > "$OUT/collect.log" 2>&1
cd "$REPO"
# same-box A/B against earlier kernels shipped beside the library: upscale_video_amd/libuva_prev.so = round 4's trunk kernel
# (-DTW_RAW_INK=0 -DTW_PFF=6 -DTW_DMA_B=0 -DTW_PRE_BAR=0 -DUVA_NO_PK_F32= and, by UVA_TW_FOLD=0, its schedule without folded strips),
# libuva_mid.so = the kernel of this round's first evidence set (-DTW_DMA_B=0 -DTW_PRE_BAR=0 -DUVA_NO_PK_F32=)
if [ -f upscale_video_amd/libuva_prev.so ]; then
  P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["kernel_ms_per_frame"], d["roofline"]["frac"])'
  VS="prev new"; [ -f upscale_video_amd/libuva_mid.so ] && VS="prev mid new"
  for wl in 2x_compact_1080p 4x_compact_1080p; do
    for i in 1 2 3; do
      for v in $VS; do
        L=$REPO/upscale_video_amd/libuva.so; [ $v != new ] && L=$REPO/upscale_video_amd/libuva_$v.so
        F=1; [ $v = prev ] && F=0
        echo -n "$wl $v: "; UVA_TW_FOLD=$F UVA_LIB_PATH=$L python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "$P"
      done
    done
  done > "$OUT/${TAG}_ab_prev_new.txt" 2>&1
fi
for wl in 4x_compact_1080p 1x_hurrdeblur_1080p chain_1x_2x_1080p 2x_compact_2160p; do
  python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_${wl}.json" 2>> "$OUT/bench.err"
done
python bench.py --workload 4x_valar_1080p --steps 30 --warmup 3 > "$OUT/${TAG}_bench_4x_valar_1080p.json" 2>> "$OUT/bench.err"
python bench.py --tile 0 --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_2x_whole_frame.json" 2>> "$OUT/bench.err"
UVA_TRUNK_FUSION=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_unfused_trunk_kernel.json" 2>> "$OUT/bench.err"
UVA_TRUNK_WINO=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_direct_trunk2_kernel.json" 2>> "$OUT/bench.err"
UVA_TW_ACT16=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_trunkw_fp32_prelu.json" 2>> "$OUT/bench.err"
python bench.py --gpus 2 --devices 0,0 --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_two_ranks_one_gpu.json" 2>> "$OUT/bench.err"
UVA_SUB10=0 python bench.py --workload 1x_hurrdeblur_1080p --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_1x_per_pair_kernels.json" 2>> "$OUT/bench.err"
# round 5: the 1x net as two launches of five layers (sub5_kernel), folded last strips off, the driver's form of the default line
UVA_SUB5=1 python bench.py --workload 1x_hurrdeblur_1080p --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_1x_sub5_two_launches.json" 2>> "$OUT/bench.err"
UVA_TW_FOLD=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_no_folded_strips.json" 2>> "$OUT/bench.err"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_form.json" 2>> "$OUT/bench.err"
python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --workload 2x_compact_2160p --steps 8 --warmup 2 --no-cpu-baseline --no-parity > "$OUT/${TAG}_bench_config5_eight_ranks_one_gpu.json" 2>> "$OUT/bench.err"
# round 6: the 1x net one, two and four frames per launch (uva_net_process_u8_device_batch; the workload's default is four), two
# ranks on the one GPU with the static and the dynamic frame queue (per-rank records), the driver's N > 1 form with one rank
for b in 1 2 4; do python bench.py --workload 1x_hurrdeblur_1080p --batch $b --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/${TAG}_bench_1x_batch$b.json" 2>> "$OUT/bench.err"; done
python bench.py --gpus 2 --devices 0,0 --dynamic --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_two_ranks_one_gpu_dynamic.json" 2>> "$OUT/bench.err"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --dynamic --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_bench_torchrun_one_rank_dynamic.json" 2>> "$OUT/bench.err"
python tools/png_decode_split.py 9 > "$OUT/${TAG}_png_decode_split.txt" 2>&1
python tools/pipe_bench.py 200 > "$OUT/${TAG}_pipe_bench.txt" 2>&1
python bench.py --workload 1x_hurrdeblur_1080p --tile 960 --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_1x_tiled_960.json" 2>> "$OUT/bench.err"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/prof1x_$TAG -o p --output-format csv -- python $REPO/bench.py --workload 1x_hurrdeblur_1080p --batch 1 --tile 0 --steps 120 --warmup 10 --no-cpu-baseline --no-parity > /dev/null 2>&1; cp $(find /tmp/prof1x_$TAG -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_kernel_stats_1x_rocprofv3.csv")
bash tools/pmc_sub10.sh /tmp/pmc_sub10_$TAG > "$OUT/${TAG}_sub10_pmc.txt" 2>&1
python tools/png_route_bench.py 384 > "$OUT/${TAG}_png_route_bench.txt" 2>&1
python tools/png_gpu_route_bench.py 240 > "$OUT/${TAG}_png_gpu_route_bench.txt" 2>&1
python tools/rawvideo_bench.py 400 > "$OUT/${TAG}_rawvideo_bench.txt" 2>&1
python tools/denoise_bench.py > "$OUT/${TAG}_denoise_bench_now.txt" 2>&1
python tools/valar_bench.py 3 > "$OUT/${TAG}_bench_valar.txt" 2>&1
UVA_GENERIC_RDB=0 UVA_GENERIC_SW=0 python tools/valar_bench.py 3 2>&1 | sed 's/^/layer by layer (UVA_GENERIC_RDB=0 UVA_GENERIC_SW=0): /' >> "$OUT/${TAG}_bench_valar.txt"
for kv in UVA_GENERIC_BATCH=0 UVA_GENERIC_FUSE_INTERP=0 UVA_GENERIC_SK=1; do
  env $kv python tools/valar_bench.py 3 2>&1 | grep -v amdgpu.ids | sed "s/^/$kv: /" >> "$OUT/${TAG}_bench_valar.txt"
done
hipcc --offload-arch=gfx950 -O3 tools/mfma_read_ratio_bench.hip -o /tmp/mrr_$TAG 2> /dev/null && /tmp/mrr_$TAG > "$OUT/${TAG}_mfma_read_ratio_bench.txt" 2>&1
hipcc --offload-arch=gfx950 -O3 tools/hbm_stream_bench.hip -o /tmp/hsb_$TAG 2> /dev/null && /tmp/hsb_$TAG > "$OUT/${TAG}_hbm_stream_bench.txt" 2>&1
bash tools/pmc_valar.sh > "$OUT/${TAG}_valar_pmc.txt" 2>&1
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/profval_$TAG -o p --output-format csv -- python $REPO/tools/valar_bench.py 3 > /dev/null 2>&1; cp $(find /tmp/profval_$TAG -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_kernel_stats_valar_rocprofv3.csv")
UVA_RDB_STAMPS=1 UVA_LIB_PATH=$REPO/upscale_video_amd/libuva_instr.so python tools/rdb4_anatomy.py > "$OUT/${TAG}_rdb4_anatomy.txt" 2>&1
python tools/valar_bench.py 60 > "$OUT/power_valar.txt" 2>/dev/null &
sleep 5
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; sleep 1; done > "$OUT/${TAG}_power_during_valar.txt"
wait
grep frames "$OUT/power_valar.txt" >> "$OUT/${TAG}_power_during_valar.txt"
python test_gpus.py -g 0,0,0,0 -s 2 -r 16 > "$OUT/${TAG}_test_gpus_harness.txt" 2>&1
# package power and shader clock while the bench runs (sustained state)
python bench.py --steps 6000 --warmup 50 --no-cpu-baseline > "$OUT/power_bench.json" 2>/dev/null &
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; sleep 1; done > "$OUT/${TAG}_power_during_bench.txt"
wait
sleep 8
python -c "import json; d=json.load(open('$OUT/power_bench.json')); print('bench --steps 6000:', d['value'], 'fps, trunk', d['config']['kernel_ms_per_frame']['trunk'], 'ms/frame, frac', d['roofline']['frac'])" >> "$OUT/${TAG}_power_during_bench.txt" 2>&1
UVA_LIB_PATH=$REPO/upscale_video_amd/libuva_instr.so python tools/trunkw_anatomy.py > "$OUT/${TAG}_trunkw_anatomy.txt" 2>&1
UVA_TRUNK_WINO=0 UVA_LIB_PATH=$REPO/upscale_video_amd/libuva_instr.so python tools/trunk2_anatomy.py > "$OUT/${TAG}_trunk2_anatomy.txt" 2>&1
UVA_LIB_PATH=$REPO/upscale_video_amd/libuva_instr.so python tools/sub10_anatomy.py > "$OUT/${TAG}_sub10_anatomy.txt" 2>&1
UVA_SUB5=1 UVA_LIB_PATH=$REPO/upscale_video_amd/libuva_instr.so python tools/sub5_anatomy.py > "$OUT/${TAG}_sub5_anatomy.txt" 2>&1
python tools/soak.py 1000 > "$OUT/${TAG}_soak.txt" 2>&1
timeout 2400 python -m pytest tests -m gpu -q > "$OUT/${TAG}_gpu_tests.txt" 2>&1
cp gpurun_out/parity_report.json "$OUT/${TAG}_parity_report.json" 2>/dev/null
cp gpurun_out/build_on_gpu_box.json "$OUT/${TAG}_build_on_gpu_box.json" 2>/dev/null
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke: ok')" 2>&1 | tail -3) > "$OUT/${TAG}_smoke.txt"
ls -la "$OUT"
