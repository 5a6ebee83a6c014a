#!/bin/bash
# Run ON THE MI355X BOX: the 1x net's evidence on ONE box -- bench lines and rocprofv3 kernel statistics of the old behaviour
# (libuva_old10.so: -DS10_BAL=0 -DS10_ROWSKIP=0) and of the default, the default's SQ counters and stamps
R="$(cd "$(dirname "$0")/.." && pwd)"; cd $R
O=$R/gpurun_out/r05_ab26; mkdir -p $O
U=$R/upscale_video_amd
bash tools/ab_libs.sh "old10 main" 3 "1x_hurrdeblur_1080p" > $O/ab_1x.txt 2>&1
for v in old10 main; do
  L=$U/libuva_$v.so; [ $v = main ] && L=$U/libuva.so
  UVA_LIB_PATH=$L python bench.py --workload 1x_hurrdeblur_1080p --steps 100 --warmup 10 > $O/bench_1x_$v.json 2>> $O/bench.err
  (cd /tmp && TMPDIR=/tmp UVA_LIB_PATH=$L rocprofv3 --kernel-trace --stats -d /tmp/prof1x_$v -o p --output-format csv -- python $R/bench.py --workload 1x_hurrdeblur_1080p --tile 0 --steps 120 --warmup 10 --no-cpu-baseline --no-parity > /dev/null 2>&1; cp $(find /tmp/prof1x_$v -name "*kernel_stats.csv" | head -1) $O/kernel_stats_1x_${v}_rocprofv3.csv)
done
bash tools/pmc_sub10.sh /tmp/pmc_sub10_ab26 > $O/sub10_pmc.txt 2>&1
UVA_LIB_PATH=$U/libuva_instr.so python tools/sub10_anatomy.py > $O/sub10_anatomy.txt 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_2x_same_box.json 2>> $O/bench.err
cat $O/ab_1x.txt; head -2 $O/kernel_stats_1x_old10_rocprofv3.csv | cut -c1-120; head -2 $O/kernel_stats_1x_main_rocprofv3.csv | cut -c1-120; cat $O/sub10_pmc.txt | tail -26
