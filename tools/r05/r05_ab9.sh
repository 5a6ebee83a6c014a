#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab9; mkdir -p $O
bash tools/ab_libs.sh "main flags flags1 flags2 flags3s0 flags3s4" 2 > $O/ab_trunkw.txt 2>&1
cat $O/ab_trunkw.txt
