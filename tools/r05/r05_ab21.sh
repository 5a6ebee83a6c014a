#!/bin/bash
# Run ON THE MI355X BOX: wave priorities under the new balance (k-loop / elsewhere; per group)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab21; mkdir -p $O
bash tools/ab_libs.sh "main pk1 pk3 pb pe1 pa pinv" 3 > $O/ab_trunkw.txt 2>&1
cat $O/ab_trunkw.txt
