#!/bin/bash
# Round 5, GPU call 8 (ON THE BOX): at which fragment the consumers' in-stream transform starts
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab7; mkdir -p $O
bash tools/ab_libs.sh "main rawf2 rawf4 rawf6 rawf8 rawf4s2" 3 > $O/ab_trunkw.txt 2>&1
cat $O/ab_trunkw.txt
