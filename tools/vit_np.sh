#!/bin/bash
# Runs ON THE GPU BOX: the Viterbi forward kernel at each workgroup shape (RELATE_VIT_NP = 4, 8 or 16 pairs a workgroup)
# usage: bash tools/vit_np.sh [pairs]
P=${1:-5120}
for np in 4 8 16; do
  RELATE_VIT_NP=$np bash tools/trace_tool.sh vitnp$np tools/auxbench.py --skip-linkage --pairs $P | grep viterbi_fwd || exit 1
done
