#!/bin/bash
ulimit -c 0
# Round 5, session 18: phase 3 with the pairs exchanged as 8-byte agent-scope atomics: region comparison, parity, step A/B
bash tools/sessions/r5/gpu_r5_s17.sh $1 2>&1 | grep -E "mean_f|rstd_f|xn_f|ffw_out|status"
unset FLAMINGO_FUSION_LIB
bash tools/sessions/r5/gpu_r5_s16.sh $1
