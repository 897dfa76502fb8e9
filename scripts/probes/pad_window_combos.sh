#!/bin/bash
for combo in "10240 4" "10240 1" "0 1" "0 4"; do set -- $combo; echo "## ZL_K2_LDS_PAD=$1 ZL_WINDOW_MUL=$2"; ZL_K2_LDS_PAD=$1 ZL_WINDOW_MUL=$2 bash scripts/ab_libs.sh "base" "" "--notes 48,72" "--notes 48,72 --hermite" "--frames 128" "--voices 4096 --buses 32 --fs 96000 --blocks-per-step 3750" "--loop-seconds 50"; done
