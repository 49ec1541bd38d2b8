#!/bin/bash
# GPU box: error of the full-model golden's gradients with components switched
run() { echo "=== $1"; shift; env "$@" python scripts/golden_full_report.py --pruned ${TOG} 2>&1 | grep "grad::" | grep "layers.1.multihead_attn.value_proj.weight\|pos_embed_layer\|encoder.layers.0.norm2\|input_proj.0.0\|stem.conv1" | sed 's/grad::transformer\.//; s/grad:://; s/err\/max//' | awk '{printf "  %-70s %s\n", $1, $2}'; }
TOG="--toggle torch_box --toggle sdp_math" run all_off EFG_FUSED_LN=0 EFG_FUSED_LINEAR=0 EFG_FUSED_GN=0 EFG_FUSED_BN=0 EFG_FUSED_LOSS=0 EFG_GT_GRAPH=0
TOG="--toggle torch_box --toggle sdp_math" run all_off_but_loss EFG_FUSED_LN=0 EFG_FUSED_LINEAR=0 EFG_FUSED_GN=0 EFG_FUSED_BN=0 EFG_GT_GRAPH=0
TOG="--toggle torch_box --toggle sdp_math" run all_off_but_ln EFG_FUSED_LINEAR=0 EFG_FUSED_GN=0 EFG_FUSED_BN=0 EFG_FUSED_LOSS=0 EFG_GT_GRAPH=0
TOG="--toggle torch_box --toggle sdp_math" run all_off_but_linear EFG_FUSED_LN=0 EFG_FUSED_GN=0 EFG_FUSED_BN=0 EFG_FUSED_LOSS=0 EFG_GT_GRAPH=0
