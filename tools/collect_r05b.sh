#!/bin/bash
# Runs ON THE GPU BOX: re-collects the rocprofv3 evidence (kernel stats + PMC passes) of the configurations whose launches changed after
# tools/collect_r05.sh ran (the single-tile weight gradients' K slices: PredRNN-V2 training).
export VPX_PROFILES_DST=gpurun_out/r05_profiles
mkdir -p $VPX_PROFILES_DST
for spec in "predrnn_train_b128|--model predrnn-pp --mode train --name predrnn_train_b128" \
            "predrnn_train_b32|--model predrnn-pp --mode train --batch 32 --name predrnn_train_b32" \
            "c5_train_b2|--model predrnn-pp --mode train --batch 2 --img 128 --channels 3 --pred 30 --layers 4 --name c5_train_b2_128x128x3_10to30_L4"; do
  tag=${spec%%|*}; args=${spec#*|}
  bash tools/prof_extra.sh $tag $args >> gpurun_out/collect_r05b.log 2>&1
  name=$(echo "$args" | sed 's/.*--name //')
  python3 tools/summarize_extra.py r05 $tag $name >> gpurun_out/collect_r05b.log 2>&1
  rm -rf gpurun_out/prof_extra/$tag
done
ls -la $VPX_PROFILES_DST | tail -12
