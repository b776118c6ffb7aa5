#!/bin/bash
# Runs ON THE GPU BOX: the round's rocprofv3 evidence in one call (kernel stats + PMC passes of the headline, then of the extras that
# changed this round). Summaries: tools/summarize_profiles.py r06; tools/summarize_extra.py r06 <tag> <name> for each tag below.
export VPX_PROFILES_DST=gpurun_out/r06_profiles
mkdir -p $VPX_PROFILES_DST
bash tools/collect_profiles.sh > gpurun_out/collect_r06.log 2>&1
python3 tools/summarize_profiles.py r06 >> gpurun_out/collect_r06.log 2>&1
rm -rf gpurun_out/prof_final
for spec in "cell_64x64x64_b128|--cell 64,64,64,64 --batch 128 --name cell_64x64x64_b128" \
            "cell_64x64x64_b128_bf16|--cell 64,64,64,64 --batch 128 --precision bf16 --name cell_64x64x64_b128_bf16" \
            "infer_b128_bf16|--precision bf16 --name infer_b128_bf16" \
            "predrnn_infer_b128|--model predrnn-pp --name predrnn_infer_b128" \
            "predrnn_train_b128|--model predrnn-pp --mode train --name predrnn_train_b128" \
            "c5_infer_b4|--model predrnn-pp --batch 4 --img 128 --channels 3 --pred 30 --layers 4 --name c5_infer_b4_128x128x3_10to30_L4" \
            "c5_train_b2|--model predrnn-pp --mode train --batch 2 --img 128 --channels 3 --pred 30 --layers 4 --name c5_train_b2_128x128x3_10to30_L4" \
            "infer_b4|--batch 4 --name infer_b4" \
            "c4_infer_b4|--batch 4 --img 128 --channels 3 --pred 20 --name c4_infer_b4_128x128x3_10to20" \
            "c4_train_b4|--mode train --batch 4 --img 128 --channels 3 --pred 20 --name c4_train_b4_128x128x3_10to20" \
            "infer_b32|--batch 32 --name infer_b32" \
            "train_b32|--mode train --batch 32 --name train_b32" \
            "infer_b128_f32|--precision f32 --name infer_b128_f32" \
            "predrnn_train_b32|--model predrnn-pp --mode train --batch 32 --name predrnn_train_b32" \
            "cell_64x64x64_b32|--cell 64,64,64,64 --batch 32 --name cell_64x64x64_b32" \
            "cell_64x64x64_b4|--cell 64,64,64,64 --batch 4 --name cell_64x64x64_b4" \
            "cell_enc1_16x64x64_b128|--cell 16,64,64,64 --batch 128 --name cell_enc1_16x64x64_b128" \
            "cell_enc2_64x96x32_b128|--cell 64,96,32,32 --batch 128 --name cell_enc2_64x96x32_b128" \
            "cell_enc3_96x96x16_b128|--cell 96,96,16,16 --batch 128 --name cell_enc3_96x96x16_b128" \
            "cell_fore2_96x96x32_b128|--cell 96,96,32,32 --batch 128 --name cell_fore2_96x96x32_b128" \
            "cell_fore1_96x64x64_b128|--cell 96,64,64,64 --batch 128 --name cell_fore1_96x64x64_b128"; do
  tag=${spec%%|*}; args=${spec#*|}
  bash tools/prof_extra.sh $tag $args >> gpurun_out/collect_r06.log 2>&1
  name=$(echo "$args" | sed 's/.*--name //')
  python3 tools/summarize_extra.py r06 $tag $name >> gpurun_out/collect_r06.log 2>&1
  rm -rf gpurun_out/prof_extra/$tag      # (the raw counter CSVs of one configuration are up to 60 MB: only the summaries travel back)
done
du -sh gpurun_out
