#!/bin/bash
# HALVA on VILA1.5-13B (SigLIP-so400m-384 + mlp_downsample), LoRA DPA on one MI355X node.  Same flag set as the reference
# recipe (src_vila/halva_vila_13b.sh of the reference runs unchanged once `bin/` is on PATH: `deepspeed` resolves to the
# launcher shim in bin/deepspeed, one process per GPU over RCCL).
set -euo pipefail
cd "$(dirname "$0")/.."
export PATH="$PWD/bin:$PATH" HSA_ENABLE_IPC_MODE_LEGACY=0
python -c 'import __graft_entry__ as g; g.build()'

MODEL=${MODEL:-/models/VILA1.5-13b}                   # local VILA checkpoint directory (llm/ vision_tower/ mm_projector/)
OUT=${OUT:-./outputs/halva-13b-384-lora}

deepspeed train_halva_vila.py \
    --lora_enable True --lora_r 128 --lora_alpha 256 --mm_projector_lr 0 \
    --deepspeed src/json/zero3.json --loss_alpha 0.2 \
    --model_name_or_path "$MODEL" --version v1 \
    --data_path data/data.json --ref_data_path data/ref_data.json --image_folder "${IMG_DIR:-default}" \
    --vision_tower google/siglip-so400m-patch14-384 --mm_vision_select_feature cls_patch --mm_projector mlp_downsample \
    --tune_vision_tower False --tune_mm_projector True --tune_language_model False --mm_vision_select_layer -2 \
    --mm_use_im_start_end False --mm_use_im_patch_token False --image_aspect_ratio resize \
    --bf16 True --output_dir "$OUT" \
    --num_train_epochs 1 --per_device_train_batch_size 4 --per_device_eval_batch_size 4 \
    --gradient_accumulation_steps 4 --evaluation_strategy "no" --save_strategy "steps" --save_steps 50000 \
    --learning_rate 2.5e-5 --weight_decay 0. --warmup_ratio 0.03 --lr_scheduler_type "cosine" --logging_steps 1 \
    --tf32 True --model_max_length 4096 --gradient_checkpointing True --dataloader_num_workers 8 \
    --lazy_preprocess True --report_to "wandb" --save_total_limit 1 --vflan_no_system_prompt True \
    --run_name halva-13b-384-lora
