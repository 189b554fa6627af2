#!/bin/bash
# HALVA-7B LoRA DPA on one MI355X node.  Same flag set as the reference recipe (its src/hallava_7b.sh also runs
# unchanged once `bin/` is on PATH: the `deepspeed` word resolves to the launcher shim in bin/deepspeed).
set -euo pipefail
cd "$(dirname "$0")/.."
export PATH="$PWD/bin:$PATH" HSA_ENABLE_IPC_MODE_LEGACY=0
python -c 'import __graft_entry__ as g; g.build()'

MODEL=${MODEL:-/models/llava-v1.5-7b}                 # local HF checkpoint directory (no hub access)
VISION=${VISION:-/models/clip-vit-large-patch14-336}
OUT=${OUT:-./outputs/halva-7b-lora}

deepspeed train_halva.py \
    --lora_enable True --lora_r 128 --lora_alpha 256 --mm_projector_lr 0 \
    --deepspeed src/json/zero3.json --loss_alpha 0.4 \
    --model_name_or_path "$MODEL" --version v1 \
    --data_path data/data.json --ref_data_path data/ref_data.json --image_folder "${IMG_DIR:-default}" \
    --vision_tower "$VISION" --mm_projector_type mlp2x_gelu --mm_vision_select_layer -2 \
    --mm_use_im_start_end False --mm_use_im_patch_token False --image_aspect_ratio pad \
    --group_by_modality_length True --bf16 True --output_dir "$OUT" \
    --num_train_epochs 1 --per_device_train_batch_size 4 --per_device_eval_batch_size 4 \
    --gradient_accumulation_steps 4 --evaluation_strategy "no" --save_strategy "steps" --save_steps 50000 \
    --learning_rate 5e-6 --weight_decay 0. --warmup_ratio 0.03 --lr_scheduler_type "cosine" --logging_steps 1 \
    --tf32 True --model_max_length 2048 --gradient_checkpointing True --dataloader_num_workers 8 \
    --lazy_preprocess True --report_to "wandb" --save_total_limit 1 --run_name halva-7b-lora
