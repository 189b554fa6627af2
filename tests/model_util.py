"""Build product models (halva_amd) from golden fixtures - test helper."""
import numpy as np
import torch

from golden_util import meta_of, tensors


def build_product_models(z, device="cuda", lora=True):
    cfg_d, clip_d = meta_of(z, "llama_cfg"), meta_of(z, "clip_cfg")
    base, clipW, fac = tensors(z, "base."), tensors(z, "clip."), tensors(z, "lora.")
    r, alpha = int(z["lora_cfg"][0]), float(z["lora_cfg"][1])
    return build_product_models_from(cfg_d, clip_d, base, clipW, fac, r, alpha, int(z["max_len"]), device, lora)


def build_product_models_from(cfg_d, clip_d, base, clipW, fac, r, alpha, max_len, device="cuda", lora=True):
    """(policy, reference, (r, alpha, factors)) from weight dicts in the fixtures' naming (HF names, fp32 tensors holding bf16 values)."""
    from halva_amd.clip import CLIPVisionConfig, CLIPVisionTower, build_vision_projector
    from halva_amd.llama import add_lora, load_hf_llama_weights
    from halva_amd.llava_model import LlavaConfig, LlavaLlamaForCausalLM

    def make(with_lora):
        cfg = LlavaConfig(**cfg_d)
        cfg.mm_projector_type, cfg.mm_hidden_size = "mlp2x_gelu", clip_d["hidden_size"]
        cfg.tokenizer_model_max_length, cfg.tokenizer_padding_side = max_len, "right"
        m = LlavaLlamaForCausalLM(cfg, dtype=torch.bfloat16, device=device)
        load_hf_llama_weights(m, base)
        vt = CLIPVisionTower("fixture", args=None, delay_load=True, config=CLIPVisionConfig(**clip_d), device=device)
        vt._alloc()
        vt.load_hf_state_dict(clipW)
        vt.requires_grad_(False)
        vt.is_loaded = True
        m.model.vision_tower = vt
        m.model.mm_projector = build_vision_projector(cfg, device=device)
        m.model.mm_projector.load_state_dict({k.split("mm_projector.")[1]: v for k, v in base.items() if "mm_projector." in k})
        for p in m.parameters():
            p.requires_grad_(False)
        if with_lora and fac:
            add_lora(m, r, alpha)
            with torch.no_grad():
                for i, layer in enumerate(m.model.layers):
                    for sub, grp in layer.groups():
                        for g, n in enumerate(grp.names):
                            key = "model.layers.%d.%s.%s" % (i, sub, n)
                            grp.A_cat[g * r:(g + 1) * r].copy_(fac[key + ".A"])
                            getattr(grp, n).lora_B["default"].weight.copy_(fac[key + ".B"])
            for p in m.model.mm_projector.parameters():
                p.requires_grad_(True)
        else:
            m._use_lora = False
        return m

    return make(lora), make(False), (r, alpha, fac)


def batch_of(z):
    return {k[len("batch."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch.")}


def build_product_vila(z, device="cuda", lora=True, max_len=None, padding_side="right"):
    """(policy, reference, (r, alpha, factors)) VILA wrappers from a vila_* fixture (llm.* / vis.* / proj.* tensors)."""
    from halva_amd.llama import LlamaConfig, add_lora, load_hf_llama_weights
    from halva_amd.siglip import SiglipVisionConfig, SiglipVisionTower
    from halva_amd.vila_model import LlamaForCausalLM, MultimodalProjector, VilaConfig, VilaLlavaLlamaModel
    cfg_d, vis_d = meta_of(z, "llama_cfg"), meta_of(z, "vis_cfg")
    llmW, visW, projW, fac = tensors(z, "llm."), tensors(z, "vis."), tensors(z, "proj."), tensors(z, "lora.")
    r, alpha = (int(z["lora_cfg"][0]), float(z["lora_cfg"][1])) if "lora_cfg" in z.files else (0, 0.0)
    if max_len is None:
        max_len = int(z["max_len"])

    def make(with_lora):
        cfg = VilaConfig(mm_hidden_size=vis_d["hidden_size"], hidden_size=cfg_d["hidden_size"])
        llm = LlamaForCausalLM(LlamaConfig(**cfg_d), torch.bfloat16, device)
        load_hf_llama_weights(llm, llmW)
        llm.config.tokenizer_model_max_length, llm.config.tokenizer_padding_side = max_len, padding_side
        vt = SiglipVisionTower("fixture", args=cfg, delay_load=True, config=SiglipVisionConfig(**vis_d), device=device)
        vt._alloc()
        vt.load_hf_state_dict(visW)
        vt.requires_grad_(False)
        vt.is_loaded = True
        proj = MultimodalProjector("mlp_downsample", cfg, device=device)
        proj.load_state_dict(projW)
        m = VilaLlavaLlamaModel(cfg, llm=llm, vision_tower=vt, mm_projector=proj, device=device)
        for p in m.parameters():
            p.requires_grad_(False)
        if with_lora and fac:
            add_lora(m.llm, r, alpha)
            with torch.no_grad():
                for i, layer in enumerate(m.llm.model.layers):
                    for sub, grp in layer.groups():
                        for g, n in enumerate(grp.names):
                            key = "model.layers.%d.%s.%s" % (i, sub, n)
                            grp.A_cat[g * r:(g + 1) * r].copy_(fac[key + ".A"])
                            getattr(grp, n).lora_B["default"].weight.copy_(fac[key + ".B"])
            for p in m.mm_projector.parameters():
                p.requires_grad_(True)
        else:
            m._use_lora = False
        return m

    return make(lora), make(False), (r, alpha, fac)
