"""The VILA twin's HallDataset + collator on CPU: [n,3,H,W] images through the real SigLIP image processor with
aspect ratio 'resize', a single-channel image through patched_normalize, and the reference quirk that the reference
sample's image comes from the training sample of the same index (vila/train/train_halva.py:1111)."""
import types

import torch

import e2e_util


def test_vila_dataset_and_collator(tmp_path, monkeypatch):
    from unittest import mock
    from transformers import SiglipImageProcessor
    import os
    from llava import conversation as conv_lib
    import vila.train.train_halva as TV
    from vila.train.transformer_normalize_monkey_patch import patched_normalize
    paths = e2e_util.build_vila(str(tmp_path))
    tok = e2e_util._Tok(model_max_length=64)
    tok.pad_token = tok.unk_token
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    proc = SiglipImageProcessor.from_pretrained(os.path.join(paths["ckpt"], "vision_tower"))
    args = types.SimpleNamespace(data_path=paths["data"], ref_data_path=paths["ref"], image_folder=paths["images"],
                                 image_aspect_ratio="resize", image_processor=proc, is_multimodal=True, mm_use_im_start_end=False)
    with mock.patch("transformers.image_transforms.normalize", new=patched_normalize):
        mod = TV.make_supervised_data_module(tok, args)
        ds, coll = mod["train_dataset"], mod["data_collator"]
        items = [ds[i] for i in range(len(ds))]
    S = paths["image_size"]
    for it in items:
        assert it["image"].shape == (1, 3, S, S) and it["ref_image"].shape == (1, 3, S, S)
        assert it["input_ids"].shape == it["labels"].shape == it["pos_signs"].shape
        assert int((it["input_ids"] == -200).sum()) == 1
        assert float(it["image"].abs().max()) <= 1.0 + 1e-6            # rescale + (x - .5) / .5
    # reference quirk: ref image == training image of the same index although the ref sample names another file
    assert any(torch.equal(it["image"], it["ref_image"]) for it in items)
    batch = coll(items[:3])
    assert batch["images"].shape == (3, 1, 3, S, S) and batch["ref_images"].shape == (3, 1, 3, S, S)
    assert batch["input_ids"].shape == batch["pos_signs"].shape
    assert len(ds.lengths) == len(ds) and all(l > 128 for l in ds.lengths)
