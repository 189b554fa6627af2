"""CPU end-to-end of the host pipeline: HallDataset (JSON -> pos/neg/ref tensors incl. image preprocessing) + collator."""
import types

import pytest
import torch

import e2e_util


def test_hall_dataset_and_collator(tmp_path, monkeypatch):
    import llava.train.train_halva as TH
    paths = e2e_util.build(str(tmp_path))
    from transformers import CLIPImageProcessor
    tok = e2e_util._Tok(model_max_length=64)
    tok.pad_token = tok.unk_token
    args = TH.DataArguments(data_path=paths["data"], ref_data_path=paths["ref"], image_folder=paths["images"], image_aspect_ratio="pad")
    args.image_processor = CLIPImageProcessor.from_pretrained(paths["vision"])
    args.is_multimodal = True
    args.mm_use_im_start_end = False
    mod = TH.make_supervised_data_module(tok, args)
    ds, coll = mod["train_dataset"], mod["data_collator"]
    assert len(ds) == 6 and len(ds.modality_lengths) == 6 and all(l > 0 for l in ds.modality_lengths)
    items = [ds[i] for i in range(4)]
    for it in items:
        assert it["input_ids"].shape == it["labels"].shape == it["pos_signs"].shape
        assert it["neg_input_ids"].shape == it["neg_signs"].shape
        assert (it["input_ids"] == -200).sum() == 1 and int(it["pos_signs"].max()) >= 1
        assert it["image"].shape == (3, 28, 28) and it["ref_image"].shape == (3, 28, 28)
        assert (it["labels"] != -100).sum() > 0
        # pos and neg differ only inside phrase spans (same lengths here)
        same = it["input_ids"] == it["neg_input_ids"]
        assert bool(((~same) <= (it["pos_signs"] > 0)).all())
    batch = coll(items)
    assert set(batch) == {"input_ids", "labels", "attention_mask", "neg_input_ids", "neg_labels", "neg_attention_mask", "pos_signs",
                          "neg_signs", "ref_input_ids", "ref_labels", "ref_attention_mask", "images", "ref_images"}
    assert batch["images"].shape == (4, 3, 28, 28) and batch["attention_mask"].dtype == torch.bool
    assert len(tok) < paths["vocab_size"], "fixture text must fit the tiny model's vocabulary"
