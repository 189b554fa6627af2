"""Sentinel ids and placeholder strings of the LLaVA data format.  The values are fixed by the released checkpoints and by
data/data.json (reference llava/constants.py:6-14; the <MASK> tags: llava/train/train_halva.py:36-37)."""
# label / input-id sentinels: never valid vocabulary ids
IGNORE_INDEX, IMAGE_TOKEN_INDEX = -100, -200

# textual placeholders that may appear in a conversation turn
_TAGS = dict(DEFAULT_IMAGE_TOKEN="image", DEFAULT_IMAGE_PATCH_TOKEN="im_patch", DEFAULT_IM_START_TOKEN="im_start",
             DEFAULT_IM_END_TOKEN="im_end", IMAGE_PLACEHOLDER="image-placeholder", DESCRIPTION_SEPARATOR="description-seperator")
globals().update({name: "<%s>" % tag for name, tag in _TAGS.items()})
__all__ = ["IGNORE_INDEX", "IMAGE_TOKEN_INDEX", *_TAGS]
