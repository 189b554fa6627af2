"""tokenizer_image_token of the reference (llava/mm_utils.py:43-62): tokenise the text around each `<image>` tag and
put IMAGE_TOKEN_INDEX between the pieces, keeping a single leading BOS."""
import torch

from llava.constants import IMAGE_TOKEN_INDEX


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    pieces = [tokenizer(part).input_ids for part in prompt.split("<image>")]
    lead_bos = bool(pieces) and bool(pieces[0]) and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if lead_bos else 0
    ids = [pieces[0][0]] if lead_bos else []
    for n, piece in enumerate(pieces):
        if n > 0:
            ids.append(image_token_index)
        ids.extend(piece[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")
