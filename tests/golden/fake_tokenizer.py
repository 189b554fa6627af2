"""Deterministic stand-in for the slow (sentencepiece, legacy) Llama tokenizer.

Test infrastructure only.  No tokenizer.model exists offline (SURVEY.md ground facts), so the
golden vectors for the masked-tokenisation path (reference llava/train/train_halva.py:263-479) are
generated and replayed with this class.  It reproduces the Llama-legacy behaviours the
reference's span-wise tokenisation walk depends on:

  * BOS (id 1) is prepended to every call,
  * the text is prefixed with the word-boundary marker and ' ' -> marker, so a leading space
    becomes a separate marker piece and a trailing space a trailing marker piece,
  * the literal "</s>" is one special piece (id 2), pad == unk == 0.

The vocabulary grows in first-seen order and is frozen into the fixture files, so replaying a
fixture never depends on dict ordering or hashing.
"""
import re
from types import SimpleNamespace

_MARK = "▁"
_PIECE = re.compile("</s>|" + _MARK + "?[A-Za-z0-9]+|" + _MARK + "|\n|.", re.S)


class FakeLlamaTokenizer:
    bos_token_id = 1
    eos_token_id = 2
    pad_token_id = 0
    unk_token_id = 0
    padding_side = "right"

    def __init__(self, model_max_length=2048, vocab=None, frozen=False):
        self.model_max_length = model_max_length
        self.vocab = dict(vocab) if vocab else {}
        self.frozen = frozen

    def _id(self, piece):
        if piece == "</s>":
            return self.eos_token_id
        if piece not in self.vocab:
            if self.frozen:
                raise KeyError("piece %r not in frozen fixture vocab" % piece)
            self.vocab[piece] = 3 + len(self.vocab)
        return self.vocab[piece]

    def pieces(self, text):
        return _PIECE.findall(_MARK + text.replace(" ", _MARK))

    def __call__(self, text):
        ids = [self.bos_token_id] + [self._id(p) for p in self.pieces(text)]
        return SimpleNamespace(input_ids=ids)

    def __len__(self):
        return 3 + len(self.vocab)
