"""Conversation templates needed by the DPA path: only the `v1` (vicuna_v1, SeparatorStyle.TWO) template the
reference asserts on (llava/train/train_halva.py:1188; template at reference llava/conversation.py:252-262).
The gradio/serving parts of the reference's file are out of scope."""
import dataclasses
from enum import Enum, auto
from typing import List


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()


@dataclasses.dataclass
class Conversation:
    system: str
    roles: List[str]
    messages: List[List[str]]
    offset: int
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: str = None
    version: str = "Unknown"
    skip_next: bool = False

    def get_prompt(self):
        if self.sep_style != SeparatorStyle.TWO:
            raise NotImplementedError("only SeparatorStyle.TWO (v1) is on the HALVA path")
        seps = (self.sep, self.sep2)
        out = self.system + seps[0]
        for i, (role, message) in enumerate(self.messages):
            if message:
                if type(message) is tuple:
                    message = message[0]
                out += role + ": " + message + seps[i % 2]
            else:
                out += role + ":"
        return out

    def append_message(self, role, message):
        self.messages.append([role, message])

    def copy(self):
        return Conversation(system=self.system, roles=self.roles, messages=[[x, y] for x, y in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2, version=self.version)


conv_vicuna_v1 = Conversation(
    system="A chat between a curious user and an artificial intelligence assistant. "
           "The assistant gives helpful, detailed, and polite answers to the user's questions.",
    roles=("USER", "ASSISTANT"), version="v1", messages=(), offset=0, sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")

default_conversation = conv_vicuna_v1
conv_templates = {"default": conv_vicuna_v1, "v1": conv_vicuna_v1, "vicuna_v1": conv_vicuna_v1}
