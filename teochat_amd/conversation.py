"""Prompt templates for the path.

Mirrors the slice of the reference's videollava/conversation.py that run_inference_single uses:
`conv_templates["v1"]` (Vicuna v1, SeparatorStyle.TWO, conversation.py:51-60,252-262) with
copy()/append_message()/get_prompt().  SINGLE and PLAIN styles are kept because they are one-liners;
the UI helpers (get_images, to_gradio_chatbot) and the MPT/LLAMA_2 styles are out of scope.
"""
import copy as _copy
import enum
from dataclasses import dataclass, field
from typing import List, Optional, Sequence


class SeparatorStyle(enum.Enum):
    SINGLE = enum.auto()
    TWO = enum.auto()
    MPT = enum.auto()
    PLAIN = enum.auto()
    LLAMA_2 = enum.auto()


def _text(message):
    # a message may be (text, image, mode) in the UI; only the text is part of the prompt
    return message[0] if isinstance(message, tuple) else message


@dataclass
class Conversation:
    system: str
    roles: Sequence[str]
    messages: List[List[Optional[str]]] = field(default_factory=list)
    offset: int = 0
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def append_message(self, role, message):
        self.messages.append([role, message])

    def copy(self):
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2,
                            version=self.version)

    def get_prompt(self) -> str:
        style = self.sep_style
        if style == SeparatorStyle.TWO:
            ends = (self.sep, self.sep2)
            parts = [self.system, ends[0]]
            for turn, (role, message) in enumerate(self.messages):
                if message:
                    parts += [role, ": ", _text(message), ends[turn % 2]]
                else:
                    parts += [role, ":"]
            return "".join(parts)
        if style == SeparatorStyle.SINGLE:
            parts = [self.system, self.sep]
            for role, message in self.messages:
                parts += [role, ": ", _text(message), self.sep] if message else [role, ":"]
            return "".join(parts)
        if style == SeparatorStyle.PLAIN:
            ends = (self.sep, self.sep2)
            parts = [self.system]
            for turn, (_, message) in enumerate(self.messages):
                if message:
                    parts += [_text(message), ends[turn % 2]]
            return "".join(parts)
        raise ValueError(f"Invalid style: {style}")

    def dict(self):
        return {"system": self.system, "roles": self.roles, "messages": self.messages, "offset": self.offset,
                "sep": self.sep, "sep2": self.sep2}


conv_vicuna_v1 = Conversation(
    system="A chat between a curious user and an artificial intelligence assistant. "
           "The assistant gives helpful, detailed, and polite answers to the user's questions.",
    roles=("USER", "ASSISTANT"), version="v1", messages=[], offset=0,
    sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")

conv_llava_plain = Conversation(system="", roles=("", ""), messages=[], offset=0, sep_style=SeparatorStyle.PLAIN,
                                sep="\n")

default_conversation = conv_vicuna_v1
conv_templates = {"v1": conv_vicuna_v1, "vicuna_v1": conv_vicuna_v1, "llava_v1": conv_vicuna_v1,
                  "plain": conv_llava_plain, "v0_plain": conv_llava_plain}
