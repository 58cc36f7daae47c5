"""`PurePromptBuilder` — the OpenVLA prompt format (prismatic/models/backbones/llm/prompting/base_prompter.py:28-73):
human turns become "In: {msg}\\nOut: ", model turns "{msg}</s>"; `get_prompt()` strips a leading <s> and trailing
whitespace (the tokenizer re-inserts BOS). `vla_prompt()` is the fixed question OpenVLA asks (openvla.py:52-54)."""
from __future__ import annotations

from typing import Optional


class PurePromptBuilder:
    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        self.model_family, self.system_prompt = model_family, system_prompt
        self.bos, self.eos = "<s>", "</s>"
        self.prompt, self.turn_count = "", 0

    def _wrap(self, role_is_human: bool, msg: str) -> str:
        return f"In: {msg}\nOut: " if role_is_human else f"{msg if msg != '' else ' '}{self.eos}"

    def add_turn(self, role: str, message: str) -> str:
        human = self.turn_count % 2 == 0
        assert role == ("human" if human else "gpt")
        wrapped = self._wrap(human, message.replace("<image>", "").strip())
        self.prompt += wrapped
        self.turn_count += 1
        return wrapped

    def get_potential_prompt(self, message: str) -> str:
        return (self.prompt + self._wrap(True, message)).removeprefix(self.bos).rstrip()

    def get_prompt(self) -> str:
        return self.prompt.removeprefix(self.bos).rstrip()


def vla_prompt(instruction: str) -> str:
    b = PurePromptBuilder("openvla")
    b.add_turn("human", f"What action should the robot take to {instruction.lower()}?")
    return b.get_prompt()
