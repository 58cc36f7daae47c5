"""Command-line parsing in the style the reference's entry scripts get from draccus (`@draccus.wrap()`,
vla-scripts/train.py:106, finetune.py:113): every dataclass field is a flag `--field value` (or `--field=value`), nested
dataclass fields are `--parent.field value`, and a registry-typed field picks its subclass with `--parent.type <id>`
(`--vla.type prism-dinosiglip-224px+mx-bridge`). Values are converted by the field's annotation (bool accepts
True/False/true/false/1/0; Optional[...] accepts None/null; Path, int, float, str, Tuple[str, ...] from a comma list).
`dump_yaml` / `encode` write the `config.yaml` + `config.json` pair train.py:134-138 leaves in the run directory."""
from __future__ import annotations

import dataclasses
import json
import sys
import typing
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Type


def _convert(text: str, tp: Any) -> Any:
    origin = typing.get_origin(tp)
    if origin is typing.Union:
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        if text in ("None", "none", "null", "~") and len(args) < len(typing.get_args(tp)):
            return None
        for a in (x for x in args if x is not str):             # try the specific members before `str`
            try:
                return _convert(text, a)
            except (ValueError, TypeError):
                continue
        return text
    if origin in (tuple, typing.Tuple):
        return tuple(t for t in text.strip("()[] ").replace(" ", "").split(",") if t)
    if tp is bool:
        if text.lower() in ("true", "1", "yes"):
            return True
        if text.lower() in ("false", "0", "no"):
            return False
        raise ValueError(f"not a boolean: {text!r}")
    if tp in (int, float, str):
        return tp(text)
    if tp is Path:
        return Path(text)
    return text


def _hints(cls: Type) -> Dict[str, Any]:
    return typing.get_type_hints(cls)


def parse(cls: Type, argv: Optional[Sequence[str]] = None):
    """Build `cls` (a dataclass) from `--flag value` arguments."""
    argv = list(sys.argv[1:] if argv is None else argv)
    pairs: List[tuple] = []
    i = 0
    while i < len(argv):
        a = argv[i]
        if not a.startswith("--"):
            raise SystemExit(f"unexpected argument {a!r} (flags look like --field value)")
        if "=" in a:
            k, v = a[2:].split("=", 1)
            v = v.lstrip("=")                                  # the README writes `--image_aug==False` in prose
            i += 1
        else:
            if i + 1 >= len(argv):
                raise SystemExit(f"flag {a} needs a value")
            k, v = a[2:], argv[i + 1]
            i += 2
        pairs.append((k, v))
    hints = _hints(cls)
    top: Dict[str, Any] = {}
    nested: Dict[str, Dict[str, str]] = {}
    for k, v in pairs:
        if "." in k:
            parent, child = k.split(".", 1)
            nested.setdefault(parent, {})[child] = v
        else:
            if k not in hints:
                raise SystemExit(f"unknown flag --{k} for {cls.__name__}; fields: {sorted(hints)}")
            top[k] = _convert(v, hints[k])
    for parent, kv in nested.items():
        if parent not in hints:
            raise SystemExit(f"unknown flag --{parent}.* for {cls.__name__}")
        base = hints[parent]
        sub = base
        if "type" in kv:
            sub = base.get_choice_class(kv.pop("type"))
        elif hasattr(base, "get_choice_class"):
            default = next(f for f in dataclasses.fields(cls) if f.name == parent)
            sub = type(default.default_factory()) if default.default_factory is not dataclasses.MISSING else base
        sh = _hints(sub)
        for c in kv:
            if c not in sh:
                raise SystemExit(f"unknown flag --{parent}.{c}; fields: {sorted(sh)}")
        top[parent] = sub(**{c: _convert(v, sh[c]) for c, v in kv.items()})
    return cls(**top)


def encode(cfg: Any) -> Any:
    """dataclass tree → plain dict (Paths as strings, tuples as lists, registry members carry their `type`)."""
    if dataclasses.is_dataclass(cfg):
        out = {f.name: encode(getattr(cfg, f.name)) for f in dataclasses.fields(cfg) if not f.name.startswith("_")}
        if hasattr(type(cfg), "get_choice_class") and hasattr(cfg, "vla_id"):
            out = {"type": cfg.vla_id, **out}
        return out
    if isinstance(cfg, Path):
        return str(cfg)
    if isinstance(cfg, (list, tuple)):
        return [encode(x) for x in cfg]
    if isinstance(cfg, dict):
        return {k: encode(v) for k, v in cfg.items()}
    return cfg


def dump_yaml(cfg: Any, path: Path) -> None:
    import yaml
    Path(path).write_text(yaml.safe_dump(encode(cfg), sort_keys=False))


def dump_yaml_and_json(cfg: Any, run_dir: Path) -> None:
    """train.py:134-138: config.yaml, then the same tree as config.json (what load_vla reads back)."""
    import yaml
    dump_yaml(cfg, Path(run_dir) / "config.yaml")
    tree = yaml.safe_load((Path(run_dir) / "config.yaml").read_text())
    (Path(run_dir) / "config.json").write_text(json.dumps(tree, indent=2))
