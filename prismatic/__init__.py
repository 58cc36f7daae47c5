"""Import root of the reference (`prismatic.*`) served by the MI355X implementation: the reference's callers keep their
import lines (`from prismatic.vla.action_tokenizer import ActionTokenizer`, `from prismatic.extern.hf.modeling_prismatic
import OpenVLAForActionPrediction`, … — vla-scripts/finetune.py:36-46, deploy.py:24-30, experiments/robot/openvla_utils.py)
and get bridgelang_amd's classes. Only modules on the OpenVLA hot path exist; importing anything else raises
ModuleNotFoundError, as an uninstalled submodule would. This package holds no code of its own: every name is an alias
registered in sys.modules below."""
import importlib
import sys
import types

_ALIASES = {
    "prismatic.extern": "bridgelang_amd.extern",
    "prismatic.extern.hf": "bridgelang_amd.extern.hf",
    "prismatic.extern.hf.configuration_prismatic": "bridgelang_amd.extern.hf.configuration_prismatic",
    "prismatic.extern.hf.modeling_prismatic": "bridgelang_amd.extern.hf.modeling_prismatic",
    "prismatic.extern.hf.processing_prismatic": "bridgelang_amd.extern.hf.processing_prismatic",
    "prismatic.vla": "bridgelang_amd.vla",
    "prismatic.vla.action_tokenizer": "bridgelang_amd.vla.action_tokenizer",
    "prismatic.vla.datasets": "bridgelang_amd.vla.datasets",
    "prismatic.util": "bridgelang_amd.util",
    "prismatic.util.data_utils": "bridgelang_amd.util.data_utils",
    "prismatic.conf": "bridgelang_amd.conf",
    "prismatic.conf.vla": "bridgelang_amd.conf.vla",
    "prismatic.models.load": "bridgelang_amd.models.load",
    "prismatic.models.materialize": "bridgelang_amd.models.materialize",
    "prismatic.models.vlms": "bridgelang_amd.models.vlms",
    "prismatic.models.vlas": "bridgelang_amd.models.vlms",
    "prismatic.models.backbones.llm.prompting": "bridgelang_amd.models.prompting",
    "prismatic.training.metrics": "bridgelang_amd.training.strategy",
    "prismatic.training.materialize": "bridgelang_amd.training.strategy",
}


def _namespace(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = []                       # a package: submodules resolve through sys.modules
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


for _alias, _target in _ALIASES.items():
    sys.modules[_alias] = importlib.import_module(_target)

# packages of the reference that are plain containers here (their public names re-exported as the reference does)
from bridgelang_amd.models.load import load_vla  # noqa: E402
from bridgelang_amd.models.materialize import (  # noqa: E402
    get_llm_backbone_and_tokenizer, get_vision_backbone_and_transform, get_vlm)
from bridgelang_amd.training.strategy import VLAMetrics, get_train_strategy  # noqa: E402

_namespace("prismatic.models", load_vla=load_vla, get_llm_backbone_and_tokenizer=get_llm_backbone_and_tokenizer,
           get_vision_backbone_and_transform=get_vision_backbone_and_transform, get_vlm=get_vlm,
           load=sys.modules["prismatic.models.load"], materialize=sys.modules["prismatic.models.materialize"])
_namespace("prismatic.models.backbones")
_namespace("prismatic.models.backbones.llm", prompting=sys.modules["prismatic.models.backbones.llm.prompting"])
_namespace("prismatic.training", VLAMetrics=VLAMetrics, get_train_strategy=get_train_strategy,
           metrics=sys.modules["prismatic.training.metrics"], materialize=sys.modules["prismatic.training.materialize"])
for _name in ("extern", "vla", "util", "conf", "models", "training"):
    globals()[_name] = sys.modules[f"prismatic.{_name}"]
