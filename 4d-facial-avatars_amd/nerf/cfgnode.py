"""Attribute-dict configuration node with the surface the NeRFace scripts use.

Mirrors the reference's `CfgNode` (nerf/cfgnode.py:36, YACS-style) for what train_transformed_rays.py /
eval_transformed_rays.py touch: construction from a (nested) dict loaded with yaml, attribute access at
any depth, `dump()` back to YAML (cfgnode.py:167), plus freeze/clone/merge helpers.  Pure host code.
"""
from __future__ import annotations

import copy
from typing import Any, Iterable, Optional

import yaml

_VALID_TYPES = (tuple, list, str, int, float, bool, type(None))


class CfgNode(dict):
    IMMUTABLE = "__immutable__"
    NEW_ALLOWED = "__new_allowed__"

    def __init__(self, init_dict: Optional[dict] = None, key_list: Optional[list] = None, new_allowed: bool = False):
        init_dict = {} if init_dict is None else init_dict
        key_list = [] if key_list is None else key_list
        super().__init__(self._convert(init_dict, key_list))
        self.__dict__[CfgNode.IMMUTABLE] = False
        self.__dict__[CfgNode.NEW_ALLOWED] = new_allowed

    @classmethod
    def _convert(cls, d: dict, key_list: list) -> dict:
        out = {}
        for k, v in d.items():
            if isinstance(v, dict) and not isinstance(v, CfgNode):
                out[k] = cls(v, key_list=key_list + [k])
            else:
                if not isinstance(v, _VALID_TYPES + (dict,)):
                    raise TypeError(f"Key {'.'.join(map(str, key_list + [k]))} with value {type(v)} is not a valid type")
                out[k] = copy.deepcopy(v)
        return out

    # attribute access -----------------------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        if self.is_frozen():
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        if name in self.__dict__:
            raise AttributeError(f"Invalid attempt to modify internal CfgNode state: {name}")
        self[name] = value

    def __str__(self) -> str:
        def indent(s, n):
            lines = s.split("\n")
            return lines[0] if len(lines) == 1 else "\n".join([lines[0]] + [" " * n + l for l in lines[1:]])
        parts = []
        for k, v in sorted(self.items()):
            sep = "\n" if isinstance(v, CfgNode) else " "
            parts.append(indent(f"{k}:{sep}{v}", 2))
        return "\n".join(parts)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({super().__repr__()})"

    # YAML -----------------------------------------------------------------------------------------
    def to_dict(self) -> dict:
        return {k: (v.to_dict() if isinstance(v, CfgNode) else copy.deepcopy(v)) for k, v in self.items()}

    def dump(self, **kwargs) -> str:
        return yaml.safe_dump(self.to_dict(), **kwargs)

    @classmethod
    def load_cfg(cls, cfg_file_obj_or_str) -> "CfgNode":
        return cls(yaml.safe_load(cfg_file_obj_or_str))

    def merge_from_file(self, cfg_filename: str) -> None:
        with open(cfg_filename, "r") as f:
            self.merge_from_other_cfg(self.load_cfg(f))

    def merge_from_other_cfg(self, other: "CfgNode") -> None:
        def merge(a, b, path):
            for k, v in a.items():
                if k in b and isinstance(b[k], CfgNode) and isinstance(v, CfgNode):
                    merge(v, b[k], path + [k])
                elif k in b or b.is_new_allowed():
                    b[k] = copy.deepcopy(v)
                else:
                    raise KeyError(f"Non-existent config key: {'.'.join(path + [k])}")
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable")
        merge(other, self, [])

    def merge_from_list(self, cfg_list: Iterable) -> None:
        cfg_list = list(cfg_list)
        if len(cfg_list) % 2:
            raise ValueError("Override list has odd length")
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            keys = full_key.split(".")
            for sub in keys[:-1]:
                d = d[sub]
            if isinstance(v, str):
                try:
                    v = yaml.safe_load(v)
                except yaml.YAMLError:
                    pass
            d[keys[-1]] = v

    # mutability -----------------------------------------------------------------------------------
    def freeze(self) -> None:
        self._immutable(True)

    def defrost(self) -> None:
        self._immutable(False)

    def is_frozen(self) -> bool:
        return self.__dict__[CfgNode.IMMUTABLE]

    def _immutable(self, flag: bool) -> None:
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._immutable(flag)

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def is_new_allowed(self) -> bool:
        return self.__dict__[CfgNode.NEW_ALLOWED]
