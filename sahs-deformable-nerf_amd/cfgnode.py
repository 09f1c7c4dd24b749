"""Attribute-style config node.

The hot path reads its options by attribute from a YACS-like node built over a YAML dict
(reference: ``nerf/cfgnode.py:36-66``; read sites ``train_utils.py:96-106,126-127,148-149,154,
239,243,255-256,303`` and ``models.py:192-296``).  This is a small independent implementation
of the part of that interface the path uses: nested dicts become nodes, keys are attributes,
``hasattr(cfg.models, "fine")`` works, and the reference's YAML files load unchanged.
"""
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init_dict=None):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = CfgNode(value) if isinstance(value, dict) and not isinstance(value, CfgNode) else value

    def clone(self):
        return copy.deepcopy(self)

    @classmethod
    def load_yaml(cls, path):
        with open(path, "r") as f:
            return cls(yaml.safe_load(f))


def default_config(kind="audio"):
    """The hot-path subset of ``config/audio/person_2_auto.yml`` (kind="audio") or of ``config/expression/person_2.yml``
    (kind="expression": NeRFaceModel with warp + hyper sheet) or of ``config/expression/person_1.yml`` (kind="expression_static":
    both off) -- same keys, same values."""
    import os
    cfg = CfgNode.load_yaml(os.path.join(os.path.dirname(__file__), "config", {"audio": "audio_hotpath.yml", "expression": "expression_hotpath.yml",
                                                                               "expression_static": "expression_hotpath.yml"}[kind]))
    if kind == "expression_static":   # person_1.yml differs from person_2.yml in exactly these keys (the hot path reads)
        cfg.models.warp.use_warp, cfg.models.hyper.use_ambient = False, False
        for node in (cfg.models.warp, cfg.models.hyper, cfg.models.coarse, cfg.models.fine):
            node.num_encoding_fn_xyz = 10
        cfg.models.hyper.num_encoding_fn_ambient = 4
    return cfg
