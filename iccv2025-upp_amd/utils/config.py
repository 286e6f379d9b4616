"""YAML -> attribute dict, with the reference's `_base_` include (reference utils/config.py:18-58)."""
import os

import yaml


class EasyDict(dict):
    """dict with attribute access, recursively applied (stand-in for the `easydict` package)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def merge_new_config(config, new_config):
    for key, val in new_config.items():
        if isinstance(val, dict):
            if key == '_base_':
                with open(new_config['_base_'], 'r') as f:
                    val = yaml.safe_load(f)
                config[key] = merge_new_config(EasyDict(), val)
            else:
                config[key] = merge_new_config(EasyDict(), val)
        elif key == '_base_':
            with open(val, 'r') as f:
                config[key] = merge_new_config(EasyDict(), yaml.safe_load(f))
        else:
            config[key] = val
    return config


def cfg_from_yaml_file(cfg_file):
    with open(cfg_file, 'r') as f:
        return merge_new_config(EasyDict(), yaml.safe_load(f))


def builtin_cfg(name):
    """Configs shipped with this package (same field names as the reference's cfgs/*.yaml)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, "cfgs")
    return cfg_from_yaml_file(os.path.join(here, name if name.endswith(".yaml") else name + ".yaml"))
