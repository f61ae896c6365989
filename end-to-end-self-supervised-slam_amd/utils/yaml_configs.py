"""reference: utils/yaml_configs.py:16-38 (YAML <-> attribute dictionary; easydict is not required)."""
import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return AttrDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def load_yaml(path):
    with open(path) as f:
        return _wrap(yaml.safe_load(f))


def save_yaml(path, cfg):
    with open(path, "w") as f:
        yaml.safe_dump({k: (dict(v) if isinstance(v, dict) else v) for k, v in dict(cfg).items()}, f)
