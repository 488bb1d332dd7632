"""Python-file configs -> attribute dicts, as det3d/torchie/utils/config.py:12-100 does (addict is not available
offline, so ConfigDict is a small dict subclass with attribute access and the same missing-key behaviour: AttributeError)."""
import importlib
import os
import sys


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name)

    def __setattr__(self, name, value):
        self[name] = value

    @staticmethod
    def wrap(v):
        if isinstance(v, dict):
            return ConfigDict({k: ConfigDict.wrap(x) for k, x in v.items()})
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict.wrap(x) for x in v)
        return v


class Config(object):
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg_dict", ConfigDict.wrap(cfg_dict or {}))
        object.__setattr__(self, "_filename", filename)

    @staticmethod
    def fromfile(filename):
        """imports the config file as a module with its directory on sys.path (sibling imports such as
        `from hrnet_cfg import hrnet_w18` work, config.py:85-88) and keeps its public names"""
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        if not filename.endswith(".py"):
            raise IOError("Only py type are supported now!")
        d, name = os.path.dirname(filename), os.path.basename(filename)[:-3]
        if "." in name:
            raise ValueError("Dots are not allowed in config file path.")
        sys.path.insert(0, d)
        try:
            mod = importlib.import_module(name)
            cfg = {k: v for k, v in mod.__dict__.items() if not k.startswith("__") and not isinstance(v, type(sys))}
        finally:
            sys.path.pop(0)
            sys.modules.pop(name, None)
        return Config(cfg, filename=filename)

    filename = property(lambda self: self._filename)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)
