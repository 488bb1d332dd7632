"""Plugin registry + cfg builder with the reference's names, behaviour and error messages
(det3d/utils/registry.py:6-78, det3d/models/registry.py:3-15): existing config dicts build unchanged."""
import inspect


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return "%s(name=%s, items=%s)" % (type(self).__name__, self._name, list(self._module_dict))

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._module_dict)

    def get(self, key):
        return self._module_dict.get(key)

    def _register_module(self, module_class):
        if not inspect.isclass(module_class):
            raise TypeError("module must be a class, but got {}".format(type(module_class)))
        key = module_class.__name__
        if key in self._module_dict:
            raise KeyError("{} is already registered in {}".format(key, self.name))
        self._module_dict[key] = module_class

    def register_module(self, cls):
        self._register_module(cls)
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    """cfg['type'] (str looked up in `registry`, or a class) is instantiated with the remaining keys;
    default_args fill in missing keys only."""
    assert isinstance(cfg, dict) and "type" in cfg
    assert isinstance(default_args, dict) or default_args is None
    kwargs = dict(cfg)
    kind = kwargs.pop("type")
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError("{} is not in the {} registry".format(kind, registry.name))
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError("type must be a str or valid type, but got {}".format(type(kind)))
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return cls(**kwargs)


READERS = Registry("reader")
BACKBONES = Registry("backbone")
IMG_BACKBONES = Registry("img_backbone")
IMG_HEADS = Registry("img_head")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
SECOND_STAGE = Registry("second_stage")
ROI_HEAD = Registry("roi_head")
POINT_HEADS = Registry("point_head")
