"""A/B hook for the measurement tools (tools/ab_env.sh, profiles/round*_experiments.md) - NOT part of the library's interface.

The library reads FOUR environment switches a user may want: LS3D_PRECISION, LS3D_REFERENCE_OUTPUTS, LS3D_CAPACITY_MODE,
LS3D_OVERLAP.  Every other scheduling / kernel-selection knob is a module-level constant with a setter (ops.set_tile_chain, ops.set_tile,
point_heads.set_fused_sffm_memory, ...) that the tests toggle directly (tests/test_gpu_parity.py::
test_every_schedule_switch_of_the_host_layer_keeps_the_logits).  To A/B such a constant on an UNMODIFIED script (bench.py under rocprofv3),

    LS3D_EXPERIMENT="ops._TILE_CHAIN=0,scn_unet._LATERAL=0,ops._CHAIN_MIN_TILES=1" python bench.py ...

sets module attributes by name once, at import time; values are parsed as int, then float, then kept as strings ("subm,conv")."""
import importlib
import os


def _value(text):
    for conv in (int, float):
        try:
            return conv(text)
        except ValueError:
            pass
    return text


def apply(spec=None):
    """-> list of (module, attribute, old, new) for every assignment of the spec (default: $LS3D_EXPERIMENT)"""
    spec = os.environ.get("LS3D_EXPERIMENT", "") if spec is None else spec
    done = []
    for item in filter(None, (s.strip() for s in spec.split(";" if ";" in spec else ","))):
        target, _, text = item.partition("=")
        mod_name, _, attr = target.strip().rpartition(".")
        mod = importlib.import_module("lidarseg3d_amd." + mod_name)
        if not hasattr(mod, attr):
            raise AttributeError("LS3D_EXPERIMENT: lidarseg3d_amd.%s has no constant %r" % (mod_name, attr))
        old = getattr(mod, attr)
        new = _value(text.strip())
        if isinstance(old, bool):
            new = bool(new)
        setattr(mod, attr, new)
        done.append((mod_name, attr, old, new))
    return done
