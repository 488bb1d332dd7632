"""Checkpoint loading compatible with the reference's files (det3d/torchie/trainer/checkpoint.py:42-173):
{meta, state_dict, optimizer} dicts saved by torch.save, `module.` prefixes stripped, non-strict by default
but — unlike the reference, which only prints — mismatches are returned so callers can assert on them."""
import torch


def load_state_dict(module, state_dict, strict=False):
    own = module.state_dict()
    missing = [k for k in own if k not in state_dict]
    unexpected, mismatched, ok = [], [], {}
    for k, v in state_dict.items():
        if k not in own:
            unexpected.append(k)
        elif tuple(own[k].shape) != tuple(v.shape):
            mismatched.append((k, tuple(own[k].shape), tuple(v.shape)))
        else:
            ok[k] = v
    if strict and (missing or unexpected or mismatched):
        raise RuntimeError("checkpoint mismatch: missing=%s unexpected=%s shape=%s" % (missing, unexpected, mismatched))
    module.load_state_dict(ok, strict=False)
    return dict(missing=missing, unexpected=unexpected, mismatched=mismatched)


def load_checkpoint(model, filename, map_location="cpu", strict=False):
    ckpt = torch.load(filename, map_location=map_location)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    report = load_state_dict(getattr(model, "module", model), sd, strict)
    return ckpt, report
