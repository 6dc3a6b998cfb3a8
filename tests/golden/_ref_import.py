"""Import shim for the upstream reference (authoring container only).

The reference lives read-only at /root/reference and needs a handful of
packages that are not installed here (torchvision, iopath, termcolor,
terminaltables).  This module inserts minimal stand-in *modules* into
``sys.modules`` (they only have to satisfy ``import`` statements; none of
the stubbed functions is called on the encoder hot path) and puts the
reference on ``sys.path``.  It is used ONLY by ``make_golden.py`` to
generate fixtures; nothing under ``tests/`` imports it at test time and it
never travels to the GPU box in a usable form (no /root/reference there).
"""
import importlib.machinery
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    mod.__path__ = []  # behave like a package so sub-imports resolve
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def install():
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision", _is_tracing=lambda: False)
        ops = _stub("torchvision.ops", batched_nms=None)
        tv.ops = ops
        # the one torchvision function the salience criterion calls (row N4): torchvision/ops/_box_convert.py,
        # (cx, cy, w, h) -> (cx - w/2, cy - h/2, cx + w/2, cy + h/2).  A three-line coordinate conversion restated
        # here because torchvision is absent; make_golden.py says so next to the fixture it affects.
        def _box_cxcywh_to_xyxy(boxes):
            import torch
            cx, cy, w, h = boxes.unbind(-1)
            return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)
        ops.boxes = _stub("torchvision.ops.boxes", _box_cxcywh_to_xyxy=_box_cxcywh_to_xyxy)
        _stub("torchvision.models")
        _stub("torchvision.models.detection")
        _stub("torchvision.models.detection.image_list", ImageList=object)
    if "iopath" not in sys.modules:
        _stub("iopath")
        _stub("iopath.common")
        names = ["HTTPURLHandler", "LazyPath", "NativePathHandler", "OneDrivePathHandler",
                 "PathHandler", "PathManager", "file_lock", "get_cache_dir"]
        _stub("iopath.common.file_io", **{n: type(n, (), {"register_handler": lambda *a, **k: None,
                                                          "__init__": lambda self, *a, **k: None}) for n in names})
    if "termcolor" not in sys.modules:
        _stub("termcolor", colored=lambda s, *a, **k: s)
    if "terminaltables" not in sys.modules:
        _stub("terminaltables", AsciiTable=object)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
