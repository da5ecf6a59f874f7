"""Import the upstream reference (read-only at /root/reference) IN THIS CONTAINER ONLY.

TEST INFRASTRUCTURE -- not part of the product.  Used by ``oracle/gen_golden.py`` to
produce the golden vectors committed under ``tests/golden/`` and by the container-only
tests that pin ``oracle/`` against the real reference.  Nothing here travels as
reference code: this file only contains build-owned shims (module *names* the reference
imports but this image lacks) and path plumbing.

Interpreter: /opt/conda/bin/python3.9 (numpy 1.26, scipy 1.7, Cython 0.29, real
bottleneck 1.3.2, real toml) or /usr/local/bin/python3 (numpy 2.2, scipy 1.15, torch;
needs the bottleneck/toml shims below).  pyximport compiles the reference's Cython file
into ``$ADAPTED_REF_BUILD`` (default ``<tmpdir>/adapted_ref_build``), OUTSIDE this repository:
the generated C file quotes the reference's source, so nothing of that build may sit in a
tree that is pushed to the GPU box (HOME is redirected there; /root/reference stays untouched).
"""
import os
import sys
import tempfile
import types

REF_ROOT = os.environ.get("ADAPTED_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)


def build_dir() -> str:
    """Where the reference-derived build outputs go: never inside the repository."""
    d = os.path.abspath(os.environ.get("ADAPTED_REF_BUILD") or os.path.join(tempfile.gettempdir(), "adapted_ref_build"))
    if os.path.commonpath([d, _REPO]) == _REPO:
        raise RuntimeError("ADAPTED_REF_BUILD=%s lies inside the repository; reference-derived files must stay out of it" % d)
    return d


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "adapted"))


def _shim_module(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def install(need_torch: bool = False):
    """Make ``import adapted`` (the reference) work. Returns the imported package."""
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    home = os.path.join(build_dir(), "home")
    os.makedirs(home, exist_ok=True)
    os.environ["HOME"] = home  # pyximport build dir = ~/.pyxbld
    import numpy  # noqa: F401
    import pandas  # noqa: F401  (must precede a bottleneck shim: pandas probes it)

    # attrs: old environments only ship the `attr` package
    try:
        import attrs  # noqa: F401
    except ImportError:
        import attr

        _shim_module("attrs", define=attr.define, field=attr.field)
    # toml -> tomli
    try:
        import toml  # noqa: F401
    except ImportError:
        import tomli

        def _load(f):
            if isinstance(f, (str, os.PathLike)):
                with open(f, "rb") as fh:
                    return tomli.load(fh)
            return tomli.load(f)

        def _dump(d, fh):
            raise NotImplementedError("toml.dump shim: not needed for the hot path")

        _shim_module("toml", load=_load, loads=tomli.loads, dump=_dump)
    # bottleneck: only the real one is acceptable for pinning move_mean/move_var
    try:
        import bottleneck  # noqa: F401
    except ImportError:
        sys.path.insert(0, _HERE)
        from bn_shim import move_mean, move_var  # build-owned f32 recurrences

        _shim_module("bottleneck", move_mean=move_mean, move_var=move_var)
    # pod5: name only (imported by the reference's file_proc, never called here)
    if "pod5" not in sys.modules:
        try:
            import pod5  # noqa: F401
        except ImportError:
            reader = _shim_module("pod5.reader", Reader=object)
            _shim_module("pod5", Reader=object, reader=reader)
    # torch: name only when the CNN path is not needed
    try:
        import torch  # noqa: F401
    except ImportError:
        if need_torch:
            raise

        class _Seq:  # placeholder base class for BoundariesCNN's definition
            def __init__(self, *a, **k):
                pass

        nn = _shim_module("torch.nn", Sequential=_Seq, Conv1d=object, ReLU=object,
                          ConvTranspose1d=object)
        _shim_module("torch", nn=nn, Tensor=object)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import adapted

    return adapted
