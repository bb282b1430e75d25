"""Import shims for the reference's NON-arithmetic third-party dependencies that are absent from this image
(apex is handled separately by megreader_amd.apex; SURVEY.md §8b/§8c lists them).  Nothing here computes
anything on the hot path: they only let ``import structure.model`` / ``import decoders`` of the unmodified
reference succeed.

  anyconfig.load(path)  -> yaml.safe_load            (reference concern/config.py:12-14)
  munch.munchify(d)     -> attribute dict            (reference concern/config.py:14)
  editdistance.eval     -> plain DP edit distance    (reference structure/measurers/sequence_recognition_measurer.py)
  tensorboardX.SummaryWriter -> no-op writer          (reference concern/log.py:75)
  everything else       -> inert placeholder modules that raise on use
"""
import importlib
import importlib.abc
import importlib.machinery
import sys
import types


class _Placeholder(types.ModuleType):
    """Module whose attributes are inert callables/classes; using them raises a clear error."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name

        class _Missing(object):
            def __init__(self, *a, **k):
                raise ImportError("%s is a placeholder for a dependency that is not installed" % full)

        _Missing.__name__ = name
        return _Missing


def _attrdict(obj):
    class AttrDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    if isinstance(obj, dict):
        return AttrDict((k, _attrdict(v)) for k, v in obj.items())
    if isinstance(obj, list):
        return [_attrdict(v) for v in obj]
    return obj


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


PLACEHOLDERS = ["cv2", "imgaug", "imgaug.augmenters", "shapely", "shapely.geometry", "pyclipper",
                "hanziconv", "ipdb", "gevent", "gevent.pywsgi", "lmdb", "redis", "boto3", "botocore", "botocore.exceptions",
                "torchvision",
                "torchvision.transforms", "torchvision.datasets", "nori2", "fire", "geventwebsocket",
                "geventwebsocket.handler"]


def install():
    """Register the shims in sys.modules for every dependency that is not importable."""
    def missing(name):
        if name in sys.modules:
            return False
        try:
            return importlib.util.find_spec(name) is None
        except (ImportError, ValueError):
            return True

    if missing("anyconfig"):
        import yaml
        m = types.ModuleType("anyconfig")

        def load(path, *a, **k):
            with open(path) as f:
                return yaml.safe_load(f)

        m.load = load
        sys.modules["anyconfig"] = m
    if missing("munch"):
        m = types.ModuleType("munch")
        m.munchify = _attrdict
        m.Munch = dict
        sys.modules["munch"] = m
    if missing("editdistance"):
        m = types.ModuleType("editdistance")
        m.eval = _edit_distance
        sys.modules["editdistance"] = m
    if missing("tensorboardX"):
        m = types.ModuleType("tensorboardX")

        class SummaryWriter(object):
            """No-op stand-in (concern/log.py:75 constructs one unconditionally): every add_* call is accepted and dropped."""

            def __init__(self, *a, **k):
                pass

            def __getattr__(self, name):
                return lambda *a, **k: None

        m.SummaryWriter = SummaryWriter
        sys.modules["tensorboardX"] = m
    for name in PLACEHOLDERS:
        if missing(name.split(".")[0]) or (name in sys.modules and isinstance(sys.modules[name], _Placeholder)) \
                or (name.split(".")[0] in sys.modules and isinstance(sys.modules[name.split(".")[0]], _Placeholder)):
            if name not in sys.modules:
                mod = _Placeholder(name)
                mod.__path__ = []
                sys.modules[name] = mod
                if "." in name:
                    parent, child = name.rsplit(".", 1)
                    setattr(sys.modules[parent], child, mod)
