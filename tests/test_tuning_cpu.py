"""mr_tuning (include/megreader_hip.h): the library's only process-wide switches -- struct layout shared by the header and the
ctypes mirror, defaults, range validation naming the offending field, the MEGREADER_TUNING environment variable.  Host only."""
import os
import re
import subprocess
import sys

import pytest

from megreader_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_fields():
    src = open(os.path.join(REPO, "include", "megreader_hip.h")).read()
    body = src[src.index("typedef struct mr_tuning {"):src.index("} mr_tuning;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return re.findall(r"int\s+(\w+)(?:\[(\d+)\])?;", body)


def test_struct_layout_matches_the_header():
    fields = _header_fields()
    names = [n for n, dim in fields if n != "reserved"]
    assert tuple(names) == tuple(_lib.TUNING_FIELDS)
    n_ints = sum(int(dim) if dim else 1 for _, dim in fields)
    import ctypes
    assert ctypes.sizeof(_lib.Tuning) == 4 * n_ints
    assert fields[-1][0] == "reserved"


def test_defaults_set_get_and_range_errors():
    lib = _lib.load()
    import ctypes
    d = _lib.Tuning()
    assert lib.mr_tuning_defaults(ctypes.byref(d)) == 0
    assert (d.nt_variant, d.bn_fused, d.bn_onepass, d.skinny_depth, d.nt_big_min_k, d.tn_taps_min_p) == (2, 1, 1, 0, 512, 10000)
    assert all(v == 0 for v in d.reserved)
    base = _lib.get_tuning()
    old = _lib.set_tuning(bn_onepass=0, tn_taps_min_p=4096)
    try:
        now = _lib.get_tuning()
        assert now["bn_onepass"] == 0 and now["tn_taps_min_p"] == 4096
    finally:
        _lib.set_tuning(**old)
    assert _lib.get_tuning()["bn_onepass"] == old["bn_onepass"]
    for bad in (dict(skinny_depth=5), dict(bn_onepass=2), dict(nt_variant=0), dict(nt_big_min_k=8)):
        with pytest.raises(RuntimeError) as e:
            _lib.set_tuning(**bad)
        assert list(bad)[0] in str(e.value)
        assert _lib.get_tuning()[list(bad)[0]] == base[list(bad)[0]]     # a rejected struct changes nothing
    with pytest.raises(KeyError):
        _lib.set_tuning(no_such_field=1)


def _run(env_value):
    env = dict(os.environ, MEGREADER_TUNING=env_value)
    code = ("import sys; sys.path.insert(0, %r); from megreader_amd import _lib; t = _lib.get_tuning(); "
            "print(t['bn_onepass'], t['tn_taps_min_p'], t['nt_deep'])" % REPO)
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)


def test_environment_variable_is_applied_once_at_init():
    r = _run("bn_onepass=0, tn_taps_min_p=123;nt_deep=2")
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout.split()[-3:] == ["0", "123", "2"]
    r = _run("bn_onepass=7")
    assert r.returncode != 0 and "bn_onepass" in (r.stderr + r.stdout)
    r = _run("no_such_switch=1")
    assert r.returncode != 0 and "no_such_switch" in (r.stderr + r.stdout)
