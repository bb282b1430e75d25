"""Host-side logic of the training path that needs no GPU: the struct mirror of the batched-prep job table, the grid
arithmetic shared between prep.py and the kernel, the pre-zeroed BatchNorm scratch arena, gradient-sink activation
rules, the wgrad row-table cache and the PMC post-processing used for bench.py's roofline.traffic."""
import ctypes
import json
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_prep_job_struct_mirror_matches_the_library():
    from megreader_amd import _lib
    from megreader_amd.nn import prep
    lib = _lib.load()
    assert ctypes.sizeof(prep.PrepJob) == lib.mr_sizeof_prep_job()
    # every key a job description carries is a struct field (or the host-only element count)
    for job in (prep.conv_job(1, (27, 1, 9, 3), 2, 3, 64, 3, 3, 3, 8, 64),
                prep.matrix_job(1, 512, 2, 512, 3, 1024, 1024, 512, 256),
                prep.bias_job(1, 2, 3, 1024, 256), prep.stem_job(1, (27, 1, 9, 3), 2, 3)):
        fields = {n for n, _ in prep.PrepJob._fields_}
        assert set(job) - fields == {"total"}


def test_prep_job_block_counts():
    from megreader_amd.nn import prep
    # conv: ceil(K/64) * ceil(R*S*Cpad/64); matrix: ceil(R/64)*ceil(C/64); bias: ceil(R/4096); stem: 1
    assert prep.job_blocks(prep.conv_job(0, (1, 1, 1, 1), 0, 0, 512, 512, 3, 3, 512, 512)) == 8 * 72
    assert prep.job_blocks(prep.conv_job(0, (1, 1, 1, 1), 0, 0, 20, 16, 3, 2, 16, 24)) == 1 * 2
    assert prep.job_blocks(prep.matrix_job(0, 512, 0, 512, 0, 2048, 1024, 512, 256)) == 16 * 8
    assert prep.job_blocks(prep.matrix_job(0, 512, 0, 512, 0, 40, 38, 512, 0)) == 1 * 8
    assert prep.job_blocks(prep.bias_job(0, 0, 0, 1024, 256)) == 1
    assert prep.job_blocks(prep.bias_job(0, 0, 0, 5000, 0)) == 2
    assert prep.job_blocks(prep.stem_job(0, (1, 1, 1, 1), 0, 3)) == 1


def test_zero_arena_hands_out_each_slice_once_per_reset():
    from megreader_amd.nn.functional import ZeroArena
    dev = torch.device("cpu")
    ZeroArena.arenas.pop(dev, None)
    a = ZeroArena.take(dev, 100)
    b = ZeroArena.take(dev, 7)
    assert a.numel() == 128 and b.numel() == 32 and a.data_ptr() != b.data_ptr()   # 32-double granularity
    assert float(a.abs().sum()) == 0.0 and float(b.abs().sum()) == 0.0
    a.fill_(3.0)
    b.fill_(5.0)
    ZeroArena.reset(dev)
    a2 = ZeroArena.take(dev, 100)
    assert a2.data_ptr() == a.data_ptr() and float(a2.abs().sum()) == 0.0           # rewound and re-zeroed
    # exhaustion -> None (callers fall back to their own memset), never a dirty slice
    assert ZeroArena.take(dev, ZeroArena.SIZE) is None
    ZeroArena.arenas.pop(dev, None)


def test_grad_sink_is_only_active_while_it_is_the_param_grad():
    from megreader_amd.nn.functional import grad_sink
    p = torch.nn.Parameter(torch.zeros(4, 3))
    assert grad_sink(p, (4, 3)) is None                       # no sink attached
    sink = torch.zeros(4, 3)
    p._mr_grad_sink = sink
    assert grad_sink(p, (4, 3)) is None                       # CPU tensors never qualify (and .grad is None)
    p.grad = torch.zeros(4, 3)
    assert grad_sink(p, (4, 3)) is None                       # .grad is a different tensor -> autograd path


def test_wgrad_rowtab_cache_builds_once_per_geometry():
    from megreader_amd.nn import functional as F
    F._ROWTABS.clear()
    dev = torch.device("cpu")
    g1 = (4, 8, 32, 256, 3, 3, 1, 1, 1, 1, 1, 1, 8, 32)
    t1, build1 = F._wgrad_rowtab(dev, g1)
    t2, build2 = F._wgrad_rowtab(dev, g1)
    t3, build3 = F._wgrad_rowtab(dev, (2,) + g1[1:])
    assert build1 == 1 and build2 == 0 and build3 == 1
    assert t1 is t2 and t1.shape == (4 * 8 * 32, 2) and t1.dtype == torch.int32 and t3.shape[0] == 2 * 8 * 32
    F._ROWTABS.clear()


def test_pmc_post_processing(tmp_path):
    fetch = tmp_path / "f.txt"
    write = tmp_path / "w.txt"
    fetch.write_text("_ZN2mr20igemm_nt_glds_kernelIDF16bLi96ELi128ELi2ENS_8EpiStoreIDF16bEEE\n"
                     "   FETCH_SIZE                   n= 30 mean=4.8e+04\n"
                     "void mr::igemm_tn_glds_kernel<2, true>\n   FETCH_SIZE                   n= 30 mean=4.2e+04\n"
                     "_ZN2mr18maxpool_fwd_kernelIDF16bEEvPKT_\n   FETCH_SIZE                   n=  3 mean=1e+04\n")
    write.write_text("_ZN2mr20igemm_nt_glds_kernelIDF16bLi96ELi128ELi2ENS_8EpiStoreIDF16bEEE\n"
                     "   WRITE_SIZE                   n= 30 mean=2.4e+04\n"
                     "void mr::igemm_tn_glds_kernel<2, true>\n   WRITE_SIZE                   n= 30 mean=3.8e+04\n")
    out = tmp_path / "t.json"
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "pmc_to_json.py"), str(fetch), str(write),
                           str(out)], stdout=subprocess.DEVNULL)
    d = json.load(open(out))
    # the JSON is stamped with the hash of the kernel sources it was measured on (bench.py ignores a stale file)
    assert len(d.pop("_kernel_source_hash")) >= 8
    assert set(d) == {"igemm_nt_kernel<bf16,96,128,conv>", "igemm_tn_kernel<bf16,conv>"}
    nt = d["igemm_nt_kernel<bf16,96,128,conv>"]
    assert nt["bytes_per_launch"] == 2 * 4.8e4 * 1024 + 2.4e4 * 1024 and nt["launches"] == 30   # FETCH_SIZE doubled


def test_synthetic_batches_match_the_oracle_generators():
    """bench.py's product leg imports nothing from oracle/: its generator must equal the oracle's (the one the
    golden fixtures were made with)."""
    from megreader_amd.synthetic import recognition_batch, recognition_batch_2d
    from oracle.crnn import synthetic_batch
    from oracle.res50ppm import synthetic_batch_2d
    a, b = recognition_batch(5, 32, 128, seed=3), synthetic_batch(5, 32, 128, seed=3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    a, b = recognition_batch_2d(5, 32, 128, seed=4, max_len=3), synthetic_batch_2d(5, 32, 128, seed=4, max_len=3)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_msgpack_record_unpack_follows_the_reference_rules():
    """data/unpack_msgpack_data.py:27-52: b'img' -> decoded image (RGB -> BGR in mode 'BGR'), other bytes -> str,
    containers recurse, byte keys decoded.  The image stays uint8 here (the float conversion runs on the GPU)."""
    import io
    import msgpack
    import numpy as np
    from PIL import Image
    from megreader_amd.data import UnpackMsgpackData
    rgb = np.random.default_rng(0).integers(0, 256, (7, 11, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format='PNG')
    rec = msgpack.dumps({b'img': buf.getvalue(), b'gt': 'Hello'.encode(), b'lines': [b'a', {b'poly': [1, 2.5]}],
                         b'n': 3}, use_bin_type=False)
    out = UnpackMsgpackData()(rec, data_id='id0')
    assert out['gt'] == 'Hello' and out['lines'] == ['a', {'poly': [1, 2.5]}] and out['n'] == 3
    assert out['data_id'] == 'id0'
    assert out['img'].dtype == np.uint8 and out['img'].flags['C_CONTIGUOUS']
    assert np.array_equal(out['img'], rgb[:, :, ::-1])
    assert np.array_equal(UnpackMsgpackData(mode='RGB').convert(rec)['img'], rgb)


# ---------------------------------------------------------------------------------------------------------------
# LMDB container reader (megreader_amd/data/lmdb_reader.py; SURVEY.md §8 f3, reference data/lmdb_dataset.py:59-88)
# ---------------------------------------------------------------------------------------------------------------
def _lmdb_records(n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    recs = {}
    for i in range(n):
        size = int(rng.choice([0, 1, 7, 100, 2029, 2030, 2031, 5000, 70000]))   # in-page and overflow (BIGDATA) values
        recs[("image-%09d" % i).encode()] = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    return recs


def test_lmdb_reader_roundtrip_named_databases(tmp_path):
    """Writer -> reader over named databases (what `env.open_db(b'image')` opens), 3-level trees, values on overflow
    pages, missing keys, key order of the cursor, and the format constants the reader relies on."""
    import struct

    from megreader_amd.data import lmdb_reader as L
    img = _lmdb_records(3000, 1)
    extra = {b"k%d" % i: b"v" * (i % 50) for i in range(10)}
    path = str(tmp_path / "db")
    fname = L.write_environment(path, {b"image": img, b"extra": extra})
    raw = open(fname, "rb").read()
    assert struct.unpack_from("<I", raw, 16)[0] == 0xBEEFC0DE and struct.unpack_from("<I", raw, 20)[0] == 1
    assert struct.unpack_from("<H", raw, 10)[0] == L.P_META and len(raw) % 4096 == 0
    env = L.open(path, max_dbs=1, lock=False)
    assert env.psize == 4096 and env.stat()["entries"] == 2          # two named databases in MAIN
    txn = env.begin(db=env.open_db(b"image"))
    assert txn.stat()["entries"] == len(img) and txn.stat()["depth"] >= 2
    for k, v in img.items():
        assert txn.get(k) == v
    assert txn.get(b"image-999999999") is None and txn.get(b"") is None and txn.get(b"zzz", b"dflt") == b"dflt"
    keys = [k for k, _ in txn.cursor()]
    assert keys == sorted(img) and len(keys) == len(img)
    t2 = env.begin(db=env.open_db(b"extra"))
    assert dict(t2.cursor()) == extra
    import pytest
    with pytest.raises(L.Error):
        env.open_db(b"nope")
    env.close()


def test_lmdb_reader_unnamed_db_small_pages_and_key_order(tmp_path):
    """MAIN-database records, 512-byte pages (deep tree), and memcmp key order with shorter-first ties."""
    from megreader_amd.data import lmdb_reader as L
    recs = {b"a": b"1", b"ab": b"2", b"b": b"3", b"a\x00": b"4", b"\xff": b"5", b"num-samples": b"5"}
    recs.update({b"key-%05d" % i: (b"%d" % i) * (i % 40) for i in range(2000)})
    path = str(tmp_path / "db2")
    L.write_environment(path, {None: recs}, psize=512)
    with L.open(path) as env:
        assert env.psize == 512
        txn = env.begin()
        assert txn.stat()["depth"] >= 3
        for k, v in recs.items():
            assert txn.get(k) == v
        assert [k for k, _ in txn.cursor()] == sorted(recs)


def test_lmdb_image_store_mirrors_reference_usage(tmp_path):
    """`LMDBImageStore` = the reference's prepare / search_image / default_unpack sequence on top of the reader."""
    import io

    import numpy as np
    from PIL import Image

    from megreader_amd.data import lmdb_reader as L
    rng = np.random.default_rng(0)
    imgs, recs = {}, {}
    for i in range(5):
        a = rng.integers(0, 256, (32, 100 + i, 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, format="PNG")
        imgs["id%d" % i] = a
        recs[("id%d" % i).encode()] = buf.getvalue()
    path = str(tmp_path / "imgdb")
    L.write_environment(path, {b"image": recs})
    store = L.LMDBImageStore([path])
    for k, a in imgs.items():
        meta = store.default_unpack(k, {"db_path": path})
        assert meta["image"].dtype == np.uint8 and np.array_equal(meta["image"], a[:, :, ::-1])
    import pytest
    with pytest.raises(AssertionError):
        store.search_image("missing", path)
    store.close()


def test_tn_taps_kernel_index_arithmetic_emulation():
    """Lane-level numpy replay of csrc/tn_taps.hip (stream table, LDS-DMA image + swizzle, ds_read_b64_tr_b16 gather, ring
    wrap, edge masks, MFMA operand layout, split map) against a direct convolution weight gradient: exact on integers."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "emulate_tn_taps", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emulate_tn_taps.py"))
    emu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emu)
    emu.check(3, 4, 9, 64, 64, 1, 1)       # one chunk range, one tile
    emu.check(3, 4, 33, 64, 72, 1, 2)      # CRNN conv4/5 geometry, ragged Cout tile, two splits
    emu.check(2, 6, 10, 64, 64, 2, 2)      # dilation 2


def test_kernel_choice_host_logic():
    """Host-side dispatch rules of the conv kernels (no GPU needed: the queries launch nothing; 256 CUs assumed when no device
    is visible): the 272x256 one-round tile for 33792 x 512, the 288x128 tile for the badly quantised N = 256 dgrad, head / tail
    where neither fits, and the all-taps wgrad eligibility (3x3, stride 1, padding == dilation, Cin % 64 == 0, W <= 62)."""
    from megreader_amd import _lib
    lib = _lib.load()
    code = lib.mr_nt_kernel_code
    assert code(1, 65536, 256, 2304, 256) == 256256          # conv3 fwd: exactly one round of 256x256 tiles
    assert code(1, 33792, 512, 4608, 512) == 272256          # conv5: 250 tiles of 272 rows, one round
    assert code(1, 33792, 256, 4608, 512) == 288128          # conv4 dgrad: 236 tiles of 288x128
    assert code(1, 70000, 256, 576, 0) == 256257             # head (one round) + 4-wave tail
    assert code(1, 262144, 128, 576, 64) == 128128           # conv1 fwd: 4-wave tile
    assert code(0, 33792, 512, 4608, 512) // 1000 <= 128     # f32: never the bf16 big tiles
    run = lib.mr_tn_taps_would_run
    old = lib.mr_set_tn_taps(1)
    try:
        #          N   H   W  Cin ldx Cout lddy R  S sh sw ph pw dh dw Ho  Wo
        assert run(256, 4, 33, 512, 512, 512, 512, 3, 3, 1, 1, 1, 1, 1, 1, 4, 33) == 1      # conv5
        assert run(256, 8, 32, 128, 128, 256, 256, 3, 3, 1, 1, 1, 1, 1, 1, 8, 32) == 1      # conv2
        assert run(256, 16, 64, 64, 64, 128, 128, 3, 3, 1, 1, 1, 1, 1, 1, 16, 64) == 0      # conv1: W = 64 > 62
        assert run(256, 2, 34, 512, 512, 512, 512, 2, 2, 1, 1, 0, 0, 1, 1, 1, 33) == 0      # conv6: 2x2
        assert run(64, 8, 32, 64, 64, 128, 128, 3, 3, 2, 2, 1, 1, 1, 1, 4, 16) == 0         # strided
        assert run(64, 8, 32, 48, 48, 128, 128, 3, 3, 1, 1, 1, 1, 1, 1, 8, 32) == 0         # Cin % 64 != 0
        assert run(64, 8, 24, 64, 64, 64, 64, 3, 3, 1, 1, 2, 2, 2, 2, 8, 24) == 1           # dilation 2 == padding 2
        assert run(64, 8, 32, 64, 64, 64, 64, 3, 3, 1, 1, 2, 2, 2, 2, 8, 32) == 0           # ... but 2*(32+2)+2 > 64 rows of halo
        assert run(64, 8, 32, 64, 64, 64, 64, 3, 3, 1, 1, 1, 1, 2, 2, 6, 30) == 0           # padding != dilation
        assert run(32, 4, 16, 256, 256, 256, 256, 3, 3, 1, 1, 1, 1, 1, 1, 4, 16) == 0       # 2048 output pixels < tn_taps_min_p
        oldp = _lib.set_tuning(tn_taps_min_p=0)
        assert run(32, 4, 16, 256, 256, 256, 256, 3, 3, 1, 1, 1, 1, 1, 1, 4, 16) == 1
        _lib.set_tuning(**oldp)
        lib.mr_set_tn_taps(0)
        assert run(256, 4, 33, 512, 512, 512, 512, 3, 3, 1, 1, 1, 1, 1, 1, 4, 33) == 0
    finally:
        lib.mr_set_tn_taps(old)


def test_bench_json_is_the_last_stdout_line_even_with_native_stdio_output():
    """bench.py's contract is ONE JSON line on stdout; RCCL prints a version banner through C stdio, which libc flushes at
    process exit -- after everything Python printed.  emit_last_line flushes the native buffer first and closes stdout behind
    the JSON line: text a native library writes before or after must not follow it."""
    code = ("import ctypes, sys; sys.path.insert(0, %r); import bench; libc = ctypes.CDLL(None); "
            "libc.printf(b'banner (buffered in C stdio)\\n'); print('python line'); "
            "bench.emit_last_line('{\"ok\": 1}'); libc.printf(b'late banner\\n'); print('late python line')") % REPO
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    lines = out.stdout.strip().splitlines()
    assert lines[-1] == '{"ok": 1}' and "banner (buffered in C stdio)" in lines and "late banner" not in lines, lines


# ---------------------------------------------------------------------------------------------------------------
# dropin.fuse_optimizers(): torch.optim.Adam / SGD stay classes -- isinstance, subclassing and the CPU
# fall-through keep working; the fused construction itself needs a GPU (tests/test_dropin_fast_gpu.py).
# ---------------------------------------------------------------------------------------------------------------
def test_fused_optimizer_alias_is_still_a_class():
    import torch
    from megreader_amd import dropin
    adam0, sgd0 = torch.optim.Adam, torch.optim.SGD
    try:
        assert dropin.fuse_optimizers() == ["Adam", "SGD"]
        alias = torch.optim.Adam
        dropin.fuse_optimizers()                                   # idempotent: not wrapped twice
        assert torch.optim.Adam is alias and alias._mr_original is adam0
        assert isinstance(alias, type) and issubclass(alias, adam0) and alias.__name__ == "Adam"
        w = torch.nn.Parameter(torch.randn(4, 3))
        opt = torch.optim.Adam((p for p in [w]), lr=1e-2)          # CPU parameters (a generator, like model.parameters())
        assert type(opt) is alias and isinstance(opt, adam0) and isinstance(opt, torch.optim.Optimizer)
        assert opt.param_groups[0]["params"][0] is w and opt.defaults["lr"] == 1e-2
        w.grad = torch.ones_like(w)
        before = w.detach().clone()
        opt.step()
        assert not torch.equal(before, w.detach())

        class Mine(torch.optim.SGD):                               # third-party style subclass of the aliased name
            def extra(self):
                return len(self.param_groups)
        mine = Mine([w], lr=0.1, momentum=0.9)
        assert isinstance(mine, sgd0) and mine.extra() == 1 and mine.defaults["momentum"] == 0.9
        # positional hyper-parameters / unsupported options fall through to torch's own optimizer
        assert isinstance(torch.optim.SGD([w], 0.1), sgd0)
        assert isinstance(torch.optim.Adam([w], lr=1e-3, amsgrad=True), adam0)
    finally:
        torch.optim.Adam, torch.optim.SGD = adam0, sgd0


def test_sync_bn_switch_is_an_explicit_hook():
    """backbones/resnet.py:bn() follows the callable registered by dropin.install() (the reference's own config.sync_bn), not a
    `config` module that merely happens to be importable."""
    import sys
    import types
    from megreader_amd.backbones import resnet
    from megreader_amd.nn import BatchNorm2d
    old = resnet._sync_bn_source
    fake = types.ModuleType("config")
    fake.sync_bn = True
    had = sys.modules.get("config")
    sys.modules["config"] = fake
    try:
        resnet.set_sync_bn_source(None)
        assert type(resnet.bn(8)) is BatchNorm2d                   # a stray module named config changes nothing
        resnet.set_sync_bn_source(lambda: 1)                       # truthiness, like `if config.sync_bn:`
        from megreader_amd.apex.parallel import SyncBatchNorm
        assert type(resnet.bn(8)) is SyncBatchNorm
    finally:
        resnet.set_sync_bn_source(old)
        if had is None:
            del sys.modules["config"]
        else:
            sys.modules["config"] = had


def test_bench_shape_table_and_conv_call_normalisation(tmp_path):
    """bench.py --shape-table: the dgrad variants with extra epilogue operands and the statistics forward map onto the plain
    calls' argument layout (FLOPs and geometry from ONE place), and the table lists M x N x K of the implicit GEMM per
    entry point (forward: pixels x Cout x taps*Cin; dgrad: input pixels x Cin x taps*Cout; wgrad: Cout x taps*Cin x pixels)."""
    import bench
    from megreader_amd import _lib
    lib = _lib.load()
    geom = (4, 8, 16, 64, 64, 128, 128, 3, 3, 1, 1, 1, 1, 1, 1, 8, 16)      # N H W Cin ldx Cout ldy R S sh sw ph pw dh dw Ho Wo
    fwd = (1, 0, 0, 0, 0, 0) + geom
    stats = (1, 0, 0, 0, 0, 0) + geom[:6] + geom[7:]                          # no relu / ldy, a sums pointer instead
    dgrad = (1, 0, 0, 0) + geom
    dgrad_add = (1, 0, 0, 0, 0) + geom
    dgrad_bnb = (1, 0, 0, 0) + (0,) * 7 + geom
    wgrad = (1, 0, 0, 0, 0) + geom
    flops = 2.0 * 4 * 8 * 16 * 128 * 9 * 64
    assert bench.conv_flops("mr_conv2d_fwd", fwd)[0] == flops
    assert bench.conv_flops("mr_conv2d_fwd_stats", stats)[0] == flops
    for name, args in (("mr_conv2d_dgrad", dgrad), ("mr_conv2d_dgrad_add", dgrad_add), ("mr_conv2d_dgrad_bnb", dgrad_bnb)):
        assert bench.conv_flops(name, args)[0] == flops, name
        assert bench.normalize_conv_call(name, args) == ("mr_conv2d_dgrad", dgrad)
    assert bench.conv_flops("mr_conv2d_wgrad", wgrad)[0] == flops
    path = str(tmp_path / "shapes.txt")
    recs = [("mr_conv2d_fwd", fwd, 0.02), ("mr_conv2d_fwd", fwd, 0.04), ("mr_conv2d_dgrad_add", dgrad_add, 0.05),
            ("mr_conv2d_wgrad", wgrad, 0.01)]
    bench.write_shape_table(path, lib, recs, "bf16", steps=2)
    rows = open(path).read().splitlines()
    assert rows[0].startswith("per_step") and len(rows) == 4
    body = "\\n".join(rows[1:])
    assert " 512    128     576 " in body.replace("  ", " ").replace("  ", " ") or "512" in body      # forward: M N K
    by_entry = {r.split()[3]: r.split() for r in rows[1:]}
    assert by_entry["mr_conv2d_fwd"][0] == "1.0" and abs(float(by_entry["mr_conv2d_fwd"][1]) - 30.0) < 1e-6
    m, n, k = (int(v) for v in by_entry["mr_conv2d_fwd"][5:8])
    assert (m, n, k) == (4 * 8 * 16, 128, 9 * 64)
    m, n, k = (int(v) for v in by_entry["mr_conv2d_dgrad_add"][5:8])
    assert (m, n, k) == (4 * 8 * 16, 64, 9 * 128)
    m, n, k = (int(v) for v in by_entry["mr_conv2d_wgrad"][5:8])
    assert (m, n, k) == (128, 9 * 64, 4 * 8 * 16)


def test_pmc_label_mapping_covers_the_kernel_families():
    """tools/pmc_to_json.py keys the PMC traffic by bench.py's kernel labels; the 8-wave kernels are labelled by their tile
    (WM x WN waves of TM x TN MFMA tiles), so that e.g. the 128x128 launches of a ResNet step -- which run on the 8-wave
    variant since round 5 -- find their traffic under the label bench.py gives them."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "pmc_to_json", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pmc_to_json.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lab = mod.label
    assert lab("_ZN2mr19igemm_nt_big_kernelIDF16bLi2ELi4ELi4ELi2ELi2ENS_8EpiStoreIDF16bEEEEvNS_6NtArgsE") == "igemm_nt_kernel<bf16,128,128,conv>"
    assert lab("_ZN2mr19igemm_nt_big_kernelIDF16bLi4ELi2ELi2ELi2ELi2ENS_8EpiStoreIDF16bEEEEv") == "igemm_nt_kernel<bf16,128,64,conv>"
    assert lab("_ZN2mr19igemm_nt_big_kernelIDF16bLi1ELi8ELi17ELi2ELi2ENS_8EpiStoreIDF16bEEEEv") == "igemm_nt_kernel<bf16,272,256,conv>"
    assert lab("_ZN2mr19igemm_nt_big_kernelIDF16bLi2ELi4ELi9ELi2ELi2ENS_8EpiStoreIDF16bEEEEv") == "igemm_nt_kernel<bf16,288,128,conv>"
    assert lab("_ZN2mr19igemm_nt_big_kernelIDF16bLi2ELi4ELi8ELi4ELi2ENS_8EpiStoreIDF16bEEEEv") == "igemm_nt_kernel<bf16,256,256,conv>"
    assert lab("_ZN2mr19igemm_nt_big_kernelIDF16bLi2ELi4ELi3ELi2ELi0ENS_8EpiStoreIDF16bEEEEv") is None      # dense GEMM: not a conv label
    assert lab("_ZN2mr20igemm_nt_glds_kernelIDF16bLi64ELi128ELi2ENS_8EpiStoreIDF16bEELi4EEEv") == "igemm_nt_kernel<bf16,64,128,conv>"
    assert lab("_ZN2mr20igemm_tn_taps_kernelILi2ELi1ELi0ELb0EEEvNS_7TapArgsE") == "igemm_tn_taps_kernel<bf16,3x3>"
    assert lab("_ZN2mr20igemm_tn_glds_kernelILi2ELb1ELi0EEEvNS_6TnArgsE") == "igemm_tn_kernel<bf16,conv>"
    # the stamp bench.py compares: both sides hash the same files
    assert len(mod.kernel_source_hash()) == 16
