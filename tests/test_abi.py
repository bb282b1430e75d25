"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares,
the ctypes signature table is complete, module parameters mirror the reference's state_dict, and the product path
refuses CPU tensors instead of silently falling back."""
import os
import re

import pytest
import torch

from megreader_amd import _lib


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    syms = _lib.header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libmegreader_hip.so does not export %s" % s
    assert lib.mr_abi_version() == 3 == _lib.ABI_VERSION


def test_signature_table_matches_header():
    with open(_lib.HEADER_PATH) as f:
        text = f.read()
    for name in _lib.header_symbols():
        if name in _lib.HOST_ONLY:
            continue
        assert name in _lib.SIGNATURES, name
        decl = re.search(r"\bint\s+%s\s*\(([^;]*?)\)\s*;" % name, text, re.S).group(1)
        nargs = len([a for a in decl.split(",") if a.strip()])
        assert nargs == len(_lib.SIGNATURES[name]), (name, nargs, len(_lib.SIGNATURES[name]))
        # pointer / integer / float kinds agree position by position
        for code, arg in zip(_lib.SIGNATURES[name], [a.strip() for a in decl.split(",")]):
            if code == "p":
                assert "*" in arg, (name, arg)
            elif code == "s":
                assert "hipStream_t" in arg, (name, arg)
            elif code == "l":
                assert "long long" in arg and "*" not in arg, (name, arg)
            elif code == "f":
                assert arg.startswith("float") and "*" not in arg, (name, arg)
            elif code == "d":
                assert arg.startswith("double") and "*" not in arg, (name, arg)
            else:
                assert arg.startswith("int") and "*" not in arg, (name, arg)


def test_no_torch_types_in_header():
    with open(_lib.HEADER_PATH) as f:
        text = f.read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # strip comments
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code


def test_modules_mirror_reference_state_dict(golden_dir):
    from megreader_amd.backbones import crnn_backbone
    from megreader_amd.decoders import CRNNDecoder
    golden = torch.load(os.path.join(golden_dir, "crnn_golden.pt"), weights_only=False)

    class BasicModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = crnn_backbone()
            self.decoder = CRNNDecoder(in_channels=512)

    torch.manual_seed(golden['weight_seed'])
    m = BasicModel()
    state = m.state_dict()
    assert list(state.keys()) == golden['state_keys']
    for k, v in state.items():
        assert tuple(v.shape) == golden['state_shapes'][k], k
        # same default initialisation (same RNG consumption order) as the reference modules
        s, a = golden['state_checksums'][k]
        assert abs(float(v.double().sum()) - s) <= 1e-9 * max(1.0, a), k


def test_product_path_has_no_cpu_fallback():
    from megreader_amd.backbones import crnn_backbone
    from megreader_amd.nn import functional as F
    with pytest.raises(NotImplementedError):
        crnn_backbone()(torch.zeros(1, 3, 32, 32))
    with pytest.raises(NotImplementedError):
        F.ctc_loss_logits(torch.zeros(4, 1, 5), torch.zeros(1, 2, dtype=torch.int64), None, torch.tensor([1]))


def test_product_code_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "megreader_amd")
    for dirpath, _, files in os.walk(root):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, fn)


def test_decode_persist_host_queries_without_a_gpu():
    """Host-only entry points of the persistent decode kernels (include/megreader_hip.h): workspace sizes follow the group layout of
    csrc/decode_persist.hip (groups of 4 rows up to 32 samples, of 8 beyond; + the XCC-id exchange + 256 status bytes), and without
    a device (or with too few CUs) `mr_decode_persist_ok` says no, so the decoder keeps its per-step launches."""
    lib = _lib.load()
    w4, w8, w32, w33 = (lib.mr_decode_persist_ws_bytes(n) for n in (4, 8, 32, 33))
    group4 = w4 - 4096 - 256
    assert group4 > 0 and w8 == 2 * group4 + 4096 + 256 and w32 == 8 * group4 + 4096 + 256
    assert (w33 - 4096 - 256) % 5 == 0 and (w33 - 4096 - 256) // 5 > group4          # 5 groups of 8 rows
    b4 = lib.mr_decode_persist_bwd_ws_bytes(4)
    assert lib.mr_decode_persist_bwd_ws_bytes(32) == 8 * (b4 - 4096 - 256) + 4096 + 256
    assert lib.mr_decode_persist_bwd_ws_bytes(33) == 0                                # the backward kernel stops at 32 samples
    if not torch.cuda.is_available():
        assert lib.mr_decode_persist_ok(_lib.dtype_code(torch.bfloat16), 16, 64, 512, 552) == 0
        assert lib.mr_decode_persist_bwd_ok(_lib.dtype_code(torch.bfloat16), 16, 64, 512, 552) == 0
