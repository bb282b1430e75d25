"""The unmodified reference's structure/model.py resolves to the HIP modules after dropin.install()
(runs only where the reference tree exists, i.e. in the build container; no forward pass on CPU)."""
import os
import subprocess
import sys

import pytest

from oracle import refimport

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, os
sys.dont_write_bytecode = True
sys.path.insert(0, %(repo)r)
os.chdir("/tmp")
import torch
import megreader_amd.dropin as dropin
inst = dropin.install(%(ref)r)
assert "crnn_backbone" in inst["backbones"] and "CRNNDecoder" in inst["decoders"], inst
import structure.model as sm                      # the reference file, unchanged
assert sm.__file__.startswith(%(ref)r)
import backbones, decoders
assert backbones.__file__.startswith(%(ref)r)     # reference package, with the hot-path factories overridden
from concern.charsets import EnglishCharset
args = {"backbone": "crnn_backbone", "decoder": "CRNNDecoder",
        "decoder_args": {"in_channels": 512, "inner_channels": 256, "need_reduce": False, "charset": EnglishCharset()}}
torch.manual_seed(1234)
model = sm.SequenceRecognitionModel(args, torch.device("cpu"))
import megreader_amd.nn as mnn
convs = [m for m in model.modules() if isinstance(m, mnn.Conv2d)]
lstms = [m for m in model.modules() if isinstance(m, mnn.LSTM)]
assert len(convs) == 7 and len(lstms) == 2, (len(convs), len(lstms))
golden = torch.load(os.path.join(%(repo)r, "tests", "golden", "crnn_golden.pt"), weights_only=False)
keys = [k.replace("model.module.", "") for k in model.state_dict().keys()]
assert keys == golden["state_keys"]
for k, v in model.state_dict().items():           # identical default initialisation as the reference modules
    s, a = golden["state_checksums"][k.replace("model.module.", "")]
    assert abs(float(v.double().sum()) - s) <= 1e-9 * max(1.0, a), k
# the DB head resolves to the HIP mirror; the other detection heads / losses still come from the reference
assert decoders.SegDetector.__module__ == "megreader_amd.decoders.seg_detector"
assert decoders.EASTDecoder.__module__.startswith("decoders.")
try:
    model.forward({"image": torch.zeros(1, 3, 32, 64), "label": torch.zeros(1, 32, dtype=torch.int32),
                   "length": torch.ones(1, dtype=torch.int32)})
except NotImplementedError:
    print("DROPIN-OK")
'''


@pytest.mark.skipif(not refimport.available(), reason="reference tree only exists in the build container")
def test_reference_model_file_builds_hip_modules():
    code = SCRIPT % {"repo": REPO, "ref": refimport.REF_ROOT}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "DROPIN-OK" in out.stdout, out.stdout + out.stderr
