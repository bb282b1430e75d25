import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs an AMD GPU (run on the MI355X box with `-m gpu`)")


# Collection order of the GPU suite.  The driver runs `pytest -m gpu -x`: one failure hides everything collected after
# it (round 3: one flaky long-tail test hid 361 of 517 tests).  The files that carry the parity evidence of the
# SURVEY section-8 rows run first -- the step that is benchmarked, the published configurations, the full-size and the
# kernel-level tests -- then the per-model golden tests, then the "next" rows (decode, pipeline, DB post-processing,
# RoI pooling); whole-network tests whose tolerances lean on chaotic amplification run last.
_ORDER = [
    "test_timed_step_gpu", "test_published_configs_gpu", "test_fullsize_parity_gpu", "test_kernels_gpu",
    "test_tn_taps_gpu", "test_tn_grouped_gpu", "test_stem_gpu", "test_crnn_gpu", "test_ctc2d_gpu", "test_res50ppm_gpu", "test_fpn_attention_gpu",
    "test_attention_kernels_gpu", "test_dcn_gpu", "test_dcn_reference_gpu", "test_ctc2d_reference_gpu", "test_seg_detector_gpu", "test_dropin_fast_gpu", "test_ddp_gpu",
    "test_ctc_decoder_gpu", "test_decode_gpu", "test_pipeline_gpu", "test_db_post_gpu", "test_deform_pool_gpu",
    "test_deformable_resnet_gpu",
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_ORDER)}

    def key(item):
        stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(stem, len(_ORDER) // 2)

    items.sort(key=key)     # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
