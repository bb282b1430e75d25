"""CPU oracle for the MI355X hot path -- TEST INFRASTRUCTURE ONLY.

Plain torch-CPU / numpy restatements of the reference's algorithms (each function cites the reference
file:line it follows).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / the timed CPU baseline (plus the diagnostics under ``tools/``, which are not
the product either).  Nothing under ``megreader_amd/`` imports it.

Pinning status:
  * crnn.py / res50ppm.py / fpn_attention.py / seg_detector.py / ctc_decoder.py / decode.py : PINNED -- bit-compared in this
    container against the unmodified reference modules imported from /root/reference (oracle/gen_golden*.py), results
    committed under tests/golden/.
  * ctc.py (explicit 1-D alpha/beta) : pinned against torch.nn.functional.ctc_loss, which is what the reference
    calls (decoders/crnn.py:48).
  * ctc2d.py : pinned by the reference's python CTCLoss2D (decoders/ctc_loss2d.py) and, round 4, by the reference's CUDA
    extension itself.
  * dcn.py, deform_pool.py : the reference has NO CPU implementation and NO tests for these CUDA-only ops
    (assets/ops/dcn/functions/deform_conv.py:130-131) -- pinned since round 4 by the reference's own extension.
  * pipeline.py (cv2 resize), db_post.py (cv2 / pyclipper / shapely) : PARITY UNPINNED -- the libraries the reference calls are
    not installed in this image.

The reference's GPU extensions as checkers (round 4): oracle/build_ref_ext.sh compiles assets/ops/dcn/src/* and
ops/ctc_2d/csrc/** for gfx950 from /root/reference where they lie into oracle/_ref/ (git-ignored; travels to the GPU box);
oracle/ref_compat/ holds the force-included compatibility header; oracle/gen_golden_dcn.py / gen_golden_ctc2d_ext.py record
their outputs on the MI355X into tests/golden/*_reference_ext.npz.
"""
