"""CPU oracle for the MI355X hot path -- TEST INFRASTRUCTURE ONLY.

Plain torch-CPU / numpy restatements of the reference's algorithms (each function cites the reference
file:line it follows).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / the timed CPU baseline.  Nothing under ``megreader_amd/`` imports it.

Pinning status (see DESIGN.md "Oracle"):
  * crnn.py / decode.py : PINNED -- bit-compared in this container against the unmodified reference modules
    imported from /root/reference (oracle/gen_golden.py), results committed under tests/golden/.
  * ctc.py (explicit 1-D alpha/beta) : pinned against torch.nn.functional.ctc_loss, which is what the reference
    calls (decoders/crnn.py:48).
  * ctc2d.py, dcn.py : the reference has NO CPU implementation and NO tests for these CUDA-only ops
    (ops/ctc_2d/csrc/ctc2d.h:20, assets/ops/dcn/functions/deform_conv.py:130-131) => "parity unpinned" by the
    reference; anchored on independent cross-checks (H=1 == F.ctc_loss, decoders/ctc_loss2d.py in its valid
    regime, zero-offset DCN == F.conv2d, autograd of the forward restatement).
"""
