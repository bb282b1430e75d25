"""First-replay anomaly of the graphed attention decoder with fixed teacher forcing (tests/test_fpn_attention_gpu.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402
from megreader_amd.decoders import AttentionDecoder  # noqa: E402
from megreader_amd.optim import FusedAdam  # noqa: E402
from megreader_amd.runtime import GraphedTrainStep  # noqa: E402

DEV = "cuda"


def run(first_graph, warmup, fixed, variant=''):
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(11)
    dec = AttentionDecoder(in_channels=256).to(DEV).train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(6, 256, 16, 64, generator=g).to(DEV)
    lab = torch.randint(2, 38, (6, 32), generator=g, dtype=torch.int32).to(DEV)
    ln = torch.randint(3, 11, (6,), generator=g).to(DEV)
    opt = FusedAdam(dec.parameters(), lr=0.0)
    opt.zero_grad()
    hold = {}

    def loss_fn():
        loss, att = dec(x, targets=lab, lengths=ln, train=True)
        hold['loss'], hold['att'] = loss, att
        return loss.mean()
    if first_graph:
        if "fixed1" in variant:
            dec.gt_as_output = False
        s1 = GraphedTrainStep(loss_fn, opt, [], warmup=2)
        print("  graph 1 replays:", [round(float(s1()), 4) for _ in range(3)])
        if "del1" in variant:
            del s1
            hold.clear()
            import gc
            gc.collect()
            torch.cuda.synchronize()
    if "own" in variant:
        os.environ["MEGREADER_CAPTURE_STREAM"] = "own"
    dec.gt_as_output = fixed
    s2 = GraphedTrainStep(loss_fn, opt, [], warmup=warmup)
    many = [round(float(s2()), 4) for _ in range(int(os.environ.get("DIAG_REPLAYS", "0")))]
    if many:
        torch.cuda.synchronize()
        print("  graph 2, %d replays: distinct losses %s" % (len(many), sorted(set(many))))
    for r in range(3):
        l = float(s2())
        torch.cuda.synchronize()
        print("  graph 2 replay %d: loss %.4f  per-sample %s | ln %s lab.sum %d x.sum %.4f flags %s" %
              (r, l, [round(v, 3) for v in hold['loss'].tolist()], ln.tolist(), int(lab.sum()), float(x.double().sum()),
               [v.tolist()[:6] for v in dec.__dict__.get("_flag_cache", {}).values()]))


def no_singletons():
    """VERDICT r4 parity item 4: rule the process-global scratch out -- no zero arena (every kernel zeroes its own scratch), no
    split-reduction workspace (plain atomics, unsplit NT launches), per-step output kernels."""
    from megreader_amd import _lib
    from megreader_amd.nn import functional as F
    from megreader_amd.decoders import attention_decoder as ad
    F.ZeroArena.take = staticmethod(lambda device, n: None)
    _lib.load()
    _lib.set_tuning(nt_ksplit=0, tn_group=1, tn_taps_group=1, bn_onepass=0)
    F._TnDefer.enabled = False
    ad.BATCHED_OUT = False


if __name__ == "__main__":
    variants = sys.argv[1:] or ["del1", "own", "fixed1", ""]
    if "nosingletons" in variants:
        no_singletons()
        variants = [v for v in variants if v != "nosingletons"] or [""]
        print("process-global scratch switched off (arena, split-reduction workspace, resident-grid BatchNorm, deferred wgrads)")
    for v in variants:
        cfg = (True, 1, True, v)
        print("first_graph=%s warmup=%d gt_as_output=%s variant=%s" % cfg)
        run(*cfg)
        os.environ.pop("MEGREADER_CAPTURE_STREAM", None)
