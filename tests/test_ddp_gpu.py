"""The N > 1 bench / training path on a single GPU: CRNN + FusedAdam (gradient sinks, prepared-weight cache) wrapped
in the apex DDP shim over a 1-rank RCCL ("nccl") process group.  With one rank the all-reduce is the identity, so
losses and parameters must follow the un-wrapped run exactly; what is exercised is the plumbing the driver's multi-GPU
bench depends on: parameter broadcast, bucket hooks fired from gradient sinks (`notify_grad_ready`), in-place bucket
all-reduce on the side stream over the flat gradient buffer, finalisation at the end of backward."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.backbones import crnn_backbone  # noqa: E402
from megreader_amd.decoders import CRNNDecoder  # noqa: E402
from megreader_amd.optim import FusedAdam  # noqa: E402
from oracle.crnn import synthetic_batch  # noqa: E402

DEV = "cuda"


class BasicModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = crnn_backbone()
        self.decoder = CRNNDecoder(in_channels=512, inner_channels=256)

    def forward(self, data, *a, **k):
        return self.decoder(self.backbone(data), *a, **k)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(wrap, steps=3):
    torch.manual_seed(0)
    model = BasicModel().to(DEV).train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    net = model
    if wrap:
        from megreader_amd.apex.parallel import DistributedDataParallel
        net = DistributedDataParallel(model, message_size=1 << 20)  # several buckets
    batch = synthetic_batch(8, 32, 128, seed=0)
    img, lab, ln = batch['image'].to(DEV), batch['label'].to(DEV), batch['length'].to(DEV).long()
    losses = []
    first_grads = None
    for _ in range(steps):
        opt.zero_grad()
        loss, _ = net(img, targets=lab, lengths=ln, train=True)
        loss = loss.mean()
        loss.backward()
        if first_grads is None:
            torch.cuda.synchronize()
            first_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    # parameters whose true gradient is zero (conv biases in front of a BatchNorm): Adam turns their round-off noise
    # into +-lr steps, so they are not comparable between two runs with differently ordered f32 atomics
    dead = {k for k, p in model.named_parameters() if p.grad is not None and float(p.grad.abs().max()) < 1e-6}
    return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, dead, first_grads


def test_ddp_shim_single_rank_rccl_matches_plain_run():
    mr.set_compute_dtype(torch.float32)
    try:
        ref_losses, ref_state, dead, ref_grads = _run(False)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1)
        try:
            losses, state, _, grads = _run(True)
        finally:
            dist.destroy_process_group()
    finally:
        mr.set_compute_dtype(torch.bfloat16)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (losses, ref_losses)
    assert len(dead) < 8
    # gradients after the first backward: the (identity) all-reduce + averaging must leave them untouched
    for k in ref_grads:
        scale = float(ref_grads[k].abs().max()) + 1e-6
        assert float((grads[k] - ref_grads[k]).abs().max()) <= 1e-4 * scale + 1e-7, k
    # parameters after 3 Adam steps: element-wise Adam turns atomics-order noise on near-zero gradients into steps of
    # up to lr per iteration, so the bound is 2 * lr * steps (a frozen or doubly-updated parameter would show in the
    # loss trajectory above)
    for k in ref_state:
        if k in dead:
            continue
        if ref_state[k].dtype.is_floating_point:
            scale = float(ref_state[k].abs().max()) + 1e-6
            assert float((state[k] - ref_state[k]).abs().max()) <= 2 * 1e-3 * 3 + 1e-4 * scale, k
        else:
            assert torch.equal(state[k], ref_state[k]), k
