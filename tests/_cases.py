"""The four published configurations (BASELINE.json configs[1..4] = what bench.py times) as memoised oracle runs, shared by
tests/test_timed_step_gpu.py, tests/test_published_configs_gpu.py and tests/test_fullsize_parity_gpu.py so that the CPU
oracle (float32 = the reference's arithmetic, float64 = ground truth of the gradient bars, tests/_parity.py) runs ONCE per
configuration and process.  Each case returns a dict:

    ora        the CPU oracle module (float32, gradients of its training pass in .grad)
    state0     its state_dict before the training forward (what the HIP model loads)
    batch      the seeded synthetic batch (CPU tensors)
    grads32    {name: gradient of the f32 oracle},  grads64 {name: gradient of the f64 oracle}
    out32 / out64   forward outputs of the two oracle passes (per case)
    build()    a fresh HIP model with state0 loaded, on the GPU, in training mode
    loss_fn(model, batch_on_device) -> scalar loss the way the reference's trainer forms it (trainer.py:127 `l.mean()`)
    optimizer(params) -> the fused optimizer of the published YAML with lr = 0 (weights stay put: gradients of later steps
                         remain comparable with the oracle's)
`<case>_hip()` returns the same dict WITHOUT the oracle passes (ora constructed for its seeded initial state only): what a
child process needs to rebuild the HIP side of a case whose gradients it receives through a file (tests/test_timed_step_gpu.py).
"""
import copy
import os
import time

import torch

DEV = "cuda"
_CACHE = {}


def _host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


class _Threads(object):
    """The CPU oracle's summation order (hence its last-bit results) depends on the thread count: use the host's cores for
    the big runs and restore the setting, so that the tests that run later see the oracle they were written against."""

    def __enter__(self):
        self.old = torch.get_num_threads()
        torch.set_num_threads(_host_threads())

    def __exit__(self, *exc):
        torch.set_num_threads(self.old)


def stage_hooks(model, names, store):
    """Record the outputs of the named sub-modules (tensor outputs only) of one forward pass into `store`; returns the handles."""
    mods = dict(model.named_modules())

    def put(name):
        def hook(_m, _inp, out):
            if isinstance(out, torch.Tensor):
                store[name] = out.detach()
        return hook
    return [mods[n].register_forward_hook(put(n)) for n in names if n in mods]


def _f64(ora32, forward, stages=(), store=None):
    ora64 = copy.deepcopy(ora32).double().train()
    ora64.zero_grad()
    handles = stage_hooks(ora64, stages, store) if store is not None else []
    forward(ora64, torch.float64).backward()
    for h in handles:
        h.remove()
    return {k: p.grad.detach().clone() for k, p in ora64.named_parameters() if p.grad is not None}


def _f64_bf16_storage(ora32, forward):
    """Gradients of the float64 oracle when every leaf module's output (and its gradient) is merely STORED in bfloat16 --
    float64 arithmetic, 8-bit mantissas between layers (tools/bf16_storage_sensitivity.py).  The yardstick of the bf16 bars of
    tests/test_timed_step_gpu.py: the distance of THIS from exact float64 is what bf16 storage costs
    whoever does the arithmetic; the HIP bf16 step must stay within 1.5 x of it per parameter group."""
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    from bf16_storage_sensitivity import _Round
    ora64 = copy.deepcopy(ora32).double().train()
    ora64.zero_grad()

    def hook(_m, _inp, out):
        if isinstance(out, torch.Tensor) and out.is_floating_point() and out.requires_grad:
            return _Round.apply(out)
        if isinstance(out, tuple) and out and isinstance(out[0], torch.Tensor) and out[0].requires_grad:
            return (_Round.apply(out[0]),) + tuple(out[1:])
        return None
    # every leaf module's output, as tools/bf16_storage_sensitivity.py does -- except the probability maps of the DB head: the
    # product keeps those sigmoids in float32 whatever the compute dtype (a probability rounded to exactly 1.0 in bf16 makes the
    # balanced-BCE loss and every gradient behind it blow up by 1e8, whoever does the arithmetic)
    handles = [m.register_forward_hook(hook) for m in ora64.modules()
               if not list(m.children()) and not isinstance(m, torch.nn.Sigmoid)]
    forward(ora64, torch.float64).backward()
    for h in handles:
        h.remove()
    return {k: p.grad.detach().clone() for k, p in ora64.named_parameters() if p.grad is not None}


def _yard_thunk(ora, fwd, out64):
    """lazy yardstick pass (the forward closures also record their float64 outputs: keep those of the exact pass)"""
    def run():
        saved = dict(out64)
        with _Threads():
            g = _f64_bf16_storage(ora, fwd)
        out64.clear()
        out64.update(saved)
        return g
    return run


def _grads(ora):
    return {k: p.grad.detach().clone() for k, p in ora.named_parameters() if p.grad is not None}


def _to_dev(batch):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def _recognition_loss(model, b):
    loss, _ = model(b['image'], targets=b['label'], lengths=b['length'].long(), train=True)
    return loss.mean()


# ------------------------------------------------------------------------------------------------ configs[1]: CRNN, N = 256
def crnn_n256_hip():
    from megreader_amd.backbones import crnn_backbone
    from megreader_amd.decoders import CRNNDecoder
    from megreader_amd.optim import FusedAdam
    from oracle.crnn import CRNNOracle, synthetic_batch

    class Model(torch.nn.Module):  # reference structure/model.py:16-24
        def __init__(self):
            super().__init__()
            self.backbone = crnn_backbone()
            self.decoder = CRNNDecoder(in_channels=512, inner_channels=256, need_reduce=False)

        def forward(self, data, *args, **kwargs):
            return self.decoder(self.backbone(data), *args, **kwargs)

    torch.manual_seed(4321)
    ora = CRNNOracle()
    state0 = {k: v.clone() for k, v in ora.state_dict().items()}
    batch = synthetic_batch(256, 32, 128, seed=11)

    def build():
        m = Model()
        m.load_state_dict(state0)
        return m.to(DEV).train()

    return dict(ora=ora, state0=state0, batch=batch, build=build, loss_fn=_recognition_loss,
                optimizer=lambda ps: FusedAdam(ps, lr=0.0), what="CRNN fp32 32x128 N=256")


def crnn_n256():
    if "crnn" in _CACHE:
        return _CACHE["crnn"]
    c = crnn_n256_hip()
    ora, batch = c["ora"], c["batch"]
    lab, ln = batch['label'], batch['length'].long()

    def fwd(m, dt):
        loss, _ = m(batch['image'].to(dt), targets=lab, lengths=ln, train=True)
        return loss.mean()

    t0 = time.time()
    grads64 = _f64(ora, fwd)
    ora.train()
    loss, logp = ora(batch['image'], targets=lab, lengths=ln, train=True)
    loss.mean().backward()
    state1 = {k: v.clone() for k, v in ora.state_dict().items()}   # BN running stats moved by the training forward
    ora.eval()
    with torch.no_grad():
        ev = ora(batch['image'], train=False)
    ora.train()
    print("oracle CRNN N=256 fwd+bwd (f32 and f64) + eval: %.1f s" % (time.time() - t0))
    c.update(state1=state1, grads32=_grads(ora), grads64=grads64,
             out32={"loss": float(loss), "logp": logp.detach(), "eval": ev})
    c["yardstick"] = lambda: _f64_bf16_storage(ora, fwd)
    _CACHE["crnn"] = c
    return c


# ------------------------------------------------------------------------------------ configs[2]: Res50-PPM + 2D-CTC, N = 256
RES50PPM_STAGES = ("backbone.0.maxpool", "backbone.0.layer1", "backbone.0.layer2", "backbone.0.layer3", "backbone.0.layer4",
                   "backbone.1")


def res50ppm_n256_hip():
    from megreader_amd.backbones import resnet50dilated_ppm
    from megreader_amd.decoders import CTCDecoder2D
    from megreader_amd.optim import FusedAdam
    from oracle.res50ppm import Res50PPM2DCTCOracle, synthetic_batch_2d

    class Model(torch.nn.Module):  # structure/model.py:16-24
        def __init__(self):
            super().__init__()
            self.backbone = resnet50dilated_ppm()
            self.decoder = CTCDecoder2D(in_channels=256)

        def forward(self, data, *a, **k):
            return self.decoder(self.backbone(data), *a, **k)

    torch.manual_seed(99)
    ora = Res50PPM2DCTCOracle(dropout=0.0)
    state0 = {k: v.clone() for k, v in ora.state_dict().items()}
    n, height, width = 256, 32, 128          # what bench.py times (bench.py: synthetic_batch_2d(bsz, 32, 128, max_len=3))
    batch = synthetic_batch_2d(n, height, width, seed=5, max_len=3)

    def build():
        m = Model()
        m.load_state_dict(state0, strict=True)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        return m.to(DEV).train()

    return dict(ora=ora, state0=state0, batch=batch, build=build, loss_fn=_recognition_loss,
                optimizer=lambda ps: FusedAdam(ps, lr=0.0), what="Res50-PPM-2DCTC fp32 32x128 N=256")


def res50ppm_n256():
    if "res50ppm" in _CACHE:
        return _CACHE["res50ppm"]
    c = res50ppm_n256_hip()
    ora, batch = c["ora"], c["batch"]
    n, height, width = 256, 32, 128
    lab, ln = batch['label'], batch['length'].long()
    out64 = {}

    def fwd(m, dt):
        loss, pred = m(batch['image'].to(dt), targets=lab, lengths=ln, train=True)
        if dt == torch.float64:
            out64['pred'], out64['loss'] = pred.detach(), loss.detach()
        return loss.mean()

    # stage outputs of both oracle passes: where along the network the forward error against exact arithmetic accrues
    stages = RES50PPM_STAGES
    st64, st32 = {}, {}
    with _Threads():
        t0 = time.time()
        grads64 = _f64(ora, fwd, stages, st64)
        ora.train()
        handles = stage_hooks(ora, stages, st32)
        loss_o, pred_o = ora(batch['image'], targets=lab, lengths=ln, train=True)
        for h in handles:
            h.remove()
        loss_o.mean().backward()
        print("oracle Res50-PPM-2DCTC %dx%d N=%d fwd+bwd (f32 and f64): %.1f s" % (height, width, n, time.time() - t0))
    c.update(grads32=_grads(ora), grads64=grads64, out64=out64, out32={"loss": loss_o.detach(), "pred": pred_o.detach()},
             stages64=st64, stages32=st32)
    c["yardstick"] = _yard_thunk(ora, fwd, out64)
    _CACHE["res50ppm"] = c
    return c


# ------------------------------------------------------------------------------------- configs[3]: FPN50 + attention, N = 32
def fpn_attention_n32_hip():
    from megreader_amd.backbones import Resnet50FPN
    from megreader_amd.decoders import AttentionDecoder
    from megreader_amd.optim import FusedAdam
    from oracle.crnn import synthetic_batch
    from oracle.fpn_attention import FPNAttentionOracle

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = Resnet50FPN(resnet_pretrained=False)
            self.decoder = AttentionDecoder(in_channels=256, gt_as_output=True)

        def forward(self, data, *a, **k):
            return self.decoder(self.backbone(data), *a, **k)

    torch.manual_seed(2024)
    ora = FPNAttentionOracle()
    state0 = {k: v.clone() for k, v in ora.state_dict().items()}
    n = 32
    batch = synthetic_batch(n, 64, 256, seed=21)

    def build():
        m = Model()
        m.load_state_dict(state0, strict=True)
        return m.to(DEV).train()

    return dict(ora=ora, state0=state0, batch=batch, build=build, loss_fn=_recognition_loss,
                optimizer=lambda ps: FusedAdam(ps, lr=0.0), what="FPN50-attention fp32 64x256 N=32")


def fpn_attention_n32():
    if "fpn" in _CACHE:
        return _CACHE["fpn"]
    c = fpn_attention_n32_hip()
    ora, batch, n = c["ora"], c["batch"], 32
    lab, ln = batch['label'], batch['length'].long()
    out64 = {}

    def fwd(m, dt):
        loss, att = m(batch['image'].to(dt), targets=lab, lengths=ln, train=True)
        if dt == torch.float64:
            out64['loss'], out64['att'] = loss.detach(), att.detach()
        return loss.mean()

    with _Threads():
        t0 = time.time()
        grads64 = _f64(ora, fwd)
        ora.train()
        loss_o, att_o = ora(batch['image'], targets=lab, lengths=ln, train=True)
        loss_o.mean().backward()
        print("oracle FPN50-attention 64x256 N=%d fwd+bwd (f32 and f64): %.1f s" % (n, time.time() - t0))
    c.update(grads32=_grads(ora), grads64=grads64, out64=out64, out32={"loss": loss_o.detach(), "att": att_o.detach()})
    c["yardstick"] = _yard_thunk(ora, fwd, out64)
    _CACHE["fpn"] = c
    return c


# --------------------------------------------------------------------------------------- configs[4]: DB detector, 640x640, N = 2
def db_n2_hip():
    from megreader_amd.backbones import deformable_resnet50
    from megreader_amd.decoders import L1BalanceCELoss, SegDetector
    from megreader_amd.optim import FusedSGD
    from megreader_amd.synthetic import detection_batch
    from oracle.res50ppm import _Res50Dilated
    from oracle.seg_detector import SegDetectorOracle

    class Oracle(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = _Res50Dilated(dilate=False, dcn=True)
            self.decoder = SegDetectorOracle(in_channels=[256, 512, 1024, 2048], adaptive=True, k=50)

        def forward(self, image):
            return self.decoder(self.backbone(image))

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = deformable_resnet50(pretrained=False)
            self.decoder = SegDetector(in_channels=[256, 512, 1024, 2048], adaptive=True, k=50)
            self.criterion = L1BalanceCELoss()

        def forward(self, image):
            return self.decoder(self.backbone(image))

    torch.manual_seed(7)
    ora = Oracle()
    state0 = {k: v.clone() for k, v in ora.state_dict().items()}
    n, size = 2, 640
    batch = detection_batch(n, size, seed=3)

    def build():
        m = Model()
        m.load_state_dict(state0, strict=True)
        return m.to(DEV).train()

    def loss_fn(model, b):
        loss, _ = model.criterion(model(b['image']), b)
        return loss

    # seg_detector_db.yaml:85-94: SGD momentum 0.9, weight decay 1e-4 -- with lr = 0 the weights stay put
    return dict(ora=ora, state0=state0, batch=batch, build=build, loss_fn=loss_fn,
                optimizer=lambda ps: FusedSGD(ps, lr=0.0, momentum=0.9, weight_decay=1e-4), what="DB detector fp32 640x640 N=2")


def db_n2():
    if "db" in _CACHE:
        return _CACHE["db"]
    from oracle.seg_detector import l1_balance_ce_loss
    c = db_n2_hip()
    ora, batch, n, size = c["ora"], c["batch"], 2, 640
    out64 = {}

    def fwd(m, dt):
        p = m(batch['image'].to(dt))
        if dt == torch.float64:
            out64.update({k: v.detach() for k, v in p.items()})
        return l1_balance_ce_loss(p, {k: v.to(dt) for k, v in batch.items()})

    with _Threads():
        t0 = time.time()
        grads64 = _f64(ora, fwd)
        ora.train()
        pred_o = ora(batch['image'])
        loss_o = l1_balance_ce_loss(pred_o, batch)
        loss_o.backward()
        print("oracle DB detector %dx%d N=%d fwd+bwd (f32 and f64): %.1f s" % (size, size, n, time.time() - t0))
    loss64 = float(l1_balance_ce_loss(out64, {k: v.double() for k, v in batch.items()}))
    c.update(grads32=_grads(ora), grads64=grads64, out64=out64,
             out32={"pred": {k: v.detach() for k, v in pred_o.items()}, "loss": float(loss_o), "loss64": loss64})
    c["yardstick"] = _yard_thunk(ora, fwd, out64)
    _CACHE["db"] = c
    return c


def to_device(batch):
    return _to_dev(batch)
