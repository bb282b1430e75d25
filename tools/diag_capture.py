"""hipGraph capture of one training step per workload / dtype / batch, each in its own process (a crash in one does not
hide the others):    python tools/diag_capture.py            # driver: spawns the cases
                     python tools/diag_capture.py res50ppm f32 256"""
import faulthandler
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def case(workload, dtype_name, n, keep=None, pre=()):
    import torch
    import megreader_amd as mr
    from megreader_amd.optim import FusedAdam, FusedSGD
    from megreader_amd.runtime import GraphedTrainStep
    from megreader_amd.synthetic import detection_batch, recognition_batch, recognition_batch_2d
    faulthandler.enable()
    dev = "cuda"
    dtype = torch.float32 if dtype_name == "f32" else torch.bfloat16
    mr.set_compute_dtype(dtype)
    torch.manual_seed(0)

    class M(torch.nn.Module):
        def __init__(self, b, d, crit=None):
            super().__init__()
            self.backbone, self.decoder, self.criterion = b, d, crit

        def forward(self, x, **k):
            return self.decoder(self.backbone(x), **k)
    if workload == "res50ppm":
        from megreader_amd.backbones import resnet50dilated_ppm
        from megreader_amd.decoders import CTCDecoder2D
        model = M(resnet50dilated_ppm(), CTCDecoder2D(in_channels=256))
        b = recognition_batch_2d(n, 32, 128, seed=0, max_len=3)
    elif workload == "fpn":
        from megreader_amd.backbones import Resnet50FPN
        from megreader_amd.decoders import AttentionDecoder
        model = M(Resnet50FPN(resnet_pretrained=False), AttentionDecoder(in_channels=256, gt_as_output=True))
        b = recognition_batch(n, 64, 256, seed=0)
    elif workload == "db":
        from megreader_amd.backbones import deformable_resnet50
        from megreader_amd.decoders import L1BalanceCELoss, SegDetector
        model = M(deformable_resnet50(pretrained=False), SegDetector(in_channels=[256, 512, 1024, 2048], adaptive=True, k=50),
                  L1BalanceCELoss())
        b = detection_batch(n, 640, seed=0)
    else:
        from megreader_amd.backbones import crnn_backbone
        from megreader_amd.decoders import CRNNDecoder
        model = M(crnn_backbone(), CRNNDecoder(in_channels=512, inner_channels=256))
        b = recognition_batch(n, 32, 128, seed=0)
    model.to(dev).train()
    b = {k: v.to(dev) for k, v in b.items()}
    opt = FusedSGD(model.parameters(), lr=0.0, momentum=0.9) if workload == "db" else FusedAdam(model.parameters(), lr=0.0)

    def loss_fn():
        if workload == "db":
            return model.criterion(model(b['image']), b)[0]
        return model(b['image'], targets=b['label'], lengths=b['length'].long(), train=True)[0].mean()
    loss = None
    import contextlib
    side = torch.cuda.Stream() if "side" in pre else None
    for _ in range(2):
      with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        opt.zero_grad()
        if "tmp" in pre:
            loss_fn().backward()          # the loss tensor (and its autograd graph) dies right after backward
        else:
            loss = loss_fn()              # ... stays alive (tests/test_timed_step_gpu.py keeps it for float(loss))
            loss.backward()
        opt.step()
        if "item" in pre:
            float(loss)
    torch.cuda.synchronize()
    if "del" in pre:
        del loss
    if "gc" in pre:
        import gc
        gc.collect()
    if "detach" in pre:
        loss = loss.detach()
    torch.cuda.synchronize()
    print("eager ok, memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30), flush=True)
    # what tests/test_timed_step_gpu.py does between the eager steps and the capture (bisection of its capture_end crash)
    params = [p for p in model.parameters() if p.grad is not None]
    if "cpu" in pre:
        host = [p.grad.double().cpu() for p in params]
        print("copied %d gradients to the host" % len(host), flush=True)
    if "cpu32" in pre:
        host = [p.grad.cpu() for p in params]
    if "clone" in pre:
        held = [p.grad.detach().clone() for p in params]
        print("cloned %d gradients" % len(held), flush=True)
    if "named" in pre:
        for k, p in model.named_parameters():
            if p.grad is not None:
                assert p.grad.data_ptr() == p._mr_grad_sink.data_ptr(), k
    if "sleep" in pre:
        import time
        time.sleep(8)
    g = GraphedTrainStep(loss_fn, opt, [], warmup=1)
    print("capture ok", flush=True)
    for _ in range(2):
        loss = g()
    torch.cuda.synchronize()
    print("replay ok, loss %.5f" % float(loss), flush=True)
    if keep is not None:
        keep.append((g, model, opt))


def seq(spec):
    """Several captures in ONE process: spec = 'crnn:f32:256,res50ppm:f32:256[,keep][,gc]'."""
    import gc
    import torch
    parts = spec.split(",")
    keep = [] if "keep" in parts else None
    for part in parts:
        if ":" not in part:
            continue
        wl, dt, n = part.split(":")
        print("-- capture", wl, dt, n, flush=True)
        case(wl, dt, int(n), keep)
        if "gc" in parts:
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) == 6 and sys.argv[1] == "pre":
        case(sys.argv[2], sys.argv[3], int(sys.argv[4]), pre=tuple(sys.argv[5].split("+")))
    elif len(sys.argv) == 2 and sys.argv[1] == "pres":
        for wl, dt, n, pre, mode in (("db", "f32", 2, "keep", "own"), ("db", "f32", 2, "keep", "same"),
                                     ("db", "f32", 2, "keep+side", "own"), ("db", "f32", 2, "del+gc", "own"),
                                     ("db", "f32", 2, "del", "same"), ("fpn", "f32", 32, "keep", "same"),
                                     ("res50ppm", "f32", 256, "keep", "same"), ("crnn", "f32", 256, "keep", "same")):
            pre = pre + "|" + mode
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "pre", wl, dt, str(n), pre.split("|")[0]],
                                 capture_output=True, text=True, timeout=600,
                                 env=dict(os.environ, MEGREADER_CAPTURE_STREAM=mode))
            lines = [ln for ln in (out.stdout + out.stderr).splitlines() if ln.strip() and "Warning" not in ln and
                     "amdgpu.ids" not in ln and "Consider using" not in ln and "float(loss" not in ln]
            print("== %s %s N=%d pre=%s: rc %d" % (wl, dt, n, pre, out.returncode))
            for ln in lines[-4:]:
                print("   ", ln[:160])
    elif len(sys.argv) == 3 and sys.argv[1] == "seq":
        seq(sys.argv[2])
    elif len(sys.argv) == 2 and sys.argv[1] == "seqs":
        for spec in ("crnn:f32:256,res50ppm:f32:256", "crnn:f32:256,res50ppm:f32:256,gc", "crnn:f32:256,res50ppm:f32:256,keep",
                     "crnn:bf16:256,res50ppm:bf16:256,fpn:bf16:32,db:bf16:2", "res50ppm:f32:256,crnn:f32:256",
                     "crnn:f32:32,res50ppm:f32:32", "res50ppm:f32:64,res50ppm:f32:64"):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "seq", spec], capture_output=True, text=True,
                                 timeout=900)
            lines = [ln for ln in (out.stdout + out.stderr).splitlines() if ln.strip() and "Warning" not in ln and
                     "amdgpu.ids" not in ln and "Consider using" not in ln and "float(loss)" not in ln]
            print("== %s: rc %d" % (spec, out.returncode))
            for ln in lines[-7:]:
                print("   ", ln[:200])
    elif len(sys.argv) >= 4:
        case(sys.argv[1], sys.argv[2], int(sys.argv[3]))
    else:
        for wl, dt, n in (("res50ppm", "f32", 32), ("res50ppm", "f32", 256), ("res50ppm", "bf16", 256), ("fpn", "f32", 32),
                          ("db", "f32", 2), ("crnn", "f32", 256)):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), wl, dt, str(n)], capture_output=True, text=True,
                                 timeout=600)
            tail = [ln for ln in (out.stdout + out.stderr).splitlines() if ln.strip()][-6:]
            print("== %s %s N=%d: rc %d" % (wl, dt, n, out.returncode))
            for ln in tail:
                print("   ", ln[:200])
